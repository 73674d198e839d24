#!/usr/bin/env python3
"""Micro-benchmark of single layers through the C ABI (HIP events, L2-warm, back-to-back).
    python tools/bench_layer.py [libpath]      # optional alternative libpmhip build (experiments)
Times are host-side (events around 50 back-to-back launches): kernels under ~15 us read as the launch rate.
PM_TIMER=1 prints device-side per-kernel averages instead (ops.KernelTimer).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import posterior_matching_amd._lib as L

    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

from posterior_matching_amd import ops
from posterior_matching_amd.ops import ACT_LEAKY, LayerGeom

CASES = [
    ("dec5 fwd 28x28 32->32 k5", LayerGeom.conv_t(28, 28, 32, 32, 5, 1, "SAME"), "fwd"),
    ("enc3 fwd 14x14 32->64 k5", LayerGeom.conv(14, 14, 32, 64, 5, 1, "SAME"), "fwd"),
    ("dec3 fwd 14x14 64->32 k5", LayerGeom.conv_t(14, 14, 64, 32, 5, 1, "SAME"), "fwd"),
    ("enc2 fwd 28->14 s2 32->32", LayerGeom.conv(28, 28, 32, 32, 5, 2, "SAME"), "fwd"),
    ("dec4 fwd 14->28 s2 32->32", LayerGeom.conv_t(14, 14, 32, 32, 5, 2, "SAME"), "fwd"),
    ("mlp dense 8192x256x256", LayerGeom.dense(256, 256), "fwd8192"),
    ("dec5 wgrad", LayerGeom.conv_t(28, 28, 32, 32, 5, 1, "SAME"), "wgrad"),
    ("enc3 wgrad", LayerGeom.conv(14, 14, 32, 64, 5, 1, "SAME"), "wgrad"),
    ("mlp wgrad 8192x256x256", LayerGeom.dense(256, 256), "wgrad8192"),
    ("enc4 fwd 14->7 s2 64->64", LayerGeom.conv(14, 14, 64, 64, 5, 2, "SAME"), "fwd"),
    ("enc2 wgrad 28->14 s2 32->32", LayerGeom.conv(28, 28, 32, 32, 5, 2, "SAME"), "wgrad"),
    ("dec3 wgrad 14x14 64->32 k5", LayerGeom.conv_t(14, 14, 64, 32, 5, 1, "SAME"), "wgrad"),
    ("enc4 wgrad 14->7 s2 64->64", LayerGeom.conv(14, 14, 64, 64, 5, 2, "SAME"), "wgrad"),
    ("enc4 dgrad 7->14 d2 64->64", LayerGeom.conv(14, 14, 64, 64, 5, 2, "SAME"), "dgrad"),
    ("enc5 fwd 7x7x64->128 k7", LayerGeom.conv(7, 7, 64, 128, 7, 1, "VALID"), "fwd"),
    ("enc5 dgrad", LayerGeom.conv(7, 7, 64, 128, 7, 1, "VALID"), "dgrad"),
    ("enc5 wgrad", LayerGeom.conv(7, 7, 64, 128, 7, 1, "VALID"), "wgrad"),
    ("mlp dgrad 8192x256x256", LayerGeom.dense(256, 256), "dgrad8192"),
    ("mlp grouped wgrad 4x8192x256x256", LayerGeom.dense(256, 256), "gwgrad8192"),
    ("vd 1x1 192->48 14x14", LayerGeom.conv(14, 14, 192, 48, 1, 1, "SAME"), "fwd"),
    ("vd 3x3 48->48 14x14", LayerGeom.conv(14, 14, 48, 48, 3, 1, "SAME"), "fwd"),
    ("vd 1x1 48->192 14x14", LayerGeom.conv(14, 14, 48, 192, 1, 1, "SAME"), "fwd"),
    ("vd dense 192->48", LayerGeom.dense(192, 48), "fwdrows"),
    ("vd dense 48->192", LayerGeom.dense(48, 192), "fwdrows"),
    ("thin enc0 fwd 28x28 1->32 k5", LayerGeom.conv(28, 28, 1, 32, 5, 1, "SAME"), "fwd"),
    ("thin penc0 fwd 28x28 2->32 k5", LayerGeom.conv(28, 28, 2, 32, 5, 1, "SAME"), "fwd"),
    ("thin dec6 fwd 28x28 32->1 k5", LayerGeom.conv_t(28, 28, 32, 1, 5, 1, "SAME"), "fwd"),
    ("thin dec6 dgrad 28x28 1->32 k5", LayerGeom.conv_t(28, 28, 32, 1, 5, 1, "SAME"), "dgrad"),
    ("thin enc0 wgrad", LayerGeom.conv(28, 28, 1, 32, 5, 1, "SAME"), "wgrad"),
    ("thin penc0 wgrad", LayerGeom.conv(28, 28, 2, 32, 5, 1, "SAME"), "wgrad"),
    ("thin dec6 wgrad", LayerGeom.conv_t(28, 28, 32, 1, 5, 1, "SAME"), "wgrad"),
    # CelebA PixelCNN (PM_BENCH_B=16): masked 3x3 layers on the 16 x 16 code grid
    ("celeb pcnn 256->256 m2x3 fwd", LayerGeom.masked_conv(16, 16, 256, 256, 3, 3, 2, 3), "fwd"),
    ("celeb pcnn 256->128 m2x3 fwd", LayerGeom.masked_conv(16, 16, 256, 128, 3, 3, 2, 3), "fwd"),
    ("celeb pcnn 256->256 m2x2 fwd", LayerGeom.masked_conv(16, 16, 256, 256, 3, 3, 2, 2), "fwd"),
    ("celeb pcnn 256->256 m2x3 dgrad", LayerGeom.masked_conv(16, 16, 256, 256, 3, 3, 2, 3), "dgrad"),
]
if os.environ.get("PM_CASES"):
    CASES = [c for c in CASES if any(k in c[0] for k in os.environ["PM_CASES"].split(","))]


def main():
    d = torch.device("cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for name, g, what in CASES:
            B = 8192 if what.endswith("8192") else int(os.environ.get("PM_BENCH_B", "256"))
            if what == "fwdrows":
                B, what = B * 196, "fwd"
            x = torch.randn((B, g.IH, g.IW, g.CI), device=d)
            w = torch.randn(g.weight_shape, device=d) * 0.05
            b = torch.zeros(g.CO, device=d)
            y = torch.empty((B, g.OH, g.OW, g.CO), device=d)
            dw, db = torch.zeros_like(w), torch.zeros_like(b)
            if what.startswith("fwd"):
                from posterior_matching_amd.models.core import ParamStore
                st = ParamStore()
                st.add("w", g.weight_shape, fan_in=1)
                hf = st.request_split("w", g, "fwd")
                st.allocate(d)
                st.load_dict({"w": w.cpu()})
                ws = st.split_view(hf) if os.environ.get("PM_BF16", "1") == "1" else None
                tmp = torch.empty((B, g.IH, g.IW, g.KH * g.KW), device=d) if g.CO == 1 else None
                fn = lambda: ops.layer_forward(g, x, w, b, y, out_act=ACT_LEAKY, wsplit=ws, tmp=tmp)
            elif what.startswith("dgrad"):
                from posterior_matching_amd.models.core import ParamStore
                st = ParamStore()
                st.add("w", g.weight_shape, fan_in=1)
                hd = st.request_split("w", g, "dgrad")
                st.allocate(d)
                st.load_dict({"w": w.cpu()})
                ws = st.split_view(hd) if os.environ.get("PM_BF16", "1") == "1" else None
                y.normal_()
                fn = lambda: ops.layer_dgrad(g, y, w, x, aux=x.clone(), aux_act=ACT_LEAKY, wsplit=ws)
            elif what.startswith("gwgrad"):
                G = 4
                xg, yg = torch.randn((G, B, g.CI), device=d), torch.randn((G, B, g.CO), device=d)
                dwg, dbg = torch.zeros((G,) + tuple(w.shape), device=d), torch.zeros((G, g.CO), device=d)
                fn = lambda: ops.layer_wgrad(g, xg, yg, dwg, dbg, B=B, groups=G, in_gs=B * g.CI, out_gs=B * g.CO,
                                             w_gs=g.CI * g.CO, bias_gs=g.CO)
            else:
                y.normal_()
                fn = lambda: ops.layer_wgrad(g, x, y, dw, db)
            for _ in range(5):
                fn()
            e0, e1 = ops.Event(), ops.Event()
            e0.record()
            reps = 50
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_ms(e1) / reps * 1e3
            kk = g.KH * g.KW
            macs = B * g.OH * g.OW * kk * g.CI * g.CO if g.kind != "convT" or g.s == 1 else B * g.IH * g.IW * kk * g.CI * g.CO
            if g.kind == "conv" and g.s > 1:
                macs = B * g.OH * g.OW * kk * g.CI * g.CO
            kt = ops.KernelTimer()
            ops.set_timer(kt)
            for _ in range(10):
                fn()
            ops.set_timer(None)
            dev_us = " + ".join(f"{v['ms'] / v['calls'] * 1e3:.1f}" for v in kt.summary().values())
            print(f"{name:30s} {us:8.1f} us host-paced  {2 * macs / us / 1e6:7.1f} TFLOP/s (nominal)   device: {dev_us} us")


if __name__ == "__main__":
    main()
