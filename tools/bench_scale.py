#!/usr/bin/env python3
"""dec5-like layer timed at several batch sizes: separates steady-state MFMA efficiency from tail/occupancy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops
from posterior_matching_amd.ops import ACT_LEAKY, LayerGeom
d = torch.device("cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for name, g in (("28x28 32->32", LayerGeom.conv_t(28, 28, 32, 32, 5, 1, "SAME")),
                    ("14x14 32->64", LayerGeom.conv(14, 14, 32, 64, 5, 1, "SAME")),
                    ("dense 256->256 (rows=B*32)", LayerGeom.dense(256, 256))):
        for B in (96, 192, 256, 384, 768, 1536):
            rows = B * 32 if g.kind == "dense" else B
            x = torch.randn((rows, g.IH, g.IW, g.CI), device=d)
            w = torch.randn(g.weight_shape, device=d) * 0.05
            b = torch.zeros(g.CO, device=d)
            y = torch.empty((rows, g.OH, g.OW, g.CO), device=d)
            fn = lambda: ops.layer_forward(g, x, w, b, y, out_act=ACT_LEAKY)
            for _ in range(3): fn()
            e0, e1 = ops.Event(), ops.Event()
            e0.record()
            for _ in range(20): fn()
            e1.record(); e1.synchronize()
            us = e0.elapsed_ms(e1) / 20 * 1e3
            macs = rows * g.OH * g.OW * g.k * g.k * g.CI * g.CO
            wgs = (rows * g.OH * g.OW + 127) // 128 * max(1, g.CO // 64)
            print(f"{name:28s} B={B:5d} wgs={wgs:6d} ({wgs/256:5.1f}/CU) {us:8.1f} us {2*macs/us/1e6:6.1f} TFLOP/s")
