import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops
from posterior_matching_amd.ops import LayerGeom, ACT_LEAKY
from posterior_matching_amd.models.core import ParamStore
d = torch.device("cuda:0")
torch.manual_seed(0)
B = 64
gp = LayerGeom.conv(28, 28, 32, 32, 5, 1, "SAME")
xp, yp = torch.randn((256, 28, 28, 32), device=d), torch.empty((256, 28, 28, 32), device=d)
wp = torch.randn(gp.weight_shape, device=d) * 0.05
st = ParamStore(); st.add("w", gp.weight_shape, fan_in=1); hf = st.request_split("w", gp, "fwd"); st.allocate(d); st.load_dict({"w": wp.cpu()})
wsp = st.split_view(hf)
s2 = torch.cuda.Stream()
cases = {"lane_fwd": LayerGeom.conv(28, 28, 1, 32, 5, 1, "SAME"), "lane_fwd2": LayerGeom.conv(28, 28, 2, 32, 5, 1, "SAME"),
         "to1": LayerGeom.conv_t(28, 28, 32, 1, 5, 1, "SAME")}
for name, g in cases.items():
    x = torch.rand((B, g.IH, g.IW, g.CI), device=d)
    w = torch.randn(g.weight_shape, device=d) * 0.1
    b = torch.zeros(g.CO, device=d)
    st2 = ParamStore(); st2.add("w", g.weight_shape, fan_in=1); h2 = st2.request_split("w", g, "fwd"); st2.allocate(d); st2.load_dict({"w": w.cpu()})
    for variant in ("plain", "bf16"):
        ws = st2.split_view(h2) if variant == "bf16" else None
        ref, worst = None, 0.0
        for it in range(30):
            y = torch.zeros((B, g.OH, g.OW, g.CO), device=d)
            torch.cuda.synchronize()
            with torch.cuda.stream(s2):
                for _ in range(3):
                    ops.layer_forward(gp, xp, wp, None, yp, wsplit=wsp)
            ops.layer_forward(g, x, w, b, y, out_act=ACT_LEAKY, wsplit=ws)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            worst = max(worst, (y - ref).abs().max().item())
        print(name, variant, "worst repeat diff", worst, "max", ref.abs().max().item())
