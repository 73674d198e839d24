import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('DBG_LIB'):
    import posterior_matching_amd._lib as L
    L.LIB_PATH = os.path.abspath(os.environ['DBG_LIB'])
from posterior_matching_amd import ops
from posterior_matching_amd.ops import LayerGeom
d = torch.device("cuda:0")
g = LayerGeom.conv_t(28, 28, 32, 1, 5, 1, "SAME")
g2 = LayerGeom.dense(256, 256)
B = int(os.environ.get('DBG_B', '256'))
torch.manual_seed(0)
x = torch.rand((B, 28, 28, 32), device=d)
dy = -torch.rand((B, 28, 28, 1), device=d) / 256
xs, ys = torch.randn((4, 8192, 256), device=d), torch.randn((4, 8192, 256), device=d)
dwg, dbg = torch.zeros((4, 256, 256), device=d), torch.zeros((4, 256), device=d)
gp = LayerGeom.conv(28, 28, 32, 32, 5, 1, "SAME")
xp, yp = torch.randn((256, 28, 28, 32), device=d), torch.empty((256, 28, 28, 32), device=d)
wp = torch.randn(gp.weight_shape, device=d) * 0.05
from posterior_matching_amd.models.core import ParamStore
st = ParamStore(); st.add("w", gp.weight_shape, fan_in=1); hf = st.request_split("w", gp, "fwd"); st.allocate(d); st.load_dict({"w": wp.cpu()})
wsp = st.split_view(hf)
s2 = torch.cuda.Stream()
x0, dy0 = x.clone(), dy.clone()
ref, worst = None, 0.0
mode = sys.argv[1] if len(sys.argv) > 1 else "grouped"
for it in range(40):
    dw, db = torch.zeros(g.weight_shape, device=d), torch.zeros(1, device=d)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        for _ in range(3):
            if mode == "grouped":
                ops.layer_wgrad(g2, xs, ys, dwg, dbg, B=8192, groups=4, in_gs=8192 * 256, out_gs=8192 * 256, w_gs=65536, bias_gs=256)
            elif mode == "torch":
                yp.add_(xp)
            elif mode == "patch":
                ops.layer_forward(gp, xp, wp, None, yp, wsplit=wsp)
            else:
                ops.layer_dgrad(g, dy, torch.ones(g.weight_shape, device=d), x.clone())
    ops.layer_wgrad(g, x, dy, dw, db if mode != "nodb" else None)
    torch.cuda.synchronize()
    if ref is None:
        ref = dw.clone()
    dd = (dw - ref).abs()
    if dd.max().item() > worst:
        worst = dd.max().item()
        bad = (dd > 0.25 * worst).nonzero()
print(mode, "B", B, "worst repeat diff", worst, "ref max", ref.abs().max().item())
print(bad[:12].tolist() if worst > 0 else None)

torch.cuda.synchronize()
print("inputs intact:", torch.equal(x, x0), torch.equal(dy, dy0))
# the same launch again with nothing running beside it
dw2, db2 = torch.zeros(g.weight_shape, device=d), torch.zeros(1, device=d)
ops.layer_wgrad(g, x, dy, dw2, db2)
torch.cuda.synchronize()
print("alone afterwards vs first:", (dw2 - ref).abs().max().item())
