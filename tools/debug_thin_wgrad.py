import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops
from posterior_matching_amd.ops import LayerGeom
d = torch.device("cuda:0")
g = LayerGeom.conv_t(28, 28, 32, 1, 5, 1, "SAME")
B = 256
torch.manual_seed(0)
x = torch.rand((B, 28, 28, 32), device=d)
dy = -torch.rand((B, 28, 28, 1), device=d) / 256
ref = None
worst = 0.0
for it in range(30):
    dw, db = torch.zeros(g.weight_shape, device=d), torch.zeros(1, device=d)
    ops.layer_wgrad(g, x, dy, dw, db)
    torch.cuda.synchronize()
    if ref is None:
        ref = dw.clone()
        # float64 reference on the CPU
        import torch.nn.functional as F
        xc, dyc = x.cpu().double(), dy.cpu().double()
        # dw[ky,kx,co=0,ci] = sum_{b,iy,ix} dy[b, iy+2-ky, ix+2-kx] * x[b,iy,ix,ci]
        pad = F.pad(dyc[..., 0], (2, 2, 2, 2))
        want = torch.zeros(5, 5, 1, 32, dtype=torch.float64)
        for ky in range(5):
            for kx in range(5):
                sh = pad[:, 4 - ky:4 - ky + 28, 4 - kx:4 - kx + 28]
                want[ky, kx, 0] = (sh[..., None] * xc).sum((0, 1, 2))
        print("max |want|", want.abs().max().item(), "err vs f64", (ref.cpu().double() - want).abs().max().item())
    worst = max(worst, (dw - ref).abs().max().item())
print("worst repeat diff", worst)
