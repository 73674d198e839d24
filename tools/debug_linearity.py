import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_parity import _inputs, _product_model
cfg, xs, x, b, eps = _inputs("mnist", 256, 9)
m = _product_model(cfg, xs)
d = torch.device("cuda:0")
xd, bd, ed = x.float().to(d), b.float().to(d), eps.float().to(d)
m(xd, bd, True, eps=ed)
g = [torch.full((256,), v, device=d) for v in (-1 / 256, 1 / 256, -1 / 256)]
def run(scale):
    m.zero_grad(); m.backward(*[scale * t for t in g]); torch.cuda.synchronize()
    return {n: t.clone() for n, t in m.grads_dict().items()}
a, a2, c = run(1.0), run(1.0), run(2.0)
gmax = max(t.abs().max().item() for t in a.values())
print("global max", gmax)
for n in a:
    e_rep = (a[n] - a2[n]).abs().max().item()
    e_lin = (c[n] - 2 * a[n]).abs().max().item()
    if e_rep > 1e-6 * gmax or e_lin > 2e-6 * gmax:
        print(f"{n:50s} max {a[n].abs().max().item():10.4g} repeat {e_rep:10.3g} lin {e_lin:10.3g}")
n = "decoder_net/conv_t_5/w"
diff = (a[n] - a2[n]).abs()
idx = (diff > 1e-3).nonzero()
print("num differing", idx.shape[0], "of", diff.numel())
print(idx[:20].tolist())
print("values", a[n][tuple(idx[0].tolist())].item() if idx.shape[0] else None, a2[n][tuple(idx[0].tolist())].item() if idx.shape[0] else None)
if len(sys.argv) > 1:
    m.concurrent = False
    m.ws.overlap_wgrad = False
    a, a2 = run(1.0), run(1.0)
    print("single stream: repeat diff", max((a[k] - a2[k]).abs().max().item() for k in a))
