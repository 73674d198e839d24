import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pm_vae_oracle as O
from tests.test_gpu_parity import _inputs, _product_model, rel_err
from posterior_matching_amd import ops
from posterior_matching_amd.engine import loss_cfg_from_config
name, B = sys.argv[1], int(sys.argv[2])
cfg, xs, x, b, eps = _inputs(name, B, 5)
m = _product_model(cfg, xs)
p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
loss, aux, out = O.pm_vae_loss(leaves, cfg, x, b, eps, 0)
grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
d = torch.device("cuda:0")
for use in (False, True):
    m.store.use_bf16 = use
    got = m(x.float().to(d), b.float().to(d), True, eps=eps.float().to(d))
    print("bf16x3" if use else "f32", {k: f"{rel_err(got[k], out[k]):.2e}" for k in got})
    g = [torch.full((B,), v, device=d) for v in (-1.0 / B, 1.0 / B, -1.0 / B)]
    m.zero_grad(); m.backward(*g); torch.cuda.synchronize()
    gd = m.grads_dict()
    for n in grads:
        print(f"   {n:48s} {rel_err(gd[n], grads[n]):.2e}")
