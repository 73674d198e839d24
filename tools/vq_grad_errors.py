"""Per-tensor gradient error table of the VQ-VAE HIP path vs the float64 oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vqvae_oracle as VO
from tests.test_gpu_vqvae import _setup, rel_err, dev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf = len(sys.argv) > 2 and sys.argv[2] == "bf16"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
cfg, x, m, p64, st64 = _setup(B, seed=seed, bf16x3=bf)
leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
loss, aux, out, new_state = VO.vqvae_loss(leaves, st64, cfg, x, True)
grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
got = m(x.float().to(dev()), is_training=True)
m.zero_grad(); m.backward(); torch.cuda.synchronize()
gd = m.grads_dict()
for n in grads:
    print(f"{n:28s} {rel_err(gd[n], grads[n]):.3e}  |g|={grads[n].norm().item():.3e}")

# intermediate gradients
leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
h = VO.conv_residual_encoder(leaves, "encoder", x, 2); h.retain_grad()
z = VO.conv2d(h, leaves["pre_vq_conv/w"], leaves["pre_vq_conv/b"], 1, "SAME"); z.retain_grad()
vq, _ = VO.vector_quantizer_ema(st64, z, 0.25, 0.99, True)
q = vq["quantize"]; q.retain_grad()
loc, scale = VO.conv_residual_decoder(leaves, "decoder", q, 2)
from oracle.pm_vae_oracle import normal_log_prob
ll = normal_log_prob(x, loc, scale).reshape(B, -1).sum(1)
(-ll.mean() + vq["loss"]).backward()
ws = m.ws
def get(name):
    for (n, shp), t in ws._bufs.items():
        if n == name: return t
    raise KeyError(name)
print("dz   ", rel_err(get("decoder/dz"), z.grad))
print("dfeat", rel_err(get("vqvae/dfeat"), h.grad * (h > 0)))
dq_only = q.grad
print("commit", rel_err(get("vq/commit_grad"), z.grad - q.grad))
d = (get("vqvae/dfeat").double().cpu() - h.grad * (h > 0))
print("max abs err rows:", d.abs().reshape(B, -1).max(1).values)
i = d.abs().argmax().item()
idx = np.unravel_index(i, d.shape) if (np := __import__("numpy")) else None
print("worst element", idx, "err", d.reshape(-1)[i].item(), "oracle h", h.reshape(-1)[i].item(),
      "gpu feat", get("encoder/res_h_1").reshape(-1)[i].item(), "oracle grad", h.grad.reshape(-1)[i].item())
nb = (d.abs() > 1e-6).sum().item()
print("elements off:", nb)
