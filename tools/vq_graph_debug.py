"""Compares every workspace buffer of an eager VQ-VAE step with a HIP-graph replay (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posterior_matching_amd import optim
from posterior_matching_amd.engine import VQVAETrainStep
from posterior_matching_amd.models.vqvae import VQVAE
from tests.ref_configs import vqvae_mnist

cfg = vqvae_mnist()
B = 16
def make(graph):
    m = VQVAE(**cfg["model"], device="cuda:0", seed=9); m.init((28, 28, 1)); m.store.use_bf16 = False
    gen = torch.Generator().manual_seed(9)
    m.load_params({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    return m, VQVAETrainStep(m, optim.adam(3e-4), B, (28, 28, 1), use_graph=graph)
ma, ta = make(False)
mb, tb = make(True)
rng = np.random.default_rng(3)
for step in range(3):
    xb = torch.tensor(rng.uniform(size=(B, 28, 28, 1)) * (rng.uniform(size=(B, 28, 28, 1)) < 0.3)).float().cuda()
    for t in (ta, tb):
        t.set_batch(xb); t.step()
    print(step, ta.read_metrics()["loss"], tb.read_metrics()["loss"])
    bad = []
    for key, a in ma.ws._bufs.items():
        b = mb.ws._bufs[key]
        if not torch.equal(a, b):
            err = (a.float() - b.float()).abs().max().item()
            if err > 1e-4 or not np.isfinite(err):
                bad.append((key[0], err))
    print("   differing buffers:", bad[:12])
