"""Sequential eager-then-graph VQ-VAE runs in one process, reporting buffers with huge / non-finite values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posterior_matching_amd import optim
from posterior_matching_amd.engine import VQVAETrainStep
from posterior_matching_amd.models.vqvae import VQVAE
from tests.ref_configs import vqvae_mnist

cfg = vqvae_mnist()
B = 16
def run(graph):
    m = VQVAE(**cfg["model"], device="cuda:0", seed=9); m.init((28, 28, 1)); m.store.use_bf16 = False
    gen = torch.Generator().manual_seed(9)
    m.load_params({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    t = VQVAETrainStep(m, optim.adam(3e-4), B, (28, 28, 1), use_graph=graph)
    rng = np.random.default_rng(3)
    for step in range(4):
        xb = torch.tensor(rng.uniform(size=(B, 28, 28, 1)) * (rng.uniform(size=(B, 28, 28, 1)) < 0.3))
        t.set_batch(xb.float().to("cuda:0")); t.step()
        print(graph, step, t.read_metrics())
        pd = m.params_dict(); sd = m.state_dict()
        for key, a in m.ws._bufs.items():
            mx = a.float().abs().max().item()
            if not np.isfinite(mx) or mx > 1e6:
                print("    bad buffer", key[0], mx)
        print("    log_scale", m.store.p["decoder/log_scale"].item(), "x max", t.x.max().item())
run(False)
run(True)
