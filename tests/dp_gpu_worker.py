"""Worker of tests/test_gpu_dp.py: 2 ranks (gloo) sharing ONE GPU run the data-parallel train steps of every
engine on their halves of a global batch; rank 0 then replays the same steps in a single-process engine on
the full batch and compares parameters (and VQ state).  Launch-plan replay is on (step 3+)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.engine import PMVAETrainStep, VDVAETrainStep, VQVAETrainStep  # noqa: E402
from posterior_matching_amd.models import PosteriorMatchingVAE  # noqa: E402
from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE  # noqa: E402
from posterior_matching_amd.models.vqvae import VQVAE  # noqa: E402
from posterior_matching_amd.parallel import init_distributed, shard_rows  # noqa: E402
from tests.ref_configs import pm_vae_gas, vqvae_mnist  # noqa: E402
from tests.test_gpu_vdvae import TINY  # noqa: E402

rank, _, world = init_distributed("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
STEPS = 4


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def compare(tag, a, b, tol):
    worst = max((rel(a[k], b[k]), k) for k in a)
    assert worst[0] < tol, (tag, worst)
    return worst[0]


rng = np.random.default_rng(5)
report = {}

# ---- PM-VAE (gas config, dense) ----
cfg, G = pm_vae_gas(), 64
xs = torch.tensor(rng.normal(size=(STEPS, G, 8)), dtype=torch.float32, device=dev)
bs = torch.tensor(rng.uniform(size=(STEPS, G, 8)) < 0.5, dtype=torch.float32, device=dev)
es = torch.tensor(rng.normal(size=(STEPS, G, 16)), dtype=torch.float32, device=dev)


def pmvae(world_size, r, rows, overlap=True):
    m = PosteriorMatchingVAE.from_config(cfg["model"], device=dev, seed=3)
    m.init((8,))
    m.store.use_bf16 = False
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg["weight_decay"]),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(m, cfg, opt, rows.stop - rows.start, (8,), world_size=world_size, rank=r, external_eps=True,
                        overlap_allreduce=overlap)
    for s in range(STEPS):
        ts.set_batch(xs[s, rows], bs[s, rows], es[s, rows])
        ts.step()
    ts.synchronize()
    if world_size > 1:          # bucketed: decoder / posterior-matching branch / encoder + the 1-D suffix; else one call
        assert ts.reducer.calls_last_step == (1 if not overlap else ts.reducer.calls_last_step) >= 1
        report[f"pm_vae_allreduce_calls_overlap_{overlap}"] = ts.reducer.calls_last_step
    return m.params_dict()


dp = pmvae(world, rank, shard_rows(G, rank, world))
dp_single = pmvae(world, rank, shard_rows(G, rank, world), overlap=False)
for k in dp:                                 # bucketed + overlapped == one all-reduce after the backward pass
    assert rel(dp[k], dp_single[k]) < 1e-5, ("overlap changes the trajectory", k)     # atomics: not bitwise
assert report["pm_vae_allreduce_calls_overlap_True"] > report["pm_vae_allreduce_calls_overlap_False"] == 1
if rank == 0:
    report["pm_vae"] = compare("pm_vae", dp, pmvae(1, 0, slice(0, G)), 2e-4)

# ---- VQ-VAE (EMA statistics psum-ed across ranks) ----
vcfg, G = vqvae_mnist(), 16
imgs = torch.tensor(rng.uniform(size=(STEPS, G, 28, 28, 1)) * (rng.uniform(size=(STEPS, G, 28, 28, 1)) < 0.3),
                    dtype=torch.float32, device=dev)


def vqvae(world_size, r, rows):
    m = VQVAE(**vcfg["model"], device=dev, seed=4)
    m.init((28, 28, 1))
    m.store.use_bf16 = False
    if world_size > 1:
        m.vq.cross_replica_axis = "i"
    ts = VQVAETrainStep(m, optim.adam(vcfg["learning_rate"]), rows.stop - rows.start, (28, 28, 1),
                        world_size=world_size, rank=r)
    for s in range(STEPS):
        ts.set_batch(imgs[s, rows])
        ts.step()
    ts.synchronize()
    return m.params_dict(), m.state_dict()


dp_p, dp_s = vqvae(world, rank, shard_rows(G, rank, world))
if rank == 0:
    one_p, one_s = vqvae(1, 0, slice(0, G))
    report["vqvae"] = compare("vqvae", dp_p, one_p, 3e-4)
    assert int(dp_s["counter"]) == STEPS
    report["vqvae_state"] = compare("vqvae_state", {k: v for k, v in dp_s.items() if k != "counter"},
                                    {k: v for k, v in one_s.items() if k != "counter"}, 3e-4)

# ---- VDVAE (global-norm clip on the reduced gradient, EMA) ----
G = 8
ximg = torch.tensor(np.round(rng.uniform(size=(STEPS, G, 7, 7, 1)) * 255.0), dtype=torch.float32, device=dev)
bimg = torch.tensor(rng.uniform(size=(STEPS, G, 7, 7, 1)) < 0.5, dtype=torch.float32, device=dev)


def vdvae(world_size, r, rows):
    m = PosteriorMatchingVDVAE(**TINY["model"], device=dev, seed=6)
    m.init()
    m.store.use_bf16 = False
    gen = torch.Generator().manual_seed(1)
    m.load_params({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    ts = VDVAETrainStep(m, TINY["lr"], rows.stop - rows.start, gradient_clip=20.0, ema_rate=0.999, world_size=world_size,
                        rank=r, external_eps=True)
    g2 = np.random.default_rng(9)
    for s in range(STEPS):
        eps = [torch.tensor(g2.normal(size=(G,) + sh[1:]), dtype=torch.float32, device=dev)[rows] for sh in m.eps_shapes(G)]
        ts.set_batch(ximg[s, rows], bimg[s, rows], eps)
        ts.step()
    ts.synchronize()
    return m.params_dict(), ts.ema_params()


dp_p, dp_e = vdvae(world, rank, shard_rows(G, rank, world))
if rank == 0:
    one_p, one_e = vdvae(1, 0, slice(0, G))
    report["vdvae"] = compare("vdvae", dp_p, one_p, 3e-4)
    report["vdvae_ema"] = compare("vdvae_ema", dp_e, one_e, 3e-4)

dist.barrier()
if rank == 0:
    print("DP-GPU-OK", report)
dist.destroy_process_group()
