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

# ---- the default arithmetic (bf16x3) and the launch forms the Trainer uses ------------------------------------------
# Gradients: after the FIRST step the flat gradient buffer holds the all-reduced sum of the per-rank mean gradients, i.e.
# world x the full-batch gradient; later steps compare losses (Adam's first updates are sign-like, parameters of two
# float32 runs drift apart on entries whose gradient is rounding noise).
def grad_and_losses(ts, feed, steps=3):
    losses, g1 = [], None
    ts.adam_cfg.zero_grad = 0            # keep the gradient buffer readable after step() (default: Adam zeroes it)
    for s in range(steps):
        feed(s)
        ts.step()
        losses.append(ts.read_metrics()["loss"])
        if s == 0:
            ts.synchronize()
            g1 = ts_store(ts).flat_g.clone()
    ts.synchronize()
    return g1, losses


def ts_store(ts):
    return ts.store if hasattr(ts, "store") else ts.model.store


def check_dp(tag, run, G, gtol=1e-3, ltol=2e-4, min_early=0):
    rows = shard_rows(G, rank, world)
    ts, (g_dp, l_dp) = run(world, rank, rows)
    calls, early = ts.reducer.calls_last_step, ts.reducer.calls_before_finish_last_step
    assert calls >= 1 and early >= min_early, (tag, calls, early)
    lt = torch.tensor(l_dp, dtype=torch.float64)
    dist.all_reduce(lt)                                  # mean over ranks of the per-rank mean losses
    if rank == 0:
        _, (g_one, l_one) = run(1, 0, slice(0, G))
        e = rel(g_dp / world, g_one)
        assert e < gtol, (tag, "gradient", e)
        for a, b in zip((lt / world).tolist(), l_one):
            assert abs(a - b) <= ltol * abs(b), (tag, "loss", a, b)
        report[tag] = {"grad_rel_err": e, "allreduce_calls": calls, "issued_before_finish": early}


# conv PM-VAE (configs/pm_vae_mnist.py, the headline workload), global batch 64; eager / launch plan and HIP graph
from tests.ref_configs import pm_vae_mnist, pm_vqvae_mnist  # noqa: E402

mcfg, G = pm_vae_mnist(), 64
mx = torch.tensor(rng.uniform(size=(3, G, 28, 28, 1)) * (rng.uniform(size=(3, G, 28, 28, 1)) < 0.3), dtype=torch.float32, device=dev)
mb = torch.tensor(rng.uniform(size=(3, G, 28, 28, 1)) < 0.5, dtype=torch.float32, device=dev)
me = torch.tensor(rng.normal(size=(3, G, 32)), dtype=torch.float32, device=dev)


def conv_pmvae(use_graph):
    def run(world_size, r, rows):
        m = PosteriorMatchingVAE.from_config(mcfg["model"], device=dev, seed=3)
        m.init((28, 28, 1))
        opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                          optim.scale_by_schedule(optim.exponential_decay(**mcfg["lr_schedule"])), optim.scale(-1.0))
        ts = PMVAETrainStep(m, mcfg, opt, rows.stop - rows.start, (28, 28, 1), world_size=world_size, rank=r,
                            external_eps=True, use_graph=use_graph)
        return ts, grad_and_losses(ts, lambda s: ts.set_batch(mx[s, rows], mb[s, rows], me[s, rows]))
    return run


check_dp("pm_vae_mnist_bf16x3", conv_pmvae(False), G)
check_dp("pm_vae_mnist_bf16x3_hip_graph", conv_pmvae(True), G)      # the Trainer's default launch form: reduce between the graphs

# PM-VQVAE stage 2 (grouped weight gradients; the up pass's buckets leave while the down pass still runs)
from posterior_matching_amd.engine import PMVQVAETrainStep  # noqa: E402
from posterior_matching_amd.models.pixel_cnn import PixelCNN  # noqa: E402
from posterior_matching_amd.models.vqvae import VQVAEPartialEncoder  # noqa: E402

S2_VQ = {"embedding_dim": 32, "num_embeddings": 24, "hidden_units": 32, "residual_hidden_units": 32,
         "residual_blocks": 1, "decay": 0.99, "use_ema": True, "commitment_cost": 0.25, "output_channels": 1}
S2 = {"pixel_cnn": {"image_shape": (3, 3), "num_resnet": 2, "num_hierarchies": 1, "num_filters": 32, "dropout": 0.5},
      "conditional_dim": 64, "lr_schedule": {"init_value": 3e-4, "decay_rate": 0.999995, "transition_steps": 1}}
G = 16
sx = torch.tensor(rng.uniform(size=(3, G, 12, 12, 1)) * (rng.uniform(size=(3, G, 12, 12, 1)) < 0.3), dtype=torch.float32, device=dev)
sb = torch.tensor(rng.uniform(size=(3, G, 12, 12, 1)) < 0.5, dtype=torch.float32, device=dev)
smask = [[torch.tensor((rng.uniform(size=(G, 3, 3, 64)) > 0.5) * 2.0, dtype=torch.float32, device=dev) for _ in range(8)]
         for _ in range(3)]


def stage2(world_size, r, rows):
    vq = VQVAE(**S2_VQ, device=dev, seed=4)
    vq.init((12, 12, 1))
    vq.store.use_bf16 = False
    penc, pcnn = VQVAEPartialEncoder(S2["conditional_dim"], S2_VQ), PixelCNN(**dict(S2["pixel_cnn"], num_indices=24))
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                      optim.scale_by_schedule(optim.exponential_decay(**S2["lr_schedule"])), optim.scale(-1.0))
    os.environ["PM_BUCKET_MB"] = "0.05"                  # this toy has 0.3 MB of gradients: buckets small enough to leave early
    ts = PMVQVAETrainStep(vq, penc, pcnn, opt, rows.stop - rows.start, (12, 12, 1), seed=4, world_size=world_size, rank=r,
                          external_dropout=True)
    os.environ.pop("PM_BUCKET_MB")

    def feed(s):
        ts.dropout_masks = [t[rows].contiguous() for t in smask[s]]
        ts.set_batch(sx[s, rows], sb[s, rows])
    return ts, grad_and_losses(ts, feed)


check_dp("pm_vqvae_stage2_bf16x3", stage2, G, min_early=1)

# VDVAE with fused Blocks and grouped weight gradients flushed per resolution on the side stream (default arithmetic)
SMALL16 = dict(TINY["model"], latent_dim=16, width=64)
G = 8
vx = torch.tensor(np.round(rng.uniform(size=(3, G, 7, 7, 1)) * 255.0), dtype=torch.float32, device=dev)
vb = torch.tensor(rng.uniform(size=(3, G, 7, 7, 1)) < 0.5, dtype=torch.float32, device=dev)


def vdvae_fused(world_size, r, rows):
    m = PosteriorMatchingVDVAE(**SMALL16, device=dev, seed=6)
    m.init()
    gen = torch.Generator().manual_seed(1)
    m.load_params({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    assert any(b[0]._fused() is not None for b in m.encoder.blocks)
    os.environ["PM_BUCKET_MB"] = "0.25"
    ts = VDVAETrainStep(m, TINY["lr"], rows.stop - rows.start, gradient_clip=200.0, ema_rate=0.999, world_size=world_size,
                        rank=r, external_eps=True)
    os.environ.pop("PM_BUCKET_MB")
    g2 = np.random.default_rng(9)
    eps = [[torch.tensor(g2.normal(size=(G,) + sh[1:]), dtype=torch.float32, device=dev) for sh in m.eps_shapes(G)]
           for _ in range(3)]
    return ts, grad_and_losses(ts, lambda s: ts.set_batch(vx[s, rows], vb[s, rows], [e[rows] for e in eps[s]]))


check_dp("pm_vdvae_fused_bf16x3", vdvae_fused, G, min_early=1)

dist.barrier()
if rank == 0:
    print("DP-GPU-OK", report)
dist.destroy_process_group()
