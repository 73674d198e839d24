"""Worker of tests/test_gpu_dp.py::test_rccl_one_rank_rehearsal: ONE process, a 1-rank RCCL (`nccl`) communicator on
cuda:0, PM_FORCE_DP=1 so that every engine builds its GradReducer: the asynchronous, bucketed all-reduce path of the
N > 1 runs (communication stream, work handles, collectives inside a replayed launch plan, clip on the reduced buffer) is
executed with the real backend.  With one rank the sum is the identity, so the trajectories must equal the plain runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.engine import PMVAETrainStep, VDVAETrainStep  # noqa: E402
from posterior_matching_amd.models import PosteriorMatchingVAE  # noqa: E402
from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE  # noqa: E402
from tests.ref_configs import pm_vae_mnist  # noqa: E402
from tests.test_gpu_vdvae import TINY  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
STEPS = 5
rng = np.random.default_rng(9)


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def pmvae(forced: bool):
    os.environ["PM_FORCE_DP"] = "1" if forced else "0"
    cfg, B = pm_vae_mnist(), 64
    m = PosteriorMatchingVAE.from_config(cfg["model"], device=dev, seed=3)
    m.init((28, 28, 1))
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg.get("weight_decay", 0.0)),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(m, cfg, opt, B, (28, 28, 1), seed=5, external_eps=True)
    assert (ts.reducer is not None) == forced
    if forced:
        assert ts.reducer.async_issue, "nccl: collectives are enqueued asynchronously on the communication stream"
    g = np.random.default_rng(1)
    for s in range(STEPS):                      # steps 3+ replay the recorded launch plan, collectives included
        x = torch.tensor(g.uniform(size=(B, 28, 28, 1)) * (g.uniform(size=(B, 28, 28, 1)) < 0.2), dtype=torch.float32, device=dev)
        b = torch.tensor(g.uniform(size=(B, 28, 28, 1)) < 0.5, dtype=torch.float32, device=dev)
        e = torch.tensor(g.normal(size=(B, 32)), dtype=torch.float32, device=dev)
        ts.set_batch(x, b, e)
        ts.step()
    ts.synchronize()
    calls = ts.reducer.calls_last_step if forced else 0
    return m.params_dict(), calls


def vdvae(forced: bool):
    os.environ["PM_FORCE_DP"] = "1" if forced else "0"
    cfg, B = TINY, 4
    m = PosteriorMatchingVDVAE(**cfg["model"], device="cuda:0", seed=1)
    m.init()
    ts = VDVAETrainStep(m, cfg["lr"], B, gradient_clip=cfg["gradient_clip"], ema_rate=cfg["ema_rate"], seed=1)
    assert (ts.reducer is not None) == forced
    g = np.random.default_rng(2)
    H = cfg["model"]["image_shape"][0]
    for s in range(STEPS):
        x = torch.tensor(np.round(g.uniform(size=(B, H, H, 1)) * 255 * (g.uniform(size=(B, H, H, 1)) < 0.2)), dtype=torch.float32, device=dev)
        b = torch.tensor(g.uniform(size=(B, H, H, 1)) < 0.5, dtype=torch.float32, device=dev)
        ts.set_batch(x, b)
        ts.step()
    ts.synchronize()
    return m.params_dict(), (ts.reducer.calls_last_step if forced else 0)


def flat(d):
    return torch.cat([d[k].double().reshape(-1) for k in sorted(d)])


dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
report = {}
for name, fn in (("pm_vae", pmvae), ("vdvae", vdvae)):
    plain, _ = fn(False)
    plain2, _ = fn(False)
    forced, calls = fn(True)
    # Adam turns gradient components at the noise level into +-lr steps, so single small tensors (a bias whose true gradient is
    # ~0) differ between ANY two runs by the order of the f32 atomics; the yardstick is the whole parameter vector, and the
    # run-to-run floor of the plain path measured here
    floor = rel(flat(plain2), flat(plain))
    got = rel(flat(forced), flat(plain))
    assert got < max(1e-5, 5 * floor), (name, got, floor)
    assert calls >= 1
    report[name] = {"rel_diff_all_params": got, "run_to_run_floor": floor, "allreduce_calls_per_step": calls}
dist.destroy_process_group()
print("RCCL-1RANK-OK", report)
