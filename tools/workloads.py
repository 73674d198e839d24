"""The training steps BASELINE.json's `configs` name, built from configs/*.py exactly as the train scripts build them,
on synthetic batches resident in HBM.  One place for bench.py's `aux.secondary` numbers, the tools/bench_*.py tables and
tests/test_gpu_zz_coverage.py (which kernels does each benchmarked step launch?).

    w = build("pm_vqvae_mnist")            # configs/pm_vqvae_mnist.py at BASELINE's batch (256)
    w.step()                               # one optimizer step on the resident batch
    measure(w, steps=10, warmup=3)         # {"images_per_sec": ..., "ms_per_step": ..., "loss": ..., ...}
"""
import os
import sys
import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Optional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name -> (config file, per-GPU batch BASELINE.json quotes, algorithmic FLOP per training example: SURVEY.md 8(d))
WORKLOADS = {
    "pm_vae_mnist": ("pm_vae_mnist.py", 256, 563.8e6),
    "pm_vae_gas": ("pm_vae_gas.py", 128, 5.26e6),
    "vqvae_mnist": ("vqvae_mnist.py", 256, 31.0e6),
    "pm_vqvae_mnist": ("pm_vqvae_mnist.py", 256, 8.85e9),
    "pm_vdvae_mnist": ("pm_vdvae_mnist.py", 8, 14.91e9),          # global 64 on 8 GPUs; the config's own per-device 16 too
    "pm_vqvae_celeb_a": ("pm_vqvae_celeb_a.py", 16, 69.5e9),      # global 128 on 8 GPUs
}


@dataclass
class Workload:
    name: str
    batch: int
    flops_per_example: float
    ts: Any                                   # the engine's train step object
    feed: Callable[[], None]                  # copies the resident synthetic batch into the step's input buffers
    params: int
    extra: Dict[str, Any] = field(default_factory=dict)

    def step(self) -> None:
        self.ts.step()

    def synchronize(self) -> None:
        self.ts.synchronize()


def _config(fname: str) -> dict:
    from posterior_matching_amd.config_dict import load_config_file

    return load_config_file(os.path.join(ROOT, "configs", fname)).to_dict()


def _pm_vae_opt(cfg):
    from posterior_matching_amd import optim

    return optim.chain(optim.scale_by_adam(**cfg.get("adam", {})), optim.add_decayed_weights(cfg.get("weight_decay", 0.0)),
                       optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))


def build(name: str, batch: Optional[int] = None, device: str = "cuda:0", f32: bool = False, seed: int = 1,
          world_size: int = 1, rank: int = 0, **step_kw) -> Workload:
    import torch

    from posterior_matching_amd import optim
    from posterior_matching_amd.data import data_shape

    fname, B0, flops = WORKLOADS[name]
    B = int(batch or B0)
    cfg = _config(fname)
    gen = torch.Generator().manual_seed(100 + rank)
    dev = torch.device(device)

    def sparse_image(shape, scale=1.0):
        return (torch.rand(shape, generator=gen) * (torch.rand(shape, generator=gen) < 0.19) * scale).to(dev)

    if name in ("pm_vae_mnist", "pm_vae_gas"):
        from posterior_matching_amd.engine import PMVAETrainStep
        from posterior_matching_amd.models import PosteriorMatchingVAE

        xs = data_shape(cfg["data"]["dataset"])
        model = PosteriorMatchingVAE.from_config(cfg["model"], device=dev, seed=seed)
        model.init(xs)
        model.store.use_bf16 = not f32
        ts = PMVAETrainStep(model, cfg, _pm_vae_opt(cfg), B, xs, seed=1234, world_size=world_size, rank=rank, **step_kw)
        if len(xs) == 3:
            x = sparse_image((B,) + xs)
            b = (torch.rand((B,) + xs[:-1] + (1,), generator=gen) < 0.5).float().to(dev)
        else:
            x = torch.randn((B,) + xs, generator=gen).to(dev)
            b = (torch.rand((B,) + xs, generator=gen) < 0.5).float().to(dev)
        return Workload(name, B, flops, ts, lambda: ts.set_batch(x, b), model.num_params)
    if name == "vqvae_mnist":
        from posterior_matching_amd.engine import VQVAETrainStep
        from posterior_matching_amd.models.vqvae import VQVAE

        xs = data_shape(cfg["data"]["dataset"])
        model = VQVAE(**cfg["model"], device=dev, seed=seed)
        model.init(xs)
        model.store.use_bf16 = not f32
        ts = VQVAETrainStep(model, optim.adam(cfg["learning_rate"]), B, xs, world_size=world_size, rank=rank, **step_kw)
        x = sparse_image((B,) + xs)
        return Workload(name, B, flops, ts, lambda: ts.set_batch(x), model.store.num_params)
    if name in ("pm_vqvae_mnist", "pm_vqvae_celeb_a"):
        from posterior_matching_amd.engine import PMVQVAETrainStep
        from posterior_matching_amd.models.pixel_cnn import PixelCNN
        from posterior_matching_amd.models.vqvae import VQVAE, VQVAEPartialEncoder

        vq_cfg = _config("vqvae_mnist.py" if name == "pm_vqvae_mnist" else "vqvae_celeb_a.py")["model"]
        xs = data_shape(cfg["data"]["dataset"])
        vq = VQVAE(**vq_cfg, device=dev, seed=seed)          # stage 1 stands in for the run directory `vqvae_dir` names
        vq.init(xs)
        penc = VQVAEPartialEncoder(cfg["conditional_dim"], vq_cfg)
        pcnn = PixelCNN(**dict(cfg["pixel_cnn"], num_indices=vq_cfg["num_embeddings"]))
        opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                          optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
        ts = PMVQVAETrainStep(vq, penc, pcnn, opt, B, xs, seed=seed, world_size=world_size, rank=rank, **step_kw)
        ts.store.use_bf16 = not f32
        x = torch.rand((B,) + xs, generator=gen).to(dev) if name == "pm_vqvae_celeb_a" else sparse_image((B,) + xs)
        b = (torch.rand((B,) + xs[:2] + (1,), generator=gen) < 0.5).float().to(dev)
        return Workload(name, B, flops, ts, lambda: ts.set_batch(x, b), ts.num_trainable_params)
    if name == "pm_vdvae_mnist":
        from posterior_matching_amd.engine import VDVAETrainStep
        from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE

        model = PosteriorMatchingVDVAE(**cfg["model"], device=dev, seed=seed)
        model.init()
        model.store.use_bf16 = not f32
        ts = VDVAETrainStep(model, cfg["lr"], B, gradient_clip=cfg["gradient_clip"], ema_rate=cfg["ema_rate"], seed=seed,
                            world_size=world_size, rank=rank, **step_kw)
        x = torch.round(sparse_image((B, 28, 28, 1), 255.0))
        b = (torch.rand((B, 28, 28, 1), generator=gen) < 0.5).float().to(dev)
        return Workload(name, B, flops, ts, lambda: ts.set_batch(x, b), model.num_params)
    raise KeyError(name)


def measure(w: Workload, steps: int = 10, warmup: int = 3) -> Dict[str, Any]:
    """whole-step throughput on the resident batch (inputs in HBM when the timed region starts)"""
    import torch

    w.feed()
    for _ in range(warmup):
        w.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    met = w.ts.read_metrics()
    out = {"workload": w.name, "per_gpu_batch": w.batch, "steps": steps, "images_per_sec": round(w.batch * steps / dt, 1),
           "ms_per_step": round(dt / steps * 1e3, 4), "loss": round(float(met["loss"]), 4),
           "whole_step_tflops": round(w.flops_per_example * w.batch / (dt / steps) / 1e12, 2), "params": w.params}
    return out


def kernels_of_one_step(w: Workload) -> set:
    """names (+ variants) of every kernel one eager optimizer step launches"""
    from posterior_matching_amd import ops

    w.feed()
    w.step()                           # allocates every buffer
    w.synchronize()
    if hasattr(w.ts, "invalidate_plan"):
        w.ts.invalidate_plan()
    ops.coverage_begin()
    try:
        w.step()                       # eager again (the plan is recorded on the step after this one)
        w.synchronize()
    finally:
        names = ops.coverage_end()
    return names
