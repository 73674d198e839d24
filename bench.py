#!/usr/bin/env python3
"""Headline benchmark: PM-VAE MNIST training images/sec (BASELINE.json metric), per-GPU batch 256.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full optimizer step of train_pm_vae.py on configs/pm_vae_mnist.py: eps sampling,
masked-encoder + encoder + decoder forward, ELBO + posterior-matching loss, full backward, Adam.
Inputs are synthetic batches resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(cfg, xs, B, budget_s=12.0):
    """The oracle's train step (torch CPU, float32) on the host cores: a reported baseline."""
    import torch

    from oracle import pm_vae_oracle as O

    p = O.init_params(cfg["model"], xs, seed=1, dtype=torch.float32)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    gen = torch.Generator().manual_seed(0)
    x = torch.rand((B,) + xs, generator=gen) * (torch.rand((B,) + xs, generator=gen) < 0.19)
    b = (torch.rand((B,) + xs, generator=gen) < 0.5).float()
    eps = torch.randn((B, cfg["model"]["latent_dim"]), generator=gen)
    O.train_step(p, m, v, cfg, x, b, eps, 0)                       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(p, m, v, cfg, x, b, eps, n + 1)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 20:
            break
    return {"value": round(n * B / dt, 1), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} optimizer steps of the torch-CPU float32 oracle at batch {B} ({dt:.1f} s); the JAX "
                      "reference itself cannot run here (jax/haiku/tfp absent)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--graph", action="store_true", help="replay one captured HIP graph per step (single stream) "
                                                        "instead of eager launches on two overlapping streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="eager steps with per-kernel HIP events")
    ap.add_argument("--f32", action="store_true", help="every GEMM on the f32 MFMA (strict-parity mode) instead of the "
                                                      "default bf16x3 split on the bf16 matrix cores")
    ap.add_argument("--wgrad-streams", action="store_true", help="weight-gradient kernels on companion streams")
    ap.add_argument("--serial", action="store_true", help="one stream (clean per-kernel durations for rocprofv3)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "the N>1 path with several ranks on ONE GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from posterior_matching_amd import ops, optim
    from posterior_matching_amd.data import SyntheticDataset
    from posterior_matching_amd.engine import PMVAETrainStep
    from posterior_matching_amd.models import PosteriorMatchingVAE
    from tests.ref_configs import pm_vae_mnist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)        # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    cfg, xs, B = pm_vae_mnist(), (28, 28, 1), args.batch
    model = PosteriorMatchingVAE.from_config(cfg["model"], device=dev, seed=1)    # same init on every rank
    model.init(xs)
    model.store.use_bf16 = not args.f32
    model.concurrent = not args.serial
    model.ws.overlap_wgrad = args.wgrad_streams
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg.get("weight_decay", 0.0)),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(model, cfg, opt, B, xs, seed=1234, world_size=world, rank=rank, use_graph=args.graph)
    pool = SyntheticDataset(cfg["data"], B, num_batches=16, seed=100 + rank, device=dev)
    batches = pool.batches

    def run(n, offset=0):
        for i in range(n):
            bt = batches[(offset + i) % len(batches)]
            ts.set_batch(bt["image"], bt["mask"])          # device-to-device copy of the resident batch
            ts.step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    metrics = ts.read_metrics()

    # live per-kernel timing with HIP events on the launching stream (eager, outside the timed region)
    roofline = None
    if rank == 0 and args.profile_steps > 0:
        model.concurrent = False        # per-kernel times: one stream, so durations do not overlap
        prof = PMVAETrainStep(model, cfg, opt, B, xs, seed=1234, world_size=1, rank=0, use_graph=False)
        prof.set_batch(batches[0]["image"], batches[0]["mask"])
        prof.step()
        prof.synchronize()
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        for _ in range(args.profile_steps):
            prof.step()
        prof.synchronize()
        ops.set_timer(None)
        summ = timer.summary()
        total_ms = sum(r["ms"] for r in summ.values())
        name, r = max(summ.items(), key=lambda kv: kv[1]["ms"])
        avg_s = r["ms"] / r["calls"] * 1e-3
        flops_per_launch = r["flops"] / r["calls"]
        achieved = flops_per_launch / avg_s / 1e12
        is_bf16 = "bf16" in name
        peak = BF16_MFMA_PEAK_TFLOPS if is_bf16 else F32_MFMA_PEAK_TFLOPS
        # HBM-side bytes per launch of this kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, with the
        # gfx950 2x correction for wide loads) of the same command, committed under profiles/ (counters cannot be read
        # from inside the process)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")) as fp:
                traffic = json.load(fp).get(name, {}).get("traffic_bytes")
        except OSError:
            pass
        roofline = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "note": ("achieved = ALGORITHMIC (float32-equivalent) flops / time; this kernel issues 3 bf16 MFMA "
                             "products per algorithmic MAC (bf16x3 split), so the matrix pipe runs at 3x this rate; "
                             "against the f32 MFMA peak of 157.3 TFLOP/s the fraction is "
                             f"{achieved / F32_MFMA_PEAK_TFLOPS:.3f}") if is_bf16 else
                            "f32 MFMA (v_mfma_f32_32x32x2_f32) runs on the VALU pipeline on gfx950",
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches_per_step": r["calls"] // args.profile_steps,
                    "share_of_kernel_time": round(r["ms"] / total_ms, 3),
                    "algorithmic_gflop_per_launch": round(flops_per_launch / 1e9, 3),
                    "algorithmic_bytes_per_launch": round(r["bytes"] / r["calls"]),
                    "traffic_source": "profiles/r01_final_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE per launch)"}
        if os.environ.get("PM_BENCH_KERNEL_TABLE"):
            rows = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])
            with open(os.environ["PM_BENCH_KERNEL_TABLE"], "w") as fp:
                for k_, v_ in rows:
                    fp.write(f"{k_:45s} calls/step {v_['calls'] // args.profile_steps:3d}  "
                             f"ms/step {v_['ms'] / args.profile_steps:8.4f}  "
                             f"TFLOP/s {(v_['flops'] / (v_['ms'] * 1e-3) / 1e12) if v_['flops'] else 0:7.2f}  "
                             f"GB/s {(v_['bytes'] / (v_['ms'] * 1e-3) / 1e9):8.1f}\n")
                fp.write(f"sum of kernel time per step: {total_ms / args.profile_steps:.4f} ms\n\n")
                calls = timer.per_call()
                per_step = len(calls) // args.profile_steps
                fp.write("launch order of the last profiled step:\n")
                for tag, detail, ms, fl in calls[-per_step:]:
                    fp.write(f"{tag:42s} {detail:52s} {ms * 1e3:9.1f} us  {fl / (ms * 1e-3) / 1e12 if fl else 0:6.2f} TFLOP/s\n")

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            "metric": "training images/sec, PM-VAE MNIST bs=256 per GPU (ELBO / PM matching-LL in `aux`)",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3" if model.store.use_bf16 else "f32", "data": "synthetic",
            "config": {"workload": "configs/pm_vae_mnist.py: conv PM-VAE 28x28x1, latent 32, TriL posterior, "
                                   "AR-GMM partial posterior, Bernoulli decoder; full train step (fwd+loss+bwd+Adam)",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "gemm_arithmetic": ("bf16x3: operands split hi+lo bf16, 3 bf16-MFMA products, f32 accumulate "
                                           "(fwd/dgrad); f32 MFMA (wgrad)") if model.store.use_bf16 else "f32 MFMA",
                       "launch": "hip_graph_1stream" if args.graph else "eager_2streams", "params": model.num_params},
            "aux": {"elbo": round(metrics["reconstruction_ll"] - metrics["beta"] * metrics["kl"], 4),
                    "matching_ll": round(metrics["matching_ll"], 4), "kl": round(metrics["kl"], 4),
                    "loss": round(metrics["loss"], 4)},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, xs, B)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
