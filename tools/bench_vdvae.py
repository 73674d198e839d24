#!/usr/bin/env python3
"""PM-VDVAE train step (configs/pm_vdvae_mnist.py) on one MI355X at the per-GPU batches BASELINE quotes
(8 = global 64 on 8 GPUs) and the reference config (16).  Secondary workload: bench.py stays the headline."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops
from posterior_matching_amd.engine import VDVAETrainStep
from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE
from tests.ref_configs import pm_vdvae_mnist

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="8,16,64")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--f32", action="store_true")
ap.add_argument("--table", default=None)
args = ap.parse_args()
cfg = pm_vdvae_mnist()
for B in [int(v) for v in args.batches.split(",")]:
    m = PosteriorMatchingVDVAE(**cfg["model"], device="cuda:0", seed=1); m.init(); m.store.use_bf16 = not args.f32
    ts = VDVAETrainStep(m, cfg["lr"], B, gradient_clip=cfg["gradient_clip"], ema_rate=cfg["ema_rate"], seed=1)
    gen = torch.Generator().manual_seed(0)
    x = torch.round(torch.rand((B, 28, 28, 1), generator=gen) * 255 * (torch.rand((B, 28, 28, 1), generator=gen) < 0.19)).cuda()
    b = (torch.rand((B, 28, 28, 1), generator=gen) < 0.5).float().cuda()
    ts.set_batch(x, b)
    for _ in range(args.warmup): ts.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): ts.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    met = ts.read_metrics()
    print(json.dumps({"workload": "pm_vdvae_mnist train step", "per_gpu_batch": B, "dtype": "f32" if args.f32 else "bf16x3",
                      "images_per_sec": round(B * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 2),
                      "algorithmic_tflops": round(14.91e9 * B / (dt / args.steps) / 1e12, 2), "loss": round(met["loss"], 2),
                      "bpd": round(met["bpd"], 3), "params": m.num_params}), flush=True)
    if args.table and B == 16:
        ts.use_plan = False; ts.invalidate_plan()  # eager: the timer brackets every C-ABI call (plan replay bypasses it)
        timer = ops.KernelTimer(); ops.set_timer(timer)
        ts.step(); ts.synchronize(); ops.set_timer(None)
        summ = timer.summary(); tot = sum(r["ms"] for r in summ.values())
        with open(args.table, "w") as fp:
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                fp.write(f"{k:48s} calls/step {v['calls']:4d}  ms/step {v['ms']:9.3f}  "
                         f"TFLOP/s {(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0:7.2f}  GB/s {v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f}\n")
            fp.write(f"sum of kernel time per step: {tot:.3f} ms over {sum(v['calls'] for v in summ.values())} launches\n")
            agg = {}
            for tag, detail, ms, fl in timer.per_call():
                a = agg.setdefault((tag, detail), [0, 0.0, 0.0])
                a[0] += 1; a[1] += ms; a[2] += fl
            for (tag, detail), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
                fp.write(f"{tag:44s} {detail:50s} x{n:4d} avg {ms / n * 1e3:7.1f} us  {fl / (ms * 1e-3) / 1e12 if fl else 0:6.2f} TFLOP/s\n")
    del ts, m
