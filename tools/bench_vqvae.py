#!/usr/bin/env python3
"""VQ-VAE (configs/vqvae_mnist.py at BASELINE's batch 256) train-step throughput on one MI355X.
Secondary workload: bench.py stays the headline (PM-VAE MNIST).  Prints one JSON line per mode."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops, optim
from posterior_matching_amd.engine import VQVAETrainStep
from posterior_matching_amd.models.vqvae import VQVAE
from tests.ref_configs import vqvae_mnist

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--table", default=None)
args = ap.parse_args()
cfg, B = vqvae_mnist(), args.batch
gen = torch.Generator().manual_seed(0)
xs = [(torch.rand((B, 28, 28, 1), generator=gen) * (torch.rand((B, 28, 28, 1), generator=gen) < 0.19)).cuda() for _ in range(8)]
for graph in (False, True):
    for bf16 in (True, False):
        m = VQVAE(**cfg["model"], device="cuda:0", seed=1); m.init((28, 28, 1)); m.store.use_bf16 = bf16
        ts = VQVAETrainStep(m, optim.adam(cfg["learning_rate"]), B, (28, 28, 1), use_graph=graph)
        def run(n):
            for i in range(n):
                ts.set_batch(xs[i % 8]); ts.step()
        run(args.warmup); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(args.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        met = ts.read_metrics()
        print(json.dumps({"workload": "vqvae_mnist train step", "batch": B, "launch": "hip_graph" if graph else "eager",
                          "dtype": "bf16x3" if bf16 else "f32", "images_per_sec": round(B * args.steps / dt, 1),
                          "ms_per_step": round(dt / args.steps * 1e3, 4), "loss": round(met["loss"], 3),
                          "perplexity": round(met["perplexity"], 2)}), flush=True)
        if args.table and not graph and bf16:
            timer = ops.KernelTimer(); ops.set_timer(timer)
            for _ in range(3): ts.step()
            ts.synchronize(); ops.set_timer(None)
            summ = timer.summary(); tot = sum(r["ms"] for r in summ.values())
            with open(args.table, "w") as fp:
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                    fp.write(f"{k:48s} calls/step {v['calls'] // 3:3d}  us/step {v['ms'] / 3 * 1e3:8.1f}  "
                             f"TFLOP/s {(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0:7.2f}  GB/s {v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f}\n")
                fp.write(f"sum of kernel time per step: {tot / 3:.4f} ms\n")
                calls = timer.per_call(); per = len(calls) // 3
                for tag, detail, ms, fl in calls[-per:]:
                    fp.write(f"{tag:46s} {detail:50s} {ms * 1e3:8.1f} us\n")
