#!/usr/bin/env python3
"""PM-VQVAE stage-2 train step on one MI355X: --config mnist (configs/pm_vqvae_mnist.py network, BASELINE batch 256) or
--config celeb_a (configs/pm_vqvae_celeb_a.py: 64x64x3, 16x16 codes, K = 512, 12 resnets; BASELINE per-GPU batch 16).
Secondary workload: bench.py stays the headline (PM-VAE MNIST)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops, optim
from posterior_matching_amd.engine import PMVQVAETrainStep
from posterior_matching_amd.models.pixel_cnn import PixelCNN
from posterior_matching_amd.models.vqvae import VQVAE, VQVAEPartialEncoder
from tests.ref_configs import pm_vqvae_celeb_a, pm_vqvae_mnist, vqvae_celeb_a, vqvae_mnist

ap = argparse.ArgumentParser()
ap.add_argument("--config", choices=["mnist", "celeb_a"], default="mnist")
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--f32", action="store_true")
ap.add_argument("--table", default=None)
args = ap.parse_args()
if args.config == "celeb_a":
    cfg, vq_cfg, B, xs = pm_vqvae_celeb_a(), vqvae_celeb_a()["model"], args.batch or 16, (64, 64, 3)
    flops_per_img = 69.5e9      # 2*frozen + 6*trainable MACs per image
else:
    cfg, vq_cfg, B, xs = pm_vqvae_mnist(), vqvae_mnist()["model"], args.batch or 256, (28, 28, 1)
    flops_per_img = 8.85e9
vq = VQVAE(**vq_cfg, device="cuda:0", seed=1); vq.init(xs)
penc = VQVAEPartialEncoder(cfg["conditional_dim"], vq_cfg)
pcnn = PixelCNN(**dict(cfg["pixel_cnn"], num_indices=vq_cfg["num_embeddings"]))
opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                  optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
ts = PMVQVAETrainStep(vq, penc, pcnn, opt, B, xs, seed=1)
ts.store.use_bf16 = not args.f32
gen = torch.Generator().manual_seed(0)
if args.config == "celeb_a":
    x = torch.rand((B,) + xs, generator=gen).cuda()
else:
    x = (torch.rand((B,) + xs, generator=gen) * (torch.rand((B,) + xs, generator=gen) < 0.19)).cuda()
b = (torch.rand((B,) + xs[:2] + (1,), generator=gen) < 0.5).float().cuda()
ts.set_batch(x, b)
for _ in range(args.warmup): ts.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps): ts.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
flops = flops_per_img * B
print(json.dumps({"workload": f"pm_vqvae_{args.config} stage-2 train step", "batch": B, "dtype": "f32" if args.f32 else "bf16x3",
                  "images_per_sec": round(B * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 2),
                  "algorithmic_tflops": round(flops / (dt / args.steps) / 1e12, 2), "loss": round(ts.read_metrics()["loss"], 3),
                  "trainable_params": ts.num_trainable_params, "hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
if args.table:
    timer = ops.KernelTimer(); ops.set_timer(timer)
    ts.step(); ts.synchronize(); ops.set_timer(None)
    summ = timer.summary(); tot = sum(r["ms"] for r in summ.values())
    with open(args.table, "w") as fp:
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            fp.write(f"{k:48s} calls/step {v['calls']:4d}  ms/step {v['ms']:9.3f}  "
                     f"TFLOP/s {(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0:7.2f}  GB/s {v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f}\n")
        fp.write(f"sum of kernel time per step: {tot:.3f} ms\n")
        seen = set()
        for tag, detail, ms, fl in timer.per_call():
            if (tag, detail) in seen: continue
            seen.add((tag, detail))
            fp.write(f"{tag:46s} {detail:52s} {ms * 1e3:9.1f} us  {fl / (ms * 1e-3) / 1e12 if fl else 0:6.2f} TFLOP/s\n")
