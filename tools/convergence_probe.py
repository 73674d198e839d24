#!/usr/bin/env python3
"""How far apart do 300-step runs of configs/pm_vae_mnist.py end when they differ (a) in arithmetic (bf16x3 vs strict f32),
(b) by a rounding-level perturbation of the initial parameters, (c) in the noise seed?  (tests/test_gpu_convergence.py's
yardstick; prints last-20 validation ELBO / matching-LL means)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tests.test_gpu_convergence as T
from tests.ref_configs import pm_vae_mnist


def main():
    cfg, xs = pm_vae_mnist(), (28, 28, 1)
    data = torch.tensor(T._strokes(T.NDATA + T.B, 3), dtype=torch.float32)
    masks = torch.tensor(T._masks(T.NDATA + T.B, 4), dtype=torch.float32)
    for tag, kw in [("bf16x3", dict(bf16x3=True, noise_seed=21)), ("f32", dict(bf16x3=False, noise_seed=21)),
                    ("f32 again", dict(bf16x3=False, noise_seed=21)),
                    ("f32 init*(1+1e-6)", dict(bf16x3=False, noise_seed=21, perturb=1e-6, pseed=1)),
                    ("f32 init*(1+1e-6) b", dict(bf16x3=False, noise_seed=21, perturb=1e-6, pseed=2)),
                    ("bf16x3 init*(1+1e-6)", dict(bf16x3=True, noise_seed=21, perturb=1e-6, pseed=1)),
                    ("f32 noise 22", dict(bf16x3=False, noise_seed=22)), ("f32 noise 23", dict(bf16x3=False, noise_seed=23))]:
        _, _, val = T._run(cfg, xs, data, masks, **kw)
        print(f"{tag:24s} val_elbo {val[0]:10.4f}  val_matching_ll {val[1]:10.4f}", flush=True)


if __name__ == "__main__":
    main()
