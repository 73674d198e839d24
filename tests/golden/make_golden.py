#!/usr/bin/env python3
"""Generates tests/golden/pm_vae_tiny.npz.

SELF-GENERATED fixture: the reference (JAX/haiku/TFP) cannot run in this environment and ships
no golden vectors, so these are outputs of OUR float64 CPU oracle
(oracle/pm_vae_oracle.py) on seeded NumPy inputs.  They pin the oracle against regressions and
give the GPU tests a file-based target; they are NOT reference-produced ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pm_vae_oracle as O  # noqa: E402

CFG = {
    "model": {
        "latent_dim": 8, "encoder_net": "ConvEncoder", "decoder_net": "ConvDecoder",
        "posterior_dist": "TriLGaussian", "partial_posterior_dist": "AutoregressiveGMM",
        "partial_posterior_dist_config": {"hidden_units": 32, "num_components": 4},
        "decoder_dist": "Bernoulli",
        "encoder_net_config": {"conv_layers": [(4, 5, 1), (8, 5, 2), (16, 7, 1)]},
        "decoder_net_config": {"conv_layers": [(8, 7, 1), (4, 5, 2), (1, 5, 1)]},
    },
    "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5000},
}
XS = (14, 14, 1)
B = 5


def main():
    rng = np.random.default_rng(2024)
    p = O.init_params(CFG["model"], XS, seed=5)
    for t in p.values():                       # non-zero biases so that every path matters
        t.add_(torch.tensor(0.05 * rng.normal(size=tuple(t.shape))).reshape(t.shape))
    x = rng.uniform(size=(B,) + XS) * (rng.uniform(size=(B,) + XS) < 0.3)
    b = (rng.uniform(size=(B,) + XS) < 0.5).astype(np.float64)
    eps = rng.normal(size=(B, CFG["model"]["latent_dim"]))
    xt, bt, et = torch.tensor(x), torch.tensor(b), torch.tensor(eps)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, aux, out = O.pm_vae_loss(leaves, CFG, xt, bt, et, 0)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    arrays = {"x": x, "b": b, "eps": eps, "loss": loss.item(),
              "reconstruction_ll": out["reconstruction_ll"].detach().numpy(), "kl": out["kl"].detach().numpy(),
              "matching_ll": out["matching_ll"].detach().numpy()}
    for (k, v), g in zip(p.items(), grads):
        arrays["param/" + k] = v.numpy()
        arrays["grad/" + k] = g.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pm_vae_tiny.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
