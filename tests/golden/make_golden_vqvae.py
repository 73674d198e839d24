#!/usr/bin/env python3
"""Generates tests/golden/vqvae_tiny.npz.

SELF-GENERATED fixture (see make_golden.py): outputs of OUR float64 CPU oracle
(oracle/vqvae_oracle.py) for one is_training=True call of a small VQ-VAE on seeded inputs - loss
terms, code indices, gradients and the haiku state after the EMA update.  NOT reference-produced.

    python tests/golden/make_golden_vqvae.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import vqvae_oracle as VO  # noqa: E402

CFG = {"model": {"embedding_dim": 32, "num_embeddings": 24, "hidden_units": 32, "residual_hidden_units": 32,
                 "residual_blocks": 1, "decay": 0.9, "use_ema": True, "commitment_cost": 0.25, "output_channels": 1},
       "learning_rate": 3e-4}
XS = (12, 12, 1)
B = 4


def main():
    rng = np.random.default_rng(77)
    p = VO.init_params(CFG["model"], 1, seed=5, dtype=torch.float64)
    for t in p.values():
        t.add_(torch.tensor(0.05 * rng.normal(size=tuple(t.shape))).reshape(t.shape))
    st = VO.init_state(CFG["model"], seed=6)
    st["vq/ema_cluster_size/hidden"] = torch.tensor(rng.uniform(size=(24,)))
    st["vq/ema_dw/hidden"] = torch.tensor(rng.normal(size=(32, 24)))
    st["vq/ema_cluster_size/counter"] = st["vq/ema_dw/counter"] = torch.tensor(3)
    x = rng.uniform(size=(B,) + XS) * (rng.uniform(size=(B,) + XS) < 0.3)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, aux, out, new_state = VO.vqvae_loss(leaves, st, CFG, torch.tensor(x), True)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    arrays = {"x": x, "loss": loss.item(), "perplexity": aux["perplexity"].item(),
              "reconstruction_loss": aux["reconstruction_loss"].item(), "vq_loss": aux["vq_loss"].item(),
              "encoding_indices": out["vq_output"]["encoding_indices"].numpy().astype(np.int32),
              "reconstruction": out["reconstruction"].detach().numpy()}
    for (k, v), g in zip(p.items(), grads):
        arrays["param/" + k] = v.numpy()
        arrays["grad/" + k] = g.numpy()
    for k, v in st.items():
        arrays["state/" + k] = v.numpy()
    for k, v in new_state.items():
        arrays["new_state/" + k] = v.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vqvae_tiny.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
