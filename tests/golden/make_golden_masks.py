#!/usr/bin/env python3
"""Generates tests/golden/masks_tiny.npz.

SELF-GENERATED fixture: the bit streams oracle/masking_oracle.py defines for the device-side mask kernels (csrc/pm_mask.hip),
stored bit-packed.  The only externally pinned piece is Philox4x32-10 itself (Random123 known-answer vectors,
tests/test_oracle_kat.py); the reference's own generators are un-seeded NumPy RandomStates and cannot be reproduced.

    python tests/golden/make_golden_masks.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import masking_oracle as MO  # noqa: E402

CASES = {"mnist": ("MNISTMaskGenerator", 24, 28, 7), "omniglot": ("OmniglotMaskGenerator", 12, 28, 8),
         "cifar10": ("Cifar10MaskGenerator", 10, 32, 9)}


def build():
    out = {}
    for key, (name, B, H, seed) in CASES.items():
        comps = MO.image_mixture_components(name)
        for step in (0, 5):
            m, d = MO.image_mask_mixture(B, H, H, comps, seed, step=step)
            out[f"{key}_step{step}_mask"] = np.packbits(m.astype(np.uint8).reshape(-1))
            out[f"{key}_step{step}_desc"] = d
    out["bernoulli_p03"] = np.packbits(MO.bernoulli_mask((9, 43), 0.3, seed=3, step=2).astype(np.uint8).reshape(-1))
    out["uniform_d21"] = np.packbits(MO.uniform_mask(16, 21, 0, 21, seed=4, step=1).astype(np.uint8).reshape(-1))
    out["uniform_bounds"] = np.packbits(MO.uniform_mask(8, 300, 75, 150, seed=5, step=0).astype(np.uint8).reshape(-1))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "masks_tiny.npz"), **build())
    print("written")
