"""Observed-feature masks (1 = observed, 0 = missing): the input contract of the hot path.

Host-side NumPy generators with the mixture structure of the reference's masking.py
(MNISTMaskGenerator :235-249, BernoulliMaskGenerator :84-91, registry :328-335).  They produce
statistically equivalent masks, not the reference's bit streams: the reference's sub-generators
own un-seeded RandomStates (masking.py:13,238-246), so its masks are not reproducible either.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np


class MaskGenerator:
    def __init__(self, seed: Optional[int] = None, dtype=np.float32):
        self._rng = np.random.default_rng(seed)
        self._dtype = dtype

    def __call__(self, shape: Sequence[int]) -> np.ndarray:
        return self.call(tuple(int(s) for s in shape)).astype(self._dtype)

    def call(self, shape):
        raise NotImplementedError


class BernoulliMaskGenerator(MaskGenerator):
    """each feature observed independently with probability p (reference masking.py:84-91)."""

    def __init__(self, p: float = 0.5, **kw):
        super().__init__(**kw)
        self.p = p

    def call(self, shape):
        return self._rng.binomial(1, self.p, size=shape)


class MNISTMaskGenerator(MaskGenerator):
    """Per-example mixture, weights [2,1,1,1,1,2,2] (reference masking.py:235-249): pixel-Bernoulli(0.5);
    top / left / bottom / right half missing; a random dim/2 square missing; a random rectangle
    covering 30-100 % of the image missing.  Shape [B, H, W, 1]."""

    def __init__(self, dim: int = 28, **kw):
        super().__init__(**kw)
        self.dim = dim
        self.weights = np.array([2, 1, 1, 1, 1, 2, 2], np.float64) / 10.0

    def call(self, shape):
        if len(shape) != 4:
            raise AssertionError(f"expected shape of size [batch_dim, height, width, channels], got {shape}")
        bsz, h, w, _ = shape
        half = self.dim // 2
        kinds = self._rng.choice(7, size=bsz, p=self.weights)
        out = np.ones((bsz, h, w, 1), np.float32)
        for i, kind in enumerate(kinds):
            m = out[i, :, :, 0]
            if kind == 0:
                m[...] = self._rng.binomial(1, 0.5, size=(h, w))
            elif kind == 1:
                m[0:self.dim, 0:half] = 0           # FixedRectangle(y1=0, x1=0, y2=dim, x2=half)
            elif kind == 2:
                m[0:half, 0:self.dim] = 0
            elif kind == 3:
                m[0:self.dim, half:self.dim] = 0
            elif kind == 4:
                m[half:self.dim, 0:self.dim] = 0
            elif kind == 5:
                x0, y0 = self._rng.integers(w - half), self._rng.integers(h - half)
                m[y0:y0 + half, x0:x0 + half] = 0
            else:
                while True:
                    x1, x2 = sorted(self._rng.integers(0, w, 2))
                    y1, y2 = sorted(self._rng.integers(0, h, 2))
                    if 0.3 * w * h <= (x2 - x1 + 1) * (y2 - y1 + 1) <= 1.0 * w * h:
                        break
                m[y1:y2 + 1, x1:x2 + 1] = 0
        return out


_GENERATORS = {"BernoulliMaskGenerator": BernoulliMaskGenerator, "MNISTMaskGenerator": MNISTMaskGenerator}


def get_mask_generator(mask_generator_name: str, **kwargs) -> MaskGenerator:
    """reference masking.py:328-335 (KeyError for generators outside the hot-path configs)."""
    return _GENERATORS[mask_generator_name](**kwargs)
