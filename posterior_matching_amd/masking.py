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


class UniformMaskGenerator(MaskGenerator):
    """q features observed, q uniform (reference masking.py:50-81): q = choice(d) when bounds is None, else
    int(d*lo) + choice(int(d*hi)); the observed set is a uniform subset."""

    def __init__(self, bounds=None, **kw):
        super().__init__(**kw)
        self.bounds = bounds

    def call(self, shape):
        b, d = shape[0], int(np.prod(shape[1:]))
        out = np.zeros((b, d), np.float32)
        for i in range(b):
            if self.bounds is None:
                q = self._rng.choice(d)
            else:
                q = int(d * self.bounds[0]) + self._rng.choice(int(d * self.bounds[1]))
            out[i, self._rng.choice(d, q, replace=False)] = 1
        return out.reshape(shape)


# ---- device-side generators: the same distributions drawn by csrc/pm_mask.hip ---------------
_IMAGE_MIXTURES = {   # name -> (dim, pixel-Bernoulli p, rectangle min_prop, max_prop); reference masking.py:235-286
    "MNISTMaskGenerator": (28, 0.5, 0.3, 1.0),
    "OmniglotMaskGenerator": (28, 0.5, 0.1, 0.6),
    "Cifar10MaskGenerator": (32, 0.3, 0.1, 0.5),
}


class DeviceMaskGenerator:
    """Masks drawn on the GPU from a Philox stream keyed by `seed`; every call advances the stream (`step`).
    `__call__(shape)` returns a float32 device tensor of that shape (images: [B, H, W, 1])."""

    def __init__(self, device, seed: Optional[int] = None, stream_id: int = 0):
        import torch

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("device-side mask generation needs the HIP library and a GPU (no CPU fallback)")
        self.seed = int(np.random.SeedSequence().entropy & (2 ** 63 - 1)) if seed is None else int(seed)
        self.stream_id = stream_id
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _advance(self):
        from . import ops

        ops.counter_increment(self._step)

    def __call__(self, shape, out=None):
        import torch

        shape = tuple(int(s) for s in shape)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        self.fill(out)
        self._advance()
        return out

    def fill(self, out) -> None:
        raise NotImplementedError


class DeviceBernoulliMaskGenerator(DeviceMaskGenerator):
    def __init__(self, p: float = 0.5, **kw):
        super().__init__(**kw)
        self.p = p

    def fill(self, out):
        from . import ops

        ops.bernoulli_mask(out, self.p, self.seed, self._step, self.stream_id)


class DeviceUniformMaskGenerator(DeviceMaskGenerator):
    def __init__(self, bounds=None, **kw):
        super().__init__(**kw)
        self.bounds = bounds

    def fill(self, out):
        from . import ops

        d = out.numel() // out.shape[0]
        lo, span = (0, d) if self.bounds is None else (int(d * self.bounds[0]), int(d * self.bounds[1]))
        ops.uniform_mask(out, lo, span, self.seed, self._step, self.stream_id)


class DeviceImageMixtureMaskGenerator(DeviceMaskGenerator):
    """MNIST / Omniglot / CIFAR-10 mixtures (weights [2,1,1,1,1,2,2]) in one launch per batch."""

    def __init__(self, name: str = "MNISTMaskGenerator", dim: Optional[int] = None, **kw):
        super().__init__(**kw)
        from ._lib import MaskComponent

        d0, p, lo, hi = _IMAGE_MIXTURES[name]
        d = dim if (dim is not None and name == "MNISTMaskGenerator") else d0
        h = d // 2
        spec = [(0, 2, dict(p=p)), (1, 1, dict(y1=0, x1=0, y2=d, x2=h)), (1, 1, dict(y1=0, x1=0, y2=h, x2=d)),
                (1, 1, dict(y1=0, x1=h, y2=d, x2=d)), (1, 1, dict(y1=h, x1=0, y2=d, x2=d)),
                (2, 2, dict(size=h)), (3, 2, dict(min_prop=lo, max_prop=hi))]
        w = np.array([x[1] for x in spec], np.float64)
        cum = np.cumsum(w / w.sum()).astype(np.float32)
        self.comps = (MaskComponent * len(spec))()
        for i, (kind, _, kwargs) in enumerate(spec):
            self.comps[i].kind = kind
            self.comps[i].cum_weight = float(cum[i])
            for k, v in kwargs.items():
                setattr(self.comps[i], k, v)

    def fill(self, out, desc_out=None):
        from . import ops

        if out.dim() != 4 or out.shape[-1] != 1:
            raise AssertionError(f"expected shape of size [batch_dim, height, width, 1], got {tuple(out.shape)}")
        ops.image_mask_mixture(out, self.comps, self.seed, self._step, self.stream_id, desc_out)


_GENERATORS = {"BernoulliMaskGenerator": BernoulliMaskGenerator, "UniformMaskGenerator": UniformMaskGenerator,
               "MNISTMaskGenerator": MNISTMaskGenerator}


def get_mask_generator(mask_generator_name: str, device=None, **kwargs):
    """reference masking.py:328-335 (KeyError for unknown names).  With `device` (a cuda device) the generator draws
    on the GPU; CelebAMaskGenerator (random bicubic pattern, masking.py:177-232) has no device form yet."""
    if device is None:
        return _GENERATORS[mask_generator_name](**kwargs)
    if mask_generator_name == "BernoulliMaskGenerator":
        return DeviceBernoulliMaskGenerator(device=device, **kwargs)
    if mask_generator_name == "UniformMaskGenerator":
        return DeviceUniformMaskGenerator(device=device, **kwargs)
    if mask_generator_name in _IMAGE_MIXTURES:
        return DeviceImageMixtureMaskGenerator(mask_generator_name, device=device, **kwargs)
    raise KeyError(mask_generator_name)
