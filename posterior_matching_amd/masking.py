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


def _check_image_shape(shape):
    if len(shape) != 4:
        raise AssertionError(f"expected shape of size [batch_dim, height, width, channels], got {tuple(shape)}")


def _child_seed(rng: np.random.Generator) -> int:
    return int(rng.integers(0, 2 ** 63 - 1))


class ImageBernoulliMaskGenerator(MaskGenerator):
    """each PIXEL observed with probability p, one mask channel (reference masking.py:94-104)."""

    def __init__(self, p: float = 0.2, **kw):
        super().__init__(**kw)
        self.p = p

    def call(self, shape):
        _check_image_shape(shape)
        return self._rng.binomial(1, self.p, size=tuple(shape[:-1]) + (1,))


class RectangleMaskGenerator(MaskGenerator):
    """a random axis-aligned rectangle covering min_prop..max_prop of the image is missing; corner pairs are
    redrawn until the area bound holds (reference masking.py:107-140)."""

    def __init__(self, min_prop: float = 0.3, max_prop: float = 1.0, **kw):
        super().__init__(**kw)
        self.min_prop, self.max_prop = min_prop, max_prop

    def call(self, shape):
        _check_image_shape(shape)
        bsz, h, w, _ = shape
        out = np.ones((bsz, h, w, 1), np.float32)
        for i in range(bsz):
            while True:
                x1, x2 = np.sort(self._rng.integers(0, w, 2))
                y1, y2 = np.sort(self._rng.integers(0, h, 2))
                if self.min_prop * w * h <= (x2 - x1 + 1) * (y2 - y1 + 1) <= self.max_prop * w * h:
                    break
            out[i, y1:y2 + 1, x1:x2 + 1] = 0
        return out


class FixedRectangleMaskGenerator(MaskGenerator):
    """rows y1:y2, columns x1:x2 missing (reference masking.py:143-157)."""

    def __init__(self, y1, x1, y2, x2, **kw):
        super().__init__(**kw)
        self.y1, self.x1, self.y2, self.x2 = y1, x1, y2, x2

    def call(self, shape):
        _check_image_shape(shape)
        out = np.ones(tuple(shape[:-1]) + (1,), np.float32)
        out[:, self.y1:self.y2, self.x1:self.x2] = 0
        return out


class SquareMaskGenerator(MaskGenerator):
    """one random size x size square missing, the same square for every example of the call (reference
    masking.py:160-174; the per-example mixtures call it with batch size 1)."""

    def __init__(self, size, **kw):
        super().__init__(**kw)
        self.size = size

    def call(self, shape):
        _check_image_shape(shape)
        _, h, w, _ = shape
        out = np.ones(tuple(shape[:-1]) + (1,), np.float32)
        x0, y0 = self._rng.integers(w - self.size), self._rng.integers(h - self.size)
        out[:, y0:y0 + self.size, x0:x0 + self.size] = 0
        return out


def _bicubic_taps(out_coords: np.ndarray, in_size: int, scale: float):
    """Pillow's resampling coefficients (Resample.c precompute_coeffs with the bicubic filter, a = -0.5, support 2)
    for the given OUTPUT coordinates of an in_size -> in_size/scale upscale: -> (first tap index [n], weights [n, 5]
    float64, zero-padded).  Upscaling leaves the filter unscaled, so at most 5 input samples meet an output pixel."""
    centre = (out_coords.astype(np.float64) + 0.5) * scale
    lo = np.maximum((centre - 2.0 + 0.5).astype(np.int64), 0)
    hi = np.minimum((centre + 2.0 + 0.5).astype(np.int64), in_size)
    j = np.arange(5)[None, :]
    t = np.abs((j + lo[:, None]) - centre[:, None] + 0.5)
    a = -0.5
    near = ((a + 2.0) * t - (a + 3.0)) * t * t + 1.0
    far = (((t - 5.0) * t + 8.0) * t - 4.0) * a
    w = np.where(t < 1.0, near, np.where(t < 2.0, far, 0.0))
    w = np.where(j < (hi - lo)[:, None], w, 0.0)
    tot = np.zeros(len(centre))
    for k in range(5):                       # Pillow sums the taps left to right in double precision
        tot = tot + w[:, k]
    return lo, w / np.where(tot != 0.0, tot, 1.0)[:, None]


def bicubic_window(low: np.ndarray, y0: int, x0: int, height: int, width: int, out_size: int) -> np.ndarray:
    """Rows y0:y0+height, columns x0:x0+width of PIL.Image.fromarray(low).resize((out_size, out_size), BICUBIC) for a
    float32 `low`, computed WITHOUT materialising the out_size x out_size image: horizontal pass (float64 sums,
    rounded to float32) over the few low-resolution rows the window touches, then the vertical pass - Pillow's order
    of operations, so the result is bit-identical to the crop of the full resize (tests/test_host_cpu.py)."""
    scale = low.shape[0] / out_size
    ylo, yw = _bicubic_taps(np.arange(y0, y0 + height), low.shape[0], scale)
    xlo, xw = _bicubic_taps(np.arange(x0, x0 + width), low.shape[1], low.shape[1] / out_size)
    r0, r1 = int(ylo.min()), int(min(ylo.max() + 5, low.shape[0]))
    rows = low[r0:r1].astype(np.float64)
    hor = np.zeros((r1 - r0, width))
    for k in range(5):
        cols = np.minimum(xlo + k, low.shape[1] - 1)
        hor = hor + rows[:, cols] * xw[None, :, k]
    hor = hor.astype(np.float32).astype(np.float64)
    out = np.zeros((height, width))
    for k in range(5):
        rr = np.minimum(ylo + k, low.shape[0] - 1) - r0
        out = out + hor[rr, :] * yw[:, k, None]
    return out.astype(np.float32)


class RandomPatternMaskGenerator(MaskGenerator):
    """Blob-shaped missing regions (reference masking.py:177-232): a resolution*max_size square of uniform noise is
    upsampled bicubically to max_size x max_size and thresholded at `density`; every mask is a random window of it whose
    covered fraction lies within density +- density_std; the noise is redrawn after update_freq * max_size^2 pixels
    have been handed out.  The reference materialises the 10 000 x 10 000 pattern (400 MB) with PIL; here only the
    requested window is interpolated (bicubic_window: the same arithmetic, bit-identical to a crop of PIL's resize)."""

    def __init__(self, max_size=10000, resolution=0.06, density=0.25, update_freq=1, **kw):
        super().__init__(**kw)
        self.max_size, self.resolution, self.density, self.update_freq = max_size, resolution, density, update_freq
        self._regenerate_cache()

    def _regenerate_cache(self):
        n = int(self.resolution * self.max_size)
        self.low_pattern = self._rng.uniform(0.0, 1.0, size=(n, n)).astype(np.float32)
        self.points_used = 0

    def window(self, y0, x0, height, width):
        """pattern[y0:y0+height, x0:x0+width] (1 = inside a blob)"""
        return (bicubic_window(self.low_pattern, y0, x0, height, width, self.max_size) < self.density).astype(np.float32)

    def call(self, shape, density_std=0.05):
        _check_image_shape(shape)
        bsz, h, w, _ = shape
        out = np.empty((bsz, h, w, 1), np.float32)
        for i in range(bsz):
            while True:
                x0 = int(self._rng.integers(0, self.max_size - w + 1))
                y0 = int(self._rng.integers(0, self.max_size - h + 1))
                res = self.window(y0, x0, h, w)
                if self.density - density_std < res.mean() < self.density + density_std:
                    break
            out[i, :, :, 0] = 1.0 - res
            self.points_used += w * h
            if self.update_freq * self.max_size ** 2 < self.points_used:
                self._regenerate_cache()
        return out


class MixtureMaskGenerator(MaskGenerator):
    """one sub-generator per example (or per batch with batch_level), drawn with the given weights (reference
    masking.py:24-47)."""

    def __init__(self, generators, weights=None, batch_level=False, **kw):
        super().__init__(**kw)
        self.generators = list(generators)
        w = np.ones(len(self.generators)) if weights is None else np.asarray(weights, np.float64)
        assert len(w) == len(self.generators)
        self.weights, self.batch_level = w / w.sum(), batch_level

    def call(self, shape):
        if self.batch_level:
            return self.generators[int(self._rng.choice(len(self.generators), p=self.weights))](shape)
        picks = self._rng.choice(len(self.generators), size=shape[0], p=self.weights)
        return np.concatenate([self.generators[i]((1,) + tuple(shape[1:])) for i in picks], axis=0)


def _half_plane_mixture(dim, pixel_p, rect_props, rng):
    half = dim // 2
    s = lambda: _child_seed(rng)   # noqa: E731
    return [ImageBernoulliMaskGenerator(pixel_p, seed=s()),
            FixedRectangleMaskGenerator(0, 0, dim, half), FixedRectangleMaskGenerator(0, 0, half, dim),
            FixedRectangleMaskGenerator(0, half, dim, dim), FixedRectangleMaskGenerator(half, 0, dim, dim),
            SquareMaskGenerator(half, seed=s()), RectangleMaskGenerator(*rect_props, seed=s())], [2, 1, 1, 1, 1, 2, 2]


class MNISTMaskGenerator(MixtureMaskGenerator):
    """Per-example mixture, weights [2,1,1,1,1,2,2] (reference masking.py:235-249): pixel-Bernoulli(0.5);
    left / top / right / bottom half missing; a random dim/2 square missing; a random rectangle
    covering 30-100 % of the image missing.  Shape [B, H, W, 1]."""

    def __init__(self, dim: int = 28, seed=None, **kw):
        gens, w = _half_plane_mixture(dim, 0.5, (0.3, 1.0), np.random.default_rng(seed))
        super().__init__(gens, weights=w, seed=seed, **kw)
        self.dim = dim

    def call(self, shape):
        _check_image_shape(shape)
        return super().call(shape)


class OmniglotMaskGenerator(MixtureMaskGenerator):
    """reference masking.py:252-267"""

    def __init__(self, seed=None, **kw):
        gens, w = _half_plane_mixture(28, 0.5, (0.1, 0.6), np.random.default_rng(seed))
        super().__init__(gens, weights=w, seed=seed, **kw)


class Cifar10MaskGenerator(MixtureMaskGenerator):
    """reference masking.py:270-286"""

    def __init__(self, seed=None, **kw):
        gens, w = _half_plane_mixture(32, 0.3, (0.1, 0.5), np.random.default_rng(seed))
        super().__init__(gens, weights=w, seed=seed, **kw)


_GCF_RECTS = [(26, 17, 58, 36), (26, 29, 58, 48), (26, 15, 37, 50), (26, 15, 37, 34), (26, 31, 37, 50), (43, 20, 62, 44)]
_SIIDGM_RECTS = [(16, 16, 48, 48), (0, 0, 64, 32), (0, 0, 32, 64), (0, 32, 64, 64), (32, 0, 64, 64)]


class GCFMaskGenerator(MixtureMaskGenerator):
    """six fixed face-region rectangles of a 64 x 64 image, equal weights (reference masking.py:289-300)"""

    def __init__(self, seed=None, **kw):
        super().__init__([FixedRectangleMaskGenerator(*r) for r in _GCF_RECTS], seed=seed, **kw)


class SIIDGMMaskGenerator(MixtureMaskGenerator):
    """random pattern / pixel-Bernoulli(0.2) / centre square / four half planes of a 64 x 64 image, weights
    [2,2,2,1,1,1,1] (reference masking.py:303-314)"""

    def __init__(self, seed=None, max_size=10000, resolution=0.06, **kw):
        rng = np.random.default_rng(seed)
        gens = [RandomPatternMaskGenerator(max_size=max_size, resolution=resolution, seed=_child_seed(rng)),
                ImageBernoulliMaskGenerator(0.2, seed=_child_seed(rng))]
        gens += [FixedRectangleMaskGenerator(*r) for r in _SIIDGM_RECTS]
        super().__init__(gens, weights=[2, 2, 2, 1, 1, 1, 1], seed=seed, **kw)


class CelebAMaskGenerator(MixtureMaskGenerator):
    """SIIDGM / GCF / random rectangle (30-100 %) with weights [1,1,2] (reference masking.py:317-325)"""

    def __init__(self, seed=None, **kw):
        rng = np.random.default_rng(seed)
        gens = [SIIDGMMaskGenerator(seed=_child_seed(rng)), GCFMaskGenerator(seed=_child_seed(rng)),
                RectangleMaskGenerator(seed=_child_seed(rng))]
        super().__init__(gens, weights=[1, 1, 2], seed=seed, **kw)


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


def _half_plane_spec(d, p, lo, hi):
    """(kind, weight, fields) of the MNIST / Omniglot / CIFAR-10 mixtures (reference masking.py:235-286)"""
    h = d // 2
    return [(0, 2, dict(p=p)), (1, 1, dict(y1=0, x1=0, y2=d, x2=h)), (1, 1, dict(y1=0, x1=0, y2=h, x2=d)),
            (1, 1, dict(y1=0, x1=h, y2=d, x2=d)), (1, 1, dict(y1=h, x1=0, y2=d, x2=d)),
            (2, 2, dict(size=h)), (3, 2, dict(min_prop=lo, max_prop=hi))]


def _celeba_spec(max_size=10000, resolution=0.06, density=0.25, density_std=0.05):
    """CelebAMaskGenerator (masking.py:317-325) flattened: the nested mixtures SIIDGM (weight 1: random pattern,
    pixel-Bernoulli(0.2), five fixed rectangles, weights [2,2,2,1,1,1,1]), GCF (weight 1: six fixed rectangles) and the
    random rectangle (weight 2) become ONE 14-component mixture with the product weights."""
    spec = [(4, 0.25 * 2 / 10, dict(size=max_size, y1=int(resolution * max_size), p=density, min_prop=density_std)),
            (0, 0.25 * 2 / 10, dict(p=0.2))]
    spec += [(1, 0.25 * (2 if i == 0 else 1) / 10, dict(y1=r[0], x1=r[1], y2=r[2], x2=r[3])) for i, r in enumerate(_SIIDGM_RECTS)]
    spec += [(1, 0.25 / 6, dict(y1=r[0], x1=r[1], y2=r[2], x2=r[3])) for r in _GCF_RECTS]
    spec += [(3, 0.5, dict(min_prop=0.3, max_prop=1.0))]
    return spec


class DeviceImageMixtureMaskGenerator(DeviceMaskGenerator):
    """MNIST / Omniglot / CIFAR-10 (weights [2,1,1,1,1,2,2]) and CelebA (14 flattened components, one of them the random
    blob pattern) mixtures in one launch per batch."""

    def __init__(self, name: str = "MNISTMaskGenerator", dim: Optional[int] = None, max_size: int = 10000,
                 resolution: float = 0.06, update_freq: float = 1, **kw):
        super().__init__(**kw)
        import torch

        from ._lib import MaskComponent

        self.pattern_state, self.pattern_refresh = None, 0
        if name == "CelebAMaskGenerator":
            spec = _celeba_spec(max_size, resolution)
            self.pattern_state = torch.zeros(2, dtype=torch.int64, device=self.device)     # noise epoch, pixels handed out
            self.pattern_refresh = int(update_freq * max_size ** 2)                       # masking.py:226-228
        else:
            d0, p, lo, hi = _IMAGE_MIXTURES[name]
            spec = _half_plane_spec(dim if (dim is not None and name == "MNISTMaskGenerator") else d0, p, lo, hi)
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
        ops.image_mask_mixture(out, self.comps, self.seed, self._step, self.stream_id, desc_out, self.pattern_state,
                               self.pattern_refresh)


_GENERATORS = {"BernoulliMaskGenerator": BernoulliMaskGenerator, "UniformMaskGenerator": UniformMaskGenerator,
               "MNISTMaskGenerator": MNISTMaskGenerator, "OmniglotMaskGenerator": OmniglotMaskGenerator,
               "CelebAMaskGenerator": CelebAMaskGenerator}


def get_mask_generator(mask_generator_name: str, device=None, **kwargs):
    """reference masking.py:328-335 (KeyError for unknown names).  With `device` (a cuda device) the generator draws
    on the GPU (CelebAMaskGenerator included: its random bicubic pattern, masking.py:177-232, is interpolated per window)."""
    if device is None:
        return _GENERATORS[mask_generator_name](**kwargs)
    if mask_generator_name == "BernoulliMaskGenerator":
        return DeviceBernoulliMaskGenerator(device=device, **kwargs)
    if mask_generator_name == "UniformMaskGenerator":
        return DeviceUniformMaskGenerator(device=device, **kwargs)
    if mask_generator_name in _IMAGE_MIXTURES or mask_generator_name == "CelebAMaskGenerator":
        return DeviceImageMixtureMaskGenerator(mask_generator_name, device=device, **kwargs)
    raise KeyError(mask_generator_name)
