"""TEST INFRASTRUCTURE ONLY - CPU restatement of the device-side mask generators (csrc/pm_mask.hip).

PARITY UNPINNED with respect to the reference: its generators (posterior_matching/masking.py) own un-seeded
NumPy RandomStates (masking.py:13 and the sub-generators built without a seed at :238-246), so no bit stream of
the reference can be reproduced.  What this file pins is (a) the Philox4x32-10 block function, against the
Random123 known-answer vectors, (b) the exact bit streams OUR kernels must produce (the GPU tests compare
bit-for-bit), and (c) the distribution each reference generator defines, restated here line by line:

  mixture, per example        masking.py:39-47   inds = choice(len(generators), B, p=weights); one mask per example
  ImageBernoulli(p)           masking.py:94-104  binomial(1, p) per pixel, one channel
  FixedRectangle(y1,x1,y2,x2) masking.py:143-157 mask[y1:y2, x1:x2] = 0
  Square(size)                masking.py:160-174 x = randint(W - size), y = randint(H - size); mask[y:y+size, x:x+size] = 0
  Rectangle(min, max)         masking.py:107-140 x1,x2 = sorted(randint(0,W,2)); y likewise; redraw until
                                                 min*W*H <= (x2-x1+1)(y2-y1+1) <= max*W*H; mask[y1:y2+1, x1:x2+1] = 0
  Bernoulli(p)                masking.py:84-91   binomial(1, p, size=shape)
  Uniform(bounds)             masking.py:50-81   q = choice(d) (bounds None) or int(d*lo) + choice(int(d*hi));
                                                 q features chosen without replacement are observed
  MNIST / Omniglot / CIFAR-10 masking.py:235-286 weights [2,1,1,1,1,2,2] over pixel-Bernoulli, the four half planes,
                                                 a dim/2 square and an area-bounded rectangle

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

PIXEL_BERNOULLI, FIXED_RECT, SQUARE, RECT = 0, 1, 2, 3
DESC_TAG = 0x80000000
RECT_MAX_TRIES = 256
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """Philox4x32-10 (Salmon et al., SC'11; Random123 philox.h).  counter: uint32 [..., 4]; returns uint32 [..., 4]."""
    c = np.asarray(counter, dtype=np.uint32).astype(np.uint64)
    c0, c1, c2, c3 = (c[..., 0].copy(), c[..., 1].copy(), c[..., 2].copy(), c[..., 3].copy())
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c1 = p1 & _MASK32
        c3 = p0 & _MASK32
        c0, c2 = n0, n2
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _key(seed: int) -> Tuple[int, int]:
    seed &= (1 << 64) - 1
    return seed & 0xFFFFFFFF, seed >> 32


def _rand_below(r: np.ndarray, n: int) -> np.ndarray:
    return ((r.astype(np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def _bern_threshold(p: float) -> int:
    return int(np.float64(np.float32(p)) * 4294967296.0)


def _counters(idx: np.ndarray, second, step: int, stream: int) -> np.ndarray:
    idx = np.asarray(idx, dtype=np.int64)
    out = np.empty(idx.shape + (4,), np.uint32)
    out[..., 0] = (idx & 0xFFFFFFFF).astype(np.uint32)
    out[..., 1] = np.asarray(second, dtype=np.int64).astype(np.uint32) if not np.isscalar(second) else np.uint32(second)
    out[..., 2] = np.uint32(step & 0xFFFFFFFF)
    out[..., 3] = np.uint32(stream & 0xFFFFFFFF)
    return out


def _element_words(e: np.ndarray, step: int, stream: int, key) -> np.ndarray:
    """word (e & 3) of counter (e >> 2 [lo], e >> 2 [hi], step, stream)"""
    q = e >> 2
    blocks = philox4x32_10(_counters(q, (q >> 32), step, stream), key)
    return np.take_along_axis(blocks, (e & 3)[..., None], axis=-1)[..., 0]


class Component:
    def __init__(self, kind: int, weight: float, p: float = 0.0, rect: Sequence[int] = (0, 0, 0, 0), size: int = 0,
                 min_prop: float = 0.0, max_prop: float = 1.0):
        self.kind, self.weight, self.p, self.rect, self.size = kind, float(weight), float(p), tuple(rect), int(size)
        self.min_prop, self.max_prop = float(min_prop), float(max_prop)


def image_mixture_components(name: str, dim: Optional[int] = None) -> List[Component]:
    """reference masking.py:235-286"""
    spec = {"MNISTMaskGenerator": (28, 0.5, 0.3, 1.0), "OmniglotMaskGenerator": (28, 0.5, 0.1, 0.6),
            "Cifar10MaskGenerator": (32, 0.3, 0.1, 0.5)}[name]
    d = dim if (dim is not None and name == "MNISTMaskGenerator") else spec[0]
    h = d // 2
    return [Component(PIXEL_BERNOULLI, 2, p=spec[1]),
            Component(FIXED_RECT, 1, rect=(0, 0, d, h)), Component(FIXED_RECT, 1, rect=(0, 0, h, d)),
            Component(FIXED_RECT, 1, rect=(0, h, d, d)), Component(FIXED_RECT, 1, rect=(h, 0, d, d)),
            Component(SQUARE, 2, size=h), Component(RECT, 2, min_prop=spec[2], max_prop=spec[3])]


def cumulative_weights(comps: Sequence[Component]) -> np.ndarray:
    w = np.array([c.weight for c in comps], np.float64)
    return np.cumsum(w / w.sum()).astype(np.float32)


def image_mask_mixture(B: int, H: int, W: int, comps: Sequence[Component], seed: int, step: int = 0,
                       stream: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """-> (mask f32 [B,H,W,1], desc int32 [B,6] = kind, y1, x1, y2, x2 (exclusive ends), component)"""
    key = _key(seed)
    cum = cumulative_weights(comps)
    d0 = philox4x32_10(_counters(np.arange(B), 0, step, stream | DESC_TAG), key)
    u = (d0[:, 0] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    mask = np.ones((B, H, W, 1), np.float32)
    desc = np.zeros((B, 6), np.int32)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        ci = len(comps) - 1
        for i in range(len(comps)):
            if u[b] < cum[i]:
                ci = i
                break
        m = comps[ci]
        y1 = x1 = y2 = x2 = 0
        if m.kind == FIXED_RECT:
            y1, x1, y2, x2 = m.rect
        elif m.kind == SQUARE:
            x1 = int(_rand_below(d0[b, 1], W - m.size))
            y1 = int(_rand_below(d0[b, 2], H - m.size))
            x2, y2 = x1 + m.size, y1 + m.size
        elif m.kind == RECT:
            area = np.float32(W * H)
            lo, hi = np.float32(m.min_prop) * area, np.float32(m.max_prop) * area
            for t in range(RECT_MAX_TRIES):
                r = philox4x32_10(_counters(np.array(b), 1 + t, step, stream | DESC_TAG), key)
                xa, xb = int(_rand_below(r[0], W)), int(_rand_below(r[1], W))
                ya, yb = int(_rand_below(r[2], H)), int(_rand_below(r[3], H))
                x1, x2 = min(xa, xb), max(xa, xb)
                y1, y2 = min(ya, yb), max(ya, yb)
                cover = np.float32((x2 - x1 + 1) * (y2 - y1 + 1))
                if lo <= cover <= hi:
                    break
            x2 += 1
            y2 += 1
        desc[b] = (m.kind, y1, x1, y2, x2, ci)
        if m.kind == PIXEL_BERNOULLI:
            e = (b * H * W + np.arange(H * W)).astype(np.int64)
            r = _element_words(e, step, stream, key)
            mask[b, :, :, 0] = (r.astype(np.uint64) < np.uint64(_bern_threshold(m.p))).reshape(H, W)
        else:
            mask[b, :, :, 0] = 1.0 - ((yy >= y1) & (yy < y2) & (xx >= x1) & (xx < x2))
    return mask, desc


def bernoulli_mask(shape: Sequence[int], p: float, seed: int, step: int = 0, stream: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    r = _element_words(np.arange(n, dtype=np.int64), step, stream, _key(seed))
    return (r.astype(np.uint64) < np.uint64(_bern_threshold(p))).astype(np.float32).reshape(shape)


def uniform_mask(B: int, D: int, lo: int, span: int, seed: int, step: int = 0, stream: int = 0) -> np.ndarray:
    key = _key(seed)
    dq = (D + 3) // 4
    i = np.arange(D, dtype=np.int64)
    out = np.zeros((B, D), np.float32)
    d0 = philox4x32_10(_counters(np.arange(B), 0, step, stream | DESC_TAG), key)
    for b in range(B):
        ctr = b * dq + (i >> 2)
        blocks = philox4x32_10(_counters(ctr, (ctr >> 32), step, stream), key)
        keys = np.take_along_axis(blocks, (i & 3)[:, None], axis=-1)[:, 0].astype(np.int64)
        nobs = min(D, lo + int(_rand_below(d0[b, 0], span)))
        order = np.lexsort((i, keys))                   # by key, ties by index
        out[b, order[:nobs]] = 1.0
    return out
