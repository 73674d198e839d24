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
  RandomPattern               masking.py:177-232 uniform noise [L, L], L = int(resolution * max_size), PIL BICUBIC resize to
                                                 [max_size, max_size], thresholded (< density) -> blobs; a mask = 1 - a random
                                                 H x W window whose blob fraction lies within density +- 0.05 (redrawn until
                                                 it does); noise redrawn after update_freq * max_size^2 pixels were handed out
  SIIDGM / GCF / CelebA       masking.py:289-325 nested mixtures, flattened here to 14 components with product weights

The random pattern is the one piece with an EXTERNAL pin: `bicubic_window` below restates Pillow's resampling (Resample.c:
precompute_coeffs with the bicubic filter a = -0.5, horizontal then vertical pass, double sums, float32 intermediate) for
a window only; tests/test_oracle_kat.py checks it bit for bit against PIL.Image.resize on the same noise field.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

PIXEL_BERNOULLI, FIXED_RECT, SQUARE, RECT, PATTERN = 0, 1, 2, 3, 4
PATTERN_TAG = 0x50415454
PATTERN_MAX_TRIES = 256
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
                 min_prop: float = 0.0, max_prop: float = 1.0, low_size: int = 0):
        """PATTERN: size = max_size, low_size = int(resolution * max_size), p = density, min_prop = density_std"""
        self.kind, self.weight, self.p, self.rect, self.size = kind, float(weight), float(p), tuple(rect), int(size)
        self.min_prop, self.max_prop, self.low_size = float(min_prop), float(max_prop), int(low_size)


GCF_RECTS = [(26, 17, 58, 36), (26, 29, 58, 48), (26, 15, 37, 50), (26, 15, 37, 34), (26, 31, 37, 50), (43, 20, 62, 44)]
SIIDGM_RECTS = [(16, 16, 48, 48), (0, 0, 64, 32), (0, 0, 32, 64), (0, 32, 64, 64), (32, 0, 64, 64)]


def celeba_components(max_size: int = 10000, resolution: float = 0.06, density: float = 0.25,
                      density_std: float = 0.05) -> List[Component]:
    """reference masking.py:289-325: CelebA = mixture[1,1,2](SIIDGM, GCF, Rectangle()); SIIDGM = mixture[2,2,2,1,1,1,1]
    (RandomPattern(max_size, resolution), ImageBernoulli(0.2), 5 fixed rectangles); GCF = 6 fixed rectangles, equal weights."""
    comps = [Component(PATTERN, 0.25 * 2 / 10, p=density, size=max_size, low_size=int(resolution * max_size),
                       min_prop=density_std),
             Component(PIXEL_BERNOULLI, 0.25 * 2 / 10, p=0.2)]
    comps += [Component(FIXED_RECT, 0.25 * (2 if i == 0 else 1) / 10, rect=r) for i, r in enumerate(SIIDGM_RECTS)]
    comps += [Component(FIXED_RECT, 0.25 / 6, rect=r) for r in GCF_RECTS]
    comps += [Component(RECT, 0.5, min_prop=0.3, max_prop=1.0)]
    return comps


def pattern_noise(low_size: int, rows: np.ndarray, cols: np.ndarray, epoch: int, stream: int, key) -> np.ndarray:
    """float32 noise cells [len(rows), len(cols)] of the low-resolution field of `epoch`: word 0 of the Philox counter
    (row * low_size + col, epoch, "PATT", stream | DESC_TAG), top 24 bits as a uniform in [0, 1)"""
    cell = (rows.astype(np.int64)[:, None] * low_size + cols.astype(np.int64)[None, :])
    ctr = np.empty(cell.shape + (4,), np.uint32)
    ctr[..., 0] = (cell & 0xFFFFFFFF).astype(np.uint32)
    ctr[..., 1] = np.uint32(epoch & 0xFFFFFFFF)
    ctr[..., 2] = np.uint32(PATTERN_TAG)
    ctr[..., 3] = np.uint32((stream | DESC_TAG) & 0xFFFFFFFF)
    w0 = philox4x32_10(ctr, key)[..., 0]
    return (w0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def _pil_bicubic(x: np.ndarray) -> np.ndarray:
    """Pillow's bicubic_filter (Resample.c), a = -0.5"""
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                    np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def bicubic_taps(out_coords: np.ndarray, in_size: int, out_size: int):
    """precompute_coeffs for an upscale (filterscale clamps to 1, support 2): -> first tap [n] and 5 normalised float64
    weights [n, 5]; taps past xmax carry weight 0"""
    scale = in_size / out_size
    centre = (out_coords.astype(np.float64) + 0.5) * scale
    lo = np.maximum((centre - 2.0 + 0.5).astype(np.int64), 0)                    # (int) truncates toward zero
    hi = np.minimum((centre + 2.0 + 0.5).astype(np.int64), in_size)
    k = np.arange(5)[None, :]
    w = np.where(k < (hi - lo)[:, None], _pil_bicubic((k + lo[:, None]) - centre[:, None] + 0.5), 0.0)
    tot = np.zeros(len(centre))
    for j in range(5):
        tot = tot + w[:, j]
    return lo, w / np.where(tot != 0.0, tot, 1.0)[:, None]


def bicubic_window(noise_fn, low_size: int, out_size: int, y0: int, x0: int, H: int, W: int) -> np.ndarray:
    """rows y0:y0+H, columns x0:x0+W of PIL.Image.fromarray(low).resize((out_size, out_size), BICUBIC), float32, where
    low[r, c] = noise_fn(rows, cols); only the cells the window touches are evaluated"""
    ylo, yw = bicubic_taps(np.arange(y0, y0 + H), low_size, out_size)
    xlo, xw = bicubic_taps(np.arange(x0, x0 + W), low_size, out_size)
    r0, r1 = int(ylo[0]), int(min(ylo[-1] + 5, low_size))
    c0, c1 = int(xlo[0]), int(min(xlo[-1] + 5, low_size))
    low = noise_fn(np.arange(r0, r1), np.arange(c0, c1)).astype(np.float64)
    hor = np.zeros((r1 - r0, W))
    for j in range(5):
        hor = hor + low[:, np.minimum(xlo + j, low_size - 1) - c0] * xw[None, :, j]
    hor = hor.astype(np.float32).astype(np.float64)
    out = np.zeros((H, W))
    for j in range(5):
        out = out + hor[np.minimum(ylo + j, low_size - 1) - r0, :] * yw[:, j, None]
    return out.astype(np.float32)


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
                       stream: int = 0, pattern_epoch: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """-> (mask f32 [B,H,W,1], desc int32 [B,6] = kind, y1, x1, y2, x2 (exclusive ends), component); for PATTERN examples
    desc carries the window origin (y0, x0) in columns 1, 2.  pattern_epoch: which noise field the PATTERN component reads
    (the device keeps it in its state tensor and bumps it once update_freq * max_size^2 pixels were handed out)."""
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
        if m.kind == PATTERN:
            hw = H * W
            lo_cnt = (float(np.float32(m.p)) - float(np.float32(m.min_prop))) * hw
            hi_cnt = (float(np.float32(m.p)) + float(np.float32(m.min_prop))) * hw
            noise = lambda rr, cc: pattern_noise(m.low_size, rr, cc, pattern_epoch, stream, key)   # noqa: E731
            for t in range(PATTERN_MAX_TRIES):
                r = philox4x32_10(_counters(np.array(b), 1 + t, step, stream | DESC_TAG), key)
                x1 = int(_rand_below(r[0], m.size - W + 1))
                y1 = int(_rand_below(r[1], m.size - H + 1))
                blob = bicubic_window(noise, m.low_size, m.size, y1, x1, H, W) < np.float32(m.p)
                cnt = float(blob.sum())
                if lo_cnt < cnt < hi_cnt:
                    break
            mask[b, :, :, 0] = 1.0 - blob
            desc[b] = (m.kind, y1, x1, 0, 0, ci)
            continue
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


def random_indices(B: int, N: int, seed: int, step: int = 0, stream: int = 0) -> np.ndarray:
    """pm_random_indices (the shuffle of reference utils.py:43 as sampling with replacement): index i = word i % 4 of the
    Philox counter (i // 4, 0, step, stream), mapped to [0, N) by multiply-shift"""
    i = np.arange(B, dtype=np.int64)
    blocks = philox4x32_10(_counters(i >> 2, 0, step, stream), _key(seed))
    words = np.take_along_axis(blocks, (i & 3)[:, None], axis=-1)[:, 0]
    return _rand_below(words, N).astype(np.int32)
