"""Naive-loop numpy statements of the XLA / TFP primitives.  TEST INFRASTRUCTURE ONLY.

Deliberately written index-by-index from the mathematical definitions (SURVEY.md Appendix A1, A2,
A6) so that they share no code path with ``pm_vae_oracle.py`` (which leans on torch's conv2d).
``tests/test_oracle_kat.py`` checks the two against each other on small shapes.  PARITY UNPINNED
(see pm_vae_oracle.py header): there is no reference-produced vector to compare with.
"""
from __future__ import annotations

import math

import numpy as np


def conv2d_nhwc(x, w, stride, padding):
    """lax.conv_general_dilated NHWC/HWIO; SAME = (total//2, total - total//2) (Appendix A1)."""
    n, ih, iw, _ = x.shape
    kh, kw, cin, cout = w.shape
    if padding == "SAME":
        oh, ow = -(-ih // stride), -(-iw // stride)
        pt = max((oh - 1) * stride + kh - ih, 0) // 2
        pl = max((ow - 1) * stride + kw - iw, 0) // 2
    else:
        oh, ow = (ih - kh) // stride + 1, (iw - kw) // stride + 1
        pt = pl = 0
    y = np.zeros((n, oh, ow, cout))
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(kh):
                for kx in range(kw):
                    sy, sx = oy * stride + ky - pt, ox * stride + kx - pl
                    if 0 <= sy < ih and 0 <= sx < iw:
                        y[:, oy, ox, :] += x[:, sy, sx, :] @ w[ky, kx]
    return y


def conv2d_transpose_nhwc(x, w, stride, padding):
    """lax.conv_transpose, weight [kh,kw,Cout,Cin], kernel NOT flipped (Appendix A2):
    out[oy] = sum_ky xdil[oy + ky - pad_a] w[ky], xdil[j] = x[j/stride] when stride | j."""
    n, ih, iw, cin = x.shape
    kh, kw, cout, _ = w.shape

    def pads(k):
        if padding == "SAME":
            pad_len = k + stride - 2
            pad_a = k - 1 if stride > k - 1 else int(math.ceil(pad_len / 2))
        else:
            pad_len = k + stride - 2 + max(k - stride, 0)
            pad_a = k - 1
        return pad_a, pad_len - pad_a

    (pa_h, pb_h), (pa_w, pb_w) = pads(kh), pads(kw)
    oh = (ih - 1) * stride + 1 + pa_h + pb_h - kh + 1
    ow = (iw - 1) * stride + 1 + pa_w + pb_w - kw + 1
    y = np.zeros((n, oh, ow, cout))
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(kh):
                for kx in range(kw):
                    jy, jx = oy + ky - pa_h, ox + kx - pa_w
                    if jy % stride or jx % stride:
                        continue
                    sy, sx = jy // stride, jx // stride
                    if 0 <= sy < ih and 0 <= sx < iw:
                        y[:, oy, ox, :] += x[:, sy, sx, :] @ w[ky, kx].T
    return y


def fill_triangular(v):
    """TFP fill_triangular lower (Appendix A6): element (r, c<=r) = cat[r*n + c],
    cat = concat(v[n:], reverse(v))."""
    m = len(v)
    n = int((math.isqrt(8 * m + 1) - 1) // 2)
    out = np.zeros((n, n))
    for r in range(n):
        for c in range(r + 1):
            t = r * n + c
            out[r, c] = v[n + t] if t < m - n else v[2 * m - n - 1 - t]
    return out


def gmm_log_pdf(logits, means, scales, z):
    """log sum_c softmax(logits)_c N(z; mean_c, scale_c), plain formula."""
    w = np.exp(logits - logits.max())
    w = w / w.sum()
    pdf = np.exp(-0.5 * ((z - means) / scales) ** 2) / (scales * math.sqrt(2 * math.pi))
    return math.log(float((w * pdf).sum()))
