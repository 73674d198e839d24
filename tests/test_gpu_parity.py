"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

float32 kernels vs a float64 oracle: tolerances are stated per test and sit well inside the
1e-3 relative bar of BASELINE.json's north_star.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import pm_vae_oracle as O
from tests.ref_configs import pm_vae_gas, pm_vae_mnist

pytestmark = pytest.mark.gpu

F64 = torch.float64


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    from tests import conftest

    conftest.confirm_compared()          # the kernels launched so far in this test have a compared result
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _compared():
    """a scalar of the step (loss / metrics) is about to be compared with the oracle's: the step's kernels count as compared"""
    from tests import conftest

    conftest.confirm_compared()


def g32(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, dtype=F64) * scale)


# ----------------------------------------------------------------------------------------------
# gather-GEMM engine vs oracle convs, forward and both gradients
# ----------------------------------------------------------------------------------------------
CONV_CASES = [
    # kind, B, H, Cin, Cout, k, s, padding
    ("conv", 3, 28, 1, 32, 5, 1, "SAME"), ("conv", 3, 28, 2, 32, 5, 1, "SAME"),
    ("conv", 2, 28, 32, 32, 5, 2, "SAME"), ("conv", 2, 14, 32, 64, 5, 1, "SAME"),
    ("conv", 2, 14, 64, 64, 5, 2, "SAME"), ("conv", 5, 7, 64, 128, 7, 1, "VALID"),
    ("conv", 2, 9, 4, 8, 4, 2, "SAME"), ("conv", 1, 11, 3, 5, 3, 1, "SAME"),
    ("convT", 5, 1, 32, 64, 7, 1, "VALID"), ("convT", 2, 7, 64, 64, 5, 2, "SAME"),
    ("convT", 2, 14, 64, 32, 5, 1, "SAME"), ("convT", 2, 14, 32, 32, 5, 2, "SAME"),
    ("convT", 2, 28, 32, 1, 5, 1, "SAME"), ("convT", 2, 6, 8, 4, 4, 2, "SAME"),
    ("dense", 37, 1, 192, 256, 1, 1, "VALID"), ("dense", 300, 1, 16, 8, 1, 1, "VALID"),
    # VQ-VAE layers (reference vqvae.py:151-249 at configs/vqvae_mnist.py sizes)
    ("conv", 8, 28, 1, 16, 4, 2, "SAME"), ("conv", 8, 14, 16, 32, 4, 2, "SAME"), ("conv", 32, 7, 32, 32, 3, 1, "SAME"),
    ("conv", 32, 7, 32, 32, 1, 1, "SAME"), ("conv", 32, 7, 32, 64, 1, 1, "SAME"), ("conv", 32, 7, 64, 32, 3, 1, "SAME"),
    ("convT", 8, 7, 32, 16, 4, 2, "SAME"), ("convT", 8, 14, 16, 1, 4, 2, "SAME"),
    # VDVAE bottleneck layers (reference vdvae.py:282-292 at width 192, bottleneck 48): C % 32 != 0 on the bf16 path
    ("conv", 4, 14, 48, 48, 3, 1, "SAME"), ("conv", 4, 7, 48, 192, 1, 1, "SAME"), ("conv", 4, 7, 192, 48, 1, 1, "SAME"),
    ("conv", 3, 3, 48, 152, 1, 1, "SAME"), ("conv", 2, 28, 48, 224, 1, 1, "SAME"), ("dense", 70, 1, 24, 40, 1, 1, "VALID"),
    # stride-1 layers on grids >= 12 wide: the patch-staged bf16x3 form (tiles 4x32 / 8x16, ragged edges, 3x3 / 5x5 / 7x7)
    ("convT", 2, 28, 32, 32, 5, 1, "SAME"), ("conv", 3, 28, 32, 32, 3, 1, "SAME"), ("conv", 2, 13, 32, 32, 5, 1, "SAME"),
    ("conv", 2, 30, 64, 64, 3, 1, "SAME"), ("conv", 1, 17, 96, 40, 7, 1, "SAME"), ("convT", 3, 12, 64, 96, 3, 1, "SAME"),
    # thin layers, lane = channel / wide -> 1 forms (csrc/pm_thin.hip): 3x3 and 5x5, ragged widths, N < 32, more
    # images than workgroups, both tap directions
    ("conv", 3, 11, 2, 16, 3, 1, "SAME"), ("conv", 300, 14, 1, 24, 5, 1, "SAME"), ("convT", 2, 10, 8, 1, 3, 1, "SAME"),
    ("conv", 2, 12, 16, 1, 5, 1, "SAME"), ("convT", 3, 9, 2, 20, 5, 1, "SAME"), ("conv", 2, 40, 1, 32, 3, 1, "SAME"),
    ("convT", 2, 9, 64, 1, 3, 1, "SAME"), ("conv", 2, 12, 32, 1, 5, 1, "SAME"),   # wide -> 1 on the matrix cores
    # batches >= 128: the image-resident forms (image_conv_bf16 / image_d2_bf16): ragged position counts, 1 / 2 / 4 row tiles
    # per wave, 1 / 2 / 4 column tiles, stride 2, flipped taps, and the class-major walk of the zero-dilated problems
    # (stride-2 transposed convolutions forward, stride-2 convolutions' data gradients; 4x4 and 5x5 kernels)
    ("conv", 128, 13, 32, 32, 5, 1, "SAME"), ("conv", 128, 14, 32, 64, 5, 2, "SAME"), ("conv", 128, 12, 64, 128, 3, 1, "SAME"),
    ("convT", 128, 9, 32, 32, 5, 1, "SAME"), ("convT", 128, 7, 64, 64, 5, 2, "SAME"), ("convT", 128, 6, 32, 32, 4, 2, "SAME"),
    ("conv", 128, 12, 32, 32, 5, 2, "SAME"), ("convT", 128, 14, 32, 32, 5, 2, "SAME"),
    # the largest LDS-resident images (25 - 28 rows: 100 - 125 KB, one workgroup per CU), stride 1 and 2, forward and as data
    # gradients (flipped taps), odd heights, 3x3 and 5x5
    ("conv", 128, 28, 32, 32, 5, 1, "SAME"), ("conv", 128, 28, 32, 32, 5, 2, "SAME"), ("convT", 128, 28, 32, 32, 5, 1, "SAME"),
    ("conv", 128, 27, 32, 64, 3, 1, "SAME"), ("conv", 128, 25, 32, 32, 5, 2, "SAME"), ("convT", 130, 26, 32, 32, 3, 1, "SAME"),
    # whole-image layers with one output position (skinny_gemm_bf16: K = 3136 over the waves of a workgroup), forward and as a
    # data gradient with flipped taps, ragged row / column tiles
    ("conv", 256, 7, 64, 128, 7, 1, "VALID"), ("convT", 130, 1, 32, 64, 7, 1, "VALID"), ("conv", 21, 6, 32, 40, 6, 1, "VALID"),
    ("dense", 256, 1, 560, 128, 1, 1, "VALID"), ("dense", 100, 1, 128, 560, 1, 1, "VALID"),   # C % 32 != 0: part-padded last chunk
    # small images under long kernels (the PixelCNN's 7 x 7 x 256 grids): two images per workgroup share one weight pass
    ("conv", 256, 7, 256, 256, 3, 1, "SAME"), ("convT", 256, 7, 256, 128, 3, 1, "SAME"),
    # plain-rows weight gradients with >= 1024 rows (the ResidualMLP hidden layers of a B = 256 step), ragged rows / columns
    ("dense", 8192, 1, 256, 256, 1, 1, "VALID"), ("dense", 2100, 1, 192, 256, 1, 1, "VALID"),
    ("dense", 1030, 1, 128, 132, 1, 1, "VALID"), ("dense", 1500, 1, 320, 200, 1, 1, "VALID"),
    # 64-channel layers on grids <= 8 wide at batches >= 32: weight gradients with one tap ROW per workgroup class
    # (rowtap_wgrad_bf16): stride 2 and 1, odd input sizes (7 / 8 output rows: a half-empty last k16 step or none), the
    # role-swapped transposed form (flipped taps, gathered = dy), the largest sizes it takes (16 x 16 gathered, 8 x 8 dense)
    ("conv", 64, 14, 64, 64, 5, 2, "SAME"), ("conv", 33, 13, 64, 64, 5, 2, "SAME"), ("conv", 48, 8, 64, 64, 5, 1, "SAME"),
    ("convT", 40, 8, 64, 64, 5, 2, "SAME"), ("conv", 32, 16, 64, 64, 5, 2, "SAME"),
    # short tile grids under a long K: the split-K form, K slices in slabs added in slice order (8 - 24 slices; the CelebA
    # PixelCNN's 8 x 8 x 256 layers at per-GPU batch 16, MLP layers at small batch, a ragged one)
    ("conv", 16, 8, 256, 256, 3, 1, "SAME"), ("dense", 64, 1, 2048, 512, 1, 1, "VALID"), ("dense", 100, 1, 1100, 200, 1, 1, "VALID"),
    # deep stride-1 layers on grids >= 12 wide whose patch only fits LDS a channel chunk at a time (patch_conv_cp_bf16): full 3 x 3
    # kernels (128-channel passes, 24 staged pieces per thread), forward and - transposed - with flipped taps, 32-wide tiles
    ("conv", 4, 16, 256, 128, 3, 1, "SAME"), ("convT", 3, 16, 256, 64, 3, 1, "SAME"), ("conv", 2, 24, 256, 40, 3, 1, "SAME"),
]


@pytest.mark.parametrize("kind,B,H,ci,co,k,s,padding", CONV_CASES)
def test_layer_fwd_dgrad_wgrad(kind, B, H, ci, co, k, s, padding):
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import ACT_LEAKY, LayerGeom

    import zlib

    # a stable per-case seed: hash() of a tuple holding a str changes with PYTHONHASHSEED from run to run
    seed = zlib.crc32(repr((kind, B, H, ci, co, k, s, padding)).encode()) + int(os.environ.get("PM_TEST_SEED_OFFSET", "0"))
    gen = torch.Generator().manual_seed(seed % 2 ** 31)
    if kind == "conv":
        geom = LayerGeom.conv(H, H, ci, co, k, s, padding)
    elif kind == "convT":
        geom = LayerGeom.conv_t(H, H, ci, co, k, s, padding)
    else:
        geom = LayerGeom.dense(ci, co)
    x = g32((B, geom.IH, geom.IW, ci), gen)
    w = g32(geom.weight_shape, gen, 1.0 / math.sqrt(k * k * ci))
    bias = g32((co,), gen, 0.1)
    dy = g32((B, geom.OH, geom.OW, co), gen)

    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    if kind == "conv":
        pre = O.conv2d(xr, wr, br, s, padding)
    elif kind == "convT":
        pre = O.conv2d_transpose(xr, wr, br, s, padding)
    else:
        pre = (xr.reshape(B, ci) @ wr + br).reshape(B, 1, 1, co)
    assert tuple(pre.shape) == (B, geom.OH, geom.OW, co)
    y = O.leaky_relu(pre)
    y.backward(dy)

    d = dev()
    xd, wd, bd = x.float().to(d), w.float().to(d), bias.float().to(d)
    yd = torch.empty((B, geom.OH, geom.OW, co), device=d)
    ops.layer_forward(geom, xd, wd, bd, yd, out_act=ACT_LEAKY)
    assert rel_err(yd, y) < 2e-6
    # the same layer on the bf16 matrix cores (bf16x3 split, f32-grade): ~1e-5 relative
    from posterior_matching_amd.models.core import ParamStore
    st = ParamStore()
    st.add("w", geom.weight_shape, fan_in=1)
    st.add("b", (co,))
    hf, hd = st.request_split("w", geom, "fwd"), st.request_split("w", geom, "dgrad")
    st.allocate(d)
    st.load_dict({"w": w})
    if st.split_view(hf) is not None:
        y2 = torch.empty_like(yd)
        ops.layer_forward(geom, xd, wd, bd, y2, out_act=ACT_LEAKY, wsplit=st.split_view(hf))
        assert rel_err(y2, y) < 3e-5

    # gradient w.r.t. the pre-activation, then dgrad / wgrad
    dpre = (dy * torch.where(pre >= 0, 1.0, 0.01)).float().to(d).contiguous()
    dxd = torch.empty_like(xd)
    ops.layer_dgrad(geom, dpre, wd, dxd)
    assert rel_err(dxd, xr.grad) < 2e-6
    if st.split_view(hd) is not None:
        dx2 = torch.empty_like(dxd)
        ops.layer_dgrad(geom, dpre, wd, dx2, wsplit=st.split_view(hd))
        assert rel_err(dx2, xr.grad) < 3e-5
        if ops.dgrad_insum_ok(geom, B, st.split_view(hd), force=True):
            # the data-gradient launch that also leaves the bias gradient (column sums of dpre; accumulates): same dx, bit for bit
            dx3, db3 = torch.empty_like(dxd), torch.full_like(bd, 0.5)
            ops.layer_dgrad(geom, dpre, wd, dx3, wsplit=st.split_view(hd), in_colsum=db3)
            assert torch.equal(dx3, dx2)
            sc = (dy * torch.where(pre >= 0, 1.0, 0.01)).abs().sum((0, 1, 2)).max().item()
            assert (db3.cpu().double() - 0.5 - br.grad).abs().max().item() < 2e-6 * sc
        # (compiled out of the default build - ops.dgrad_insum_ok is False then; -DPM_IMAGE_INSUM builds run this branch)
    dwd, dbd = torch.zeros_like(wd), torch.zeros_like(bd)
    ops.layer_wgrad(geom, xd, dpre, dwd, dbd, bf16=False)          # f32 MFMA
    assert rel_err(dwd, wr.grad) < 2e-6
    # a bias gradient is a sum over every position: when the terms cancel (single-channel outputs) the f32 rounding is
    # small against the terms, not against the result -> the yardstick is sum |dpre| per channel
    db_scale = (dy * torch.where(pre >= 0, 1.0, 0.01)).abs().sum((0, 1, 2)).max().item()
    assert (dbd.cpu().double() - br.grad).abs().max().item() < 2e-6 * db_scale
    dwd.zero_(); dbd.zero_()
    ops.layer_wgrad(geom, xd, dpre, dwd, dbd, bf16=True)           # bf16x3 where the shape qualifies
    assert rel_err(dwd, wr.grad) < 3e-5
    assert (dbd.cpu().double() - br.grad).abs().max().item() < 3e-5 * db_scale
    # the same launches with gradients that live in a ParamStore: every kernel form leaves per-split PARTIAL SUMS (plain
    # stores into the store's arenas), added in a fixed order when the gradients are read - twice the same bits, both modes
    for use_bf16 in (False, True):
        runs = []
        for _ in range(2):
            st.zero_grad()
            ops.layer_wgrad(geom, xd, dpre, st.g["w"], st.g["b"], bf16=use_bf16)
            gd = st.to_dict("g")
            runs.append((gd["w"].clone(), gd["b"].clone()))
        assert rel_err(runs[0][0], wr.grad) < (3e-5 if use_bf16 else 2e-6)
        assert (runs[0][1].cpu().double() - br.grad).abs().max().item() < (3e-5 if use_bf16 else 2e-6) * db_scale
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_epilogue_aux_res_inact():
    """in_act on load, act'(aux) and residual in the epilogue (ResidualMLP backward chain)."""
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import ACT_RELU, LayerGeom

    gen = torch.Generator().manual_seed(3)
    B, ci, co = 70, 48, 40
    x, w, bias = g32((B, ci), gen), g32((ci, co), gen, 0.2), g32((co,), gen)
    aux, res = g32((B, co), gen), g32((B, co), gen)
    want = (O.relu(x) @ w + bias) * (aux > 0).double() + res
    d = dev()
    geom = LayerGeom.dense(ci, co)
    desc = geom._desc(B, "fwd")
    desc.in_act, desc.aux_act = ACT_RELU, ACT_RELU
    out = torch.empty((B, co), device=d)
    ops.gather_gemm(desc, x.float().to(d), w.float().to(d), bias.float().to(d), aux.float().to(d), res.float().to(d), out)
    assert rel_err(out, want) < 2e-6


# ----------------------------------------------------------------------------------------------
# heads
# ----------------------------------------------------------------------------------------------
def test_thin_forms_inact_aux_res():
    """pending input activation, act'(aux) and residual on the thin kernels' lane / wide->1 forms, and the
    input activation of the thin weight gradient."""
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import ACT_RELU, LayerGeom

    gen = torch.Generator().manual_seed(5)
    d = dev()
    for ci, co, H in ((2, 32, 13), (32, 1, 13)):
        B = 3
        geom = LayerGeom.conv(H, H, ci, co, 5, 1, "SAME")
        x, w, bias = g32((B, H, H, ci), gen), g32(geom.weight_shape, gen, 0.2), g32((co,), gen)
        aux, res = g32((B, H, H, co), gen), g32((B, H, H, co), gen)
        want = O.conv2d(O.relu(x), w, bias, 1, "SAME") * (aux > 0).double() + res
        desc = geom._desc(B, "fwd")
        desc.in_act, desc.aux_act = ACT_RELU, ACT_RELU
        assert ops._thin_ok(desc)
        out = torch.empty((B, H, H, co), device=d)
        ops.thin_conv(desc, x.float().to(d), w.float().to(d), bias.float().to(d), aux.float().to(d),
                      res.float().to(d), out)
        assert rel_err(out, want) < 2e-6
    # wgrad with relu pending on the gathered operand
    B, H, ci, co = 4, 13, 2, 32
    geom = LayerGeom.conv(H, H, ci, co, 5, 1, "SAME")
    x, dy = g32((B, H, H, ci), gen), g32((B, H, H, co), gen)
    wr = g32(geom.weight_shape, gen).requires_grad_(True)
    O.conv2d(O.relu(x), wr, None, 1, "SAME").backward(dy)
    dw, db = torch.zeros(geom.weight_shape, device=d), torch.zeros(co, device=d)
    ops.layer_wgrad(geom, x.float().to(d), dy.float().to(d), dw, db, in_act=ACT_RELU)
    assert rel_err(dw, wr.grad) < 2e-6
    assert rel_err(db, dy.sum((0, 1, 2))) < 2e-6


@pytest.mark.parametrize("B,k", [(5, 32), (7, 16), (256, 32)])
def test_tril_sample_kl(B, k):
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(10 + B)
    P = k + k * (k + 1) // 2
    prm, eps = g32((B, P), gen), g32((B, k), gen)
    dz, gk = g32((B, k), gen), g32((B,), gen)
    pr = prm.clone().requires_grad_(True)
    loc, tril = pr[:, :k], O.fill_scale_tril(pr[:, k:])
    z = loc + torch.einsum("bij,bj->bi", tril, eps)
    kl = O.mvn_tril_kl_to_std_normal(loc, tril)
    ((z * dz).sum() + (kl * gk).sum()).backward()
    d = dev()
    prd, epd = prm.float().to(d), eps.float().to(d)
    zd, kld = torch.empty((B, k), device=d), torch.empty(B, device=d)
    ops.tril_sample_kl_fwd(prd, epd, zd, kld)
    assert rel_err(zd, z) < 2e-6 and rel_err(kld, kl) < 2e-6
    dprm = torch.full((B, P), float("nan"), device=d)
    ops.tril_sample_kl_bwd(prd, epd, dz.float().to(d), gk.float().to(d), dprm)
    assert rel_err(dprm, pr.grad) < 2e-6
    # two gradients w.r.t. z summed on load (the decoder's and the posterior-matching branch's): dz = a + b
    a = g32((B, k), gen)
    dprm2 = torch.full((B, P), float("nan"), device=d)
    ops.tril_sample_kl_bwd(prd, epd, a.float().to(d), gk.float().to(d), dprm2, dz2=(dz.float() - a.float()).to(d))
    assert rel_err(dprm2, pr.grad) < 2e-6


@pytest.mark.parametrize("B,k", [(5, 16), (130, 32)])
def test_tril_logprob(B, k):
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(20 + B)
    P = k + k * (k + 1) // 2
    prm, z, g = g32((B, P), gen), g32((B, k), gen), g32((B,), gen)
    pr, zr = prm.clone().requires_grad_(True), z.clone().requires_grad_(True)
    lp = O.mvn_tril_log_prob(zr, pr[:, :k], O.fill_scale_tril(pr[:, k:]))
    (lp * g).sum().backward()
    d = dev()
    lpd = torch.empty(B, device=d)
    ops.tril_logprob_fwd(prm.float().to(d), z.float().to(d), lpd)
    assert rel_err(lpd, lp) < 5e-6
    dprm, dzd = torch.full((B, P), float("nan"), device=d), torch.empty((B, k), device=d)
    ops.tril_logprob_bwd(prm.float().to(d), z.float().to(d), g.float().to(d), dprm, dzd)
    assert rel_err(dprm, pr.grad) < 2e-5 and rel_err(dzd, zr.grad) < 2e-5


def test_bernoulli_and_normal_ll():
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import ACT_LEAKY

    gen = torch.Generator().manual_seed(30)
    B, D = 9, 784
    pre, x, g = g32((B, D), gen, 3.0), torch.rand((B, D), generator=gen, dtype=F64), g32((B,), gen)
    pr = pre.clone().requires_grad_(True)
    logits = O.leaky_relu(pr)
    ll = O.bernoulli_log_prob(logits, x).sum(-1)
    (ll * g).sum().backward()
    d = dev()
    lg = logits.detach().float().to(d)
    lld = torch.empty(B, device=d)
    ops.bernoulli_ll_fwd(lg, x.float().to(d), lld)
    assert rel_err(lld, ll) < 2e-6
    dpre = torch.empty((B, D), device=d)
    ops.bernoulli_ll_bwd(lg, x.float().to(d), g.float().to(d), dpre, ACT_LEAKY)
    assert rel_err(dpre, pr.grad) < 2e-6
    # the one-pass form the train step uses (d loss / d ll on the device before the forward pass): same numbers, bit for bit
    ll2, dpre2 = torch.empty(B, device=d), torch.empty((B, D), device=d)
    ops.bernoulli_ll_fwd_bwd(lg, x.float().to(d), g.float().to(d), ll2, dpre2, ACT_LEAKY)
    assert rel_err(ll2, ll) < 2e-6 and rel_err(dpre2, pr.grad) < 2e-6
    assert torch.equal(ll2, lld) and torch.equal(dpre2, dpre)

    loc, ls = g32((B, 8), gen), torch.tensor(0.3, dtype=F64)
    xs = g32((B, 8), gen)
    lr_, lsr = loc.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    nl = O.normal_log_prob(xs, lr_, torch.exp(lsr)).sum(-1)
    (nl * g).sum().backward()
    nld = torch.empty(B, device=d)
    lsd = ls.float().reshape(()).to(d)
    ops.normal_ll_fwd(loc.float().to(d), xs.float().to(d), lsd, nld)
    assert rel_err(nld, nl) < 2e-6
    dloc, dls = torch.empty((B, 8), device=d), torch.zeros((), device=d)
    ops.normal_ll_bwd(loc.float().to(d), xs.float().to(d), lsd, g.float().to(d), dloc, dls)
    assert rel_err(dloc, lr_.grad) < 2e-6 and rel_err(dls, lsr.grad) < 2e-6


@pytest.mark.parametrize("B,k,nc", [(6, 32, 10), (37, 10, 10), (5, 3, 4), (130, 50, 16)])   # k = 10: configs/pm_vade_mnist.py
def test_gmm_logprob_and_argmm_input(B, k, nc):
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(40 + k)
    cd = 128
    head, z, g = g32((k * B, 3 * nc), gen), g32((B, k), gen), g32((B,), gen)
    hr, zr = head.clone().requires_grad_(True), z.clone().requires_grad_(True)
    # row i*B+b holds the 30 parameters of latent dim i at step i
    h3 = hr.reshape(k, B, 3 * nc)
    cols = torch.stack([O.gmm_log_prob_columns(h3[i], zr[:, i:i + 1], 1, nc)[:, 0] for i in range(k)], 0)
    mll = cols.sum(0)
    (mll * g).sum().backward()
    d = dev()
    md = torch.empty(B, device=d)
    ops.gmm_logprob_fwd(head.float().to(d), z.float().to(d), md, nc)
    assert rel_err(md, mll) < 2e-6
    dh, dz = torch.empty((k * B, 3 * nc), device=d), torch.zeros((B, k), device=d)
    ops.gmm_logprob_bwd(head.float().to(d), z.float().to(d), g.float().to(d), dh, dz, nc, accumulate_dz=True)
    assert rel_err(dh, hr.grad) < 5e-6 and rel_err(dz, zr.grad) < 5e-6

    ctx = g32((B, cd), gen)
    inp = torch.empty((k * B, 2 * k + cd), device=d)
    ops.argmm_build_input(z.float().to(d), ctx.float().to(d), inp)
    ar = torch.arange(k, dtype=F64)
    mask = (ar[None, :] < ar[:, None]).double()[:, None, :].expand(k, B, k)
    want = torch.cat([z[None] * mask, mask, ctx[None].expand(k, B, cd)], -1).reshape(k * B, -1)
    assert torch.equal(inp.cpu().double(), want.float().double())
    dinp = g32((k * B, 2 * k + cd), gen)
    dzd, dctx = torch.zeros((B, k), device=d), torch.empty((B, cd), device=d)
    ops.argmm_input_bwd(dinp.float().to(d), dzd, dctx, B, k, cd, accumulate_dz=True)
    d3 = dinp.reshape(k, B, -1)
    assert rel_err(dzd, (d3[:, :, :k] * mask).sum(0)) < 2e-6
    assert rel_err(dctx, d3[:, :, 2 * k:].sum(0)) < 2e-6


# ----------------------------------------------------------------------------------------------
# whole model: forward outputs, every gradient, and a few optimizer steps
# ----------------------------------------------------------------------------------------------
def _inputs(cfg_name, B, seed):
    rng = np.random.default_rng(seed)
    if cfg_name == "mnist":
        cfg, xs = pm_vae_mnist(), (28, 28, 1)
        x = rng.uniform(size=(B,) + xs) * (rng.uniform(size=(B,) + xs) < 0.19)
        b = (rng.uniform(size=(B,) + xs) < 0.5).astype(np.float64)
        b[: B // 2, :, 14:, :] = 0.0
    else:
        cfg, xs = pm_vae_gas(), (8,)
        x = rng.normal(size=(B,) + xs)
        b = (rng.uniform(size=(B,) + xs) < 0.5).astype(np.float64)
    eps = rng.normal(size=(B, cfg["model"]["latent_dim"]))
    return cfg, xs, torch.tensor(x), torch.tensor(b), torch.tensor(eps)


def _product_model(cfg, xs, seed=11, perturb=True, bf16x3=True):
    from posterior_matching_amd.models import PosteriorMatchingVAE

    m = PosteriorMatchingVAE.from_config(cfg["model"], device="cuda:0", seed=seed)
    m.init(xs)
    m.store.use_bf16 = bf16x3
    if perturb:  # haiku init has zero biases / log_scale: move them so that their paths are exercised
        gen = torch.Generator().manual_seed(seed)
        vals = {n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()}
        m.load_params(vals)
    return m


@pytest.mark.parametrize("name,B,bf16x3", [("mnist", 6, False), ("gas", 37, False), ("mnist", 6, True),
                                           ("gas", 37, True)])
def test_model_forward_and_grads(name, B, bf16x3):
    """bf16x3=False: every GEMM on the f32 MFMA (strict yardstick).  bf16x3=True (the default fast
    path): forward / data-gradient GEMMs on the bf16 matrix cores with hi+lo operand splitting."""
    from posterior_matching_amd import ops
    from posterior_matching_amd.engine import loss_cfg_from_config

    cfg, xs, x, b, eps = _inputs(name, B, 5)
    m = _product_model(cfg, xs, bf16x3=bf16x3)
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    assert list(p64) == list(O.param_shapes(cfg["model"], xs))
    step = 13500                                                     # gas: beta = 0.5 there
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = O.pm_vae_loss(leaves, cfg, x, b, eps, step)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))

    d = dev()
    xd, bd, ed = x.float().to(d), b.float().to(d), eps.float().to(d)
    got = m(xd, bd, is_training=True, eps=ed)
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(got[key], out[key]) < (1e-4 if bf16x3 else 1e-5), key
    step_dev = torch.tensor([step], dtype=torch.int32, device=d)
    metrics = torch.zeros(8, device=d)
    g = [torch.empty(B, device=d) for _ in range(3)]
    ops.pmvae_loss(got["reconstruction_ll"], got["kl"], got["matching_ll"], loss_cfg_from_config(cfg, B), step_dev,
                   metrics, *g)
    mv = metrics.cpu().double()
    assert abs(mv[0].item() - loss.item()) < (1e-4 if bf16x3 else 1e-5) * abs(loss.item())
    assert mv[4].item() == pytest.approx(aux["beta"])
    m.zero_grad()
    m.backward(*g)
    torch.cuda.synchronize()
    gd = m.grads_dict()
    # yardstick: the same restatement evaluated in float32 on the CPU.  The HIP path (f32 MFMA,
    # k-ordered fma chains) may not be worse than 3x what plain f32 arithmetic gives, and never
    # worse than 2e-4 relative per tensor (the north-star bar is 1e-3).
    l32 = {n: t.float().clone().requires_grad_(True) for n, t in p64.items()}
    loss32, _, _ = O.pm_vae_loss(l32, cfg, x.float(), b.float(), eps.float(), step)
    g32_ = dict(zip(l32, torch.autograd.grad(loss32, list(l32.values()))))
    table = [(rel_err(gd[n], grads[n]), rel_err(g32_[n], grads[n]), n) for n in grads]
    for e, e32, n in table:
        if not bf16x3:
            assert e < max(5e-5, 10 * e32) and e < 2e-4, (n, e, e32)
        else:
            # Forward values carry ~5e-6 (16 kept mantissa bits per operand); the loss gradient
            # w.r.t. them is ill-conditioned (softmax / sigmoid differences), which amplifies that
            # ~100x exactly as it amplifies the 1e-7 of the f32 path to 1e-5.  Small batches are the
            # worst case.  The bar for this mode: 1e-2 per tensor, outputs (above) at 1e-5.
            assert e < 1e-2, (n, e, e32)


@pytest.mark.parametrize("name,D,step", [("power", 6, 13500), ("hepmass", 21, 700), ("bsds", 63, 130000)])
def test_remaining_uci_configs_forward_loss_and_gradients(name, D, step):
    """configs/pm_vae_{power,hepmass,bsds}.py (the gas family; bsds: latent 64, five LayerNorm blocks, the monotonic beta
    schedule - beta = 0.5 at step 130 000): outputs 1e-4, loss 1e-4, every gradient tensor 1e-2 vs the oracle (default arithmetic)"""
    import tests.ref_configs as RC
    from posterior_matching_amd import ops
    from posterior_matching_amd.engine import loss_cfg_from_config

    cfg, xs, B = getattr(RC, f"pm_vae_{name}")(), (D,), 33
    rng = np.random.default_rng(D)
    x, b = torch.tensor(rng.normal(size=(B, D))), torch.tensor((rng.uniform(size=(B, D)) < 0.5).astype(np.float64))
    eps = torch.tensor(rng.normal(size=(B, cfg["model"]["latent_dim"])))
    m = _product_model(cfg, xs)
    # a freshly initialised 64 x 64 scale_tril behind LayerNorm features has O(1) off-diagonals against a 0.69 diagonal: its
    # triangular solve grows like 1.5^64 and log q(z | x_o) is -1e70 in float64 (-inf in float32, on either side).  A trained
    # head is well conditioned; the test scales the two TriL projections down instead of training one.
    vals = {n: (0.05 * t if n.endswith("posterior_dist/linear/w") else t) for n, t in m.params_dict().items()}
    m.load_params(vals)
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = O.pm_vae_loss(leaves, cfg, x, b, eps, step)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    d = dev()
    got = m(x.float().to(d), b.float().to(d), is_training=True, eps=eps.float().to(d))
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(got[key], out[key]) < 1e-4, key
    metrics, g = torch.zeros(8, device=d), [torch.empty(B, device=d) for _ in range(3)]
    ops.pmvae_loss(got["reconstruction_ll"], got["kl"], got["matching_ll"], loss_cfg_from_config(cfg, B),
                   torch.tensor([step], dtype=torch.int32, device=d), metrics, *g)
    assert abs(metrics[0].item() - loss.item()) < 1e-4 * abs(loss.item())
    assert metrics[4].item() == pytest.approx(aux["beta"])
    if name == "bsds":
        assert aux["beta"] == pytest.approx(0.5)
    m.zero_grad()
    m.backward(*g)
    torch.cuda.synchronize()
    gd = m.grads_dict()
    worst = max((rel_err(gd[n], grads[n]), n) for n in grads if grads[n].norm() > 0)
    assert worst[0] < 1e-2, worst


def test_bf16x3_gradients_at_batch_64():
    """The per-tensor gradient bar of the default arithmetic, justified by measurement (tools/grad_errors.py mnist 64 on
    MI355X): at B = 64 the STRICT f32 path itself sits at 1e-4 .. 1.5e-3 per tensor against the float64 oracle (d loss / d
    activations is ill-conditioned: softmax / sigmoid differences), bf16x3 at 6e-5 .. 2.6e-3.  Bars: bf16x3 < 5e-3 per
    tensor (2x tighter than the B = 6 bar above, where single examples dominate) and strict f32 < 2.5e-3.  (No fixed
    ratio between the two holds per tensor: encoder_net/conv_1/w measured 1.3e-3 vs 8e-5, decoder_net/conv_t_0/w 1.5e-3 in
    BOTH modes.)"""
    _gradients_at_batch(64)


def test_gradients_at_the_benchmarked_batch_256():
    """The same comparison at BASELINE's batch (the configuration bench.py times): every kernel the headline step
    dispatches to at B = 256 (image-resident convolutions, image / patch weight gradients, skinny GEMMs, the fused
    ResidualMLP chain) produces the gradient tensors checked here, all of them against the float64 oracle."""
    _gradients_at_batch(256)


def _gradients_at_batch(B):
    cfg, xs, x, b, eps = _inputs("mnist", B, 5)
    m = _product_model(cfg, xs)
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = O.pm_vae_loss(leaves, cfg, x, b, eps, 0)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    d = dev()
    errs = {}
    for use in (False, True):
        m.store.use_bf16 = use
        m.store.split_all()
        got = m(x.float().to(d), b.float().to(d), True, eps=eps.float().to(d))
        for key in ("reconstruction_ll", "kl", "matching_ll"):
            assert rel_err(got[key], out[key]) < (1e-4 if use else 1e-5), (use, key)
        g = [torch.full((B,), v, device=d) for v in (-1.0 / B, 1.0 / B, -1.0 / B)]
        m.zero_grad()
        m.backward(*g)
        torch.cuda.synchronize()
        gd = m.grads_dict()
        errs[use] = {n: rel_err(gd[n], grads[n]) for n in grads}
    for n in grads:
        assert errs[False][n] < 2.5e-3, (n, errs[False][n])
        assert errs[True][n] < 5e-3, (n, errs[True][n], errs[False][n])


def test_adam_kernel_matches_optax_chain():
    """pm_adam_step vs the oracle's optax restatement on identical (p, g, m, v, count)."""
    from posterior_matching_amd import ops, optim

    gen = torch.Generator().manual_seed(50)
    n, n_decay = 10007, 9000
    p, g = g32((n,), gen, 0.05), g32((n,), gen, 1e-3)        # weights ~ 1/sqrt(fan_in)
    m, v = g32((n,), gen, 1e-4), g32((n,), gen, 1e-4).abs() * 1e-3
    cfg = {"lr_schedule": {"init_value": 1e-3, "decay_rate": 0.9, "transition_steps": 5000}, "weight_decay": 1e-2}
    count = 7
    # the oracle decays `ndim != 1` leaves: a [n_decay, 1] matrix and a 1-D vector
    po = {"w": p[:n_decay].clone().reshape(-1, 1), "b": p[n_decay:].clone()}
    go = {"w": g[:n_decay].reshape(-1, 1), "b": g[n_decay:]}
    mo = {"w": m[:n_decay].clone().reshape(-1, 1), "b": m[n_decay:].clone()}
    vo = {"w": v[:n_decay].clone().reshape(-1, 1), "b": v[n_decay:].clone()}
    O.adam_update(po, go, mo, vo, count, cfg)
    d = dev()
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(1e-2),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    pd, gd, md, vd = (t.float().to(d) for t in (p, g, m, v))
    cnt = torch.tensor([count], dtype=torch.int32, device=d)
    ops.adam_step(pd, gd, md, vd, n_decay, cnt, opt.adam_cfg())
    want_p = torch.cat([po["w"].reshape(-1), po["b"]])
    upd, upd_want = pd.cpu().double() - p.float().double(), want_p - p
    # the UPDATE itself, not p (which would hide it); p is stored in f32, so the update carries
    # ~6e-8*|p|/|update| ~ 3e-5 of representation noise
    assert rel_err(upd, upd_want) < 1e-4
    assert rel_err(md, torch.cat([mo["w"].reshape(-1), mo["b"]])) < 1e-6
    assert rel_err(vd, torch.cat([vo["w"].reshape(-1), vo["b"]])) < 1e-6


@pytest.mark.parametrize("name,B", [("mnist", 16), ("gas", 128)])
def test_train_steps_match_oracle(name, B):
    """4 optimizer steps of the graph-captured HIP path vs the oracle's train_step.

    Adam's first updates are ~lr*sign(g): gradient entries at rounding-noise level flip sign in ANY
    float32 implementation, so the yardstick is the float32 run of the oracle itself - the HIP
    trajectory must stay as close to the float64 oracle as that one does (x3), and inside 1e-3."""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVAETrainStep

    cfg, xs, _, _, _ = _inputs(name, B, 7)
    m = _product_model(cfg, xs, bf16x3=False)
    p = {n: t.cpu().double() for n, t in m.params_dict().items()}
    mo = {n: torch.zeros_like(t) for n, t in p.items()}
    vo = {n: torch.zeros_like(t) for n, t in p.items()}
    p32 = {n: t.float() for n, t in p.items()}
    mo32 = {n: torch.zeros_like(t) for n, t in p32.items()}
    vo32 = {n: torch.zeros_like(t) for n, t in p32.items()}
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg.get("weight_decay", 0.0)),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(m, cfg, opt, B, xs, use_graph=(name == "mnist"), external_eps=True)   # both launch forms
    for step in range(4):
        _, _, x, b, eps = _inputs(name, B, 100 + step)
        ts.set_batch(x.float().to(dev()), b.float().to(dev()), eps.float().to(dev()))
        ts.step()
        loss, aux, _ = O.train_step(p, mo, vo, cfg, x, b, eps, step)
        loss32, aux32, _ = O.train_step(p32, mo32, vo32, cfg, x.float(), b.float(), eps.float(), step)
        got = ts.read_metrics()
        _compared()
        for key, want, w32 in (("loss", loss, loss32), ("kl", aux["kl"], aux32["kl"]),
                               ("matching_ll", aux["matching_ll"], aux32["matching_ll"]),
                               ("reconstruction_ll", aux["reconstruction_ll"], aux32["reconstruction_ll"])):
            want, w32 = float(want), float(w32)
            # steps 0-1 test the kernels; from step 2 on the trajectories of ANY two float32
            # implementations drift apart (the float32 oracle is 3e-4 off the float64 one by step 3)
            tol = max((1e-4 if step < 2 else 1e-3) * abs(want), (3 if step < 2 else 6) * abs(w32 - want))
            assert abs(got[key] - want) <= tol, (step, key, got[key], want, w32)
            assert abs(got[key] - want) <= 5e-3 * abs(want), (step, key, got[key], want)
    assert int(ts.step_dev.item()) == 4
    after = m.params_dict()
    worst = max((rel_err(after[n], p[n]), n) for n in p)
    worst32 = max((rel_err(p32[n], p[n]), n) for n in p)
    assert worst[0] < max(1e-5, 6 * worst32[0]), (worst, worst32)


@pytest.mark.parametrize("B,steps", [(64, 6), (256, 4)])
def test_bf16x3_training_trajectory_within_1e3(B, steps):
    """Default fast path (bf16x3 GEMMs, two streams, launch-plan replay from step 3): optimizer steps at B = 64 and at
    the BASELINE batch 256 (the benchmarked configuration and arithmetic); ELBO, KL and matching-LL of every step
    within 1e-3 relative of the float64 oracle trajectory."""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVAETrainStep

    cfg, xs, _, _, _ = _inputs("mnist", B, 7)
    m = _product_model(cfg, xs, bf16x3=True)
    p = {n: t.cpu().double() for n, t in m.params_dict().items()}
    mo = {n: torch.zeros_like(t) for n, t in p.items()}
    vo = {n: torch.zeros_like(t) for n, t in p.items()}
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(m, cfg, opt, B, xs, external_eps=True)
    for step in range(steps):
        _, _, x, b, eps = _inputs("mnist", B, 300 + step)
        ts.set_batch(x.float().to(dev()), b.float().to(dev()), eps.float().to(dev()))
        ts.step()
        loss, aux, _ = O.train_step(p, mo, vo, cfg, x, b, eps, step)
        got = ts.read_metrics()
        _compared()
        elbo = float(aux["reconstruction_ll"] - aux["kl"])
        assert abs((got["reconstruction_ll"] - got["kl"]) - elbo) <= 1e-3 * abs(elbo), (step, got, elbo)
        assert abs(got["kl"] - float(aux["kl"])) <= 1e-3 * abs(float(aux["kl"])), (step, got)
        assert abs(got["matching_ll"] - float(aux["matching_ll"])) <= 1e-3 * abs(float(aux["matching_ll"])), (step, got)


def test_full_batch_loss_and_properties():
    """BASELINE size (B=256): loss vs oracle, plus size-independent properties: the per-example
    outputs do not depend on batch composition, and the gradient is linear in the upstream g."""
    from posterior_matching_amd import ops

    cfg, xs, x, b, eps = _inputs("mnist", 256, 9)
    m = _product_model(cfg, xs)
    d = dev()
    xd, bd, ed = x.float().to(d), b.float().to(d), eps.float().to(d)
    out = {k_: v.clone() for k_, v in m(xd, bd, True, eps=ed).items()}
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    ref = O.pm_vae_forward(p64, cfg["model"], x, b, eps)
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(out[key], ref[key]) < 1e-5, key
    elbo = (ref["reconstruction_ll"] - ref["kl"]).mean().item()
    elbo_d = (out["reconstruction_ll"] - out["kl"]).mean().item()
    assert abs(elbo_d - elbo) < 1e-5 * abs(elbo)
    # permutation of the batch permutes the outputs
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(1))
    out_p = m(xd[perm].contiguous(), bd[perm].contiguous(), True, eps=ed[perm].contiguous())
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(out_p[key], out[key][perm.to(d)]) < 1e-5, key
    # linearity of backward in g (same forward state): grad(2g) == 2 grad(g)
    g = [torch.full((256,), v, device=d) for v in (-1 / 256, 1 / 256, -1 / 256)]
    m.zero_grad(); m.backward(*g)
    g1 = m.store.flat_g.clone()
    m.zero_grad(); m.backward(*[2 * t for t in g])
    assert rel_err(m.store.flat_g, 2 * g1) < 1e-5


def test_golden_fixture_forward_and_grads():
    """HIP path vs the committed fixture tests/golden/pm_vae_tiny.npz (self-generated by the oracle)."""
    import os

    from posterior_matching_amd.models import PosteriorMatchingVAE
    from tests.golden.make_golden import CFG, XS

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pm_vae_tiny.npz"))
    m = PosteriorMatchingVAE.from_config(CFG["model"], device="cuda:0")
    m.init(XS)
    m.load_params({k[len("param/"):]: z[k] for k in z.files if k.startswith("param/")})
    d = dev()
    B = z["x"].shape[0]
    out = m(torch.tensor(z["x"]).float().to(d), torch.tensor(z["b"]).float().to(d), True,
            eps=torch.tensor(z["eps"]).float().to(d))
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(out[key], torch.tensor(z[key])) < 1e-5, key
    g = [torch.full((B,), v, device=d) for v in (-1.0 / B, 1.0 / B, -1.0 / B)]
    m.zero_grad()
    m.backward(*g)
    gd = m.grads_dict()
    for name, got in gd.items():
        assert rel_err(got, torch.tensor(z["grad/" + name])) < 1e-4, name


def test_train_script_end_to_end(tmp_path):
    """train_pm_vae.py --config configs/pm_vae_gas.py for a few steps: runs, logs, checkpoints."""
    import json
    import os
    import pickle
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "train_pm_vae.py"), "--config",
                          os.path.join(root, "configs", "pm_vae_gas.py"), "--config.steps=30",
                          "--config.validation_freq=15", "--config.seed=3"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    run = os.path.join(tmp_path, "runs", os.listdir(os.path.join(tmp_path, "runs"))[0])
    cfg = json.load(open(os.path.join(run, "model_config.json")))
    assert cfg["latent_dim"] == 16
    lines = [json.loads(l) for l in open(os.path.join(run, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [15, 30]
    assert all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) for l in lines)
    sys.path.insert(0, root)
    state = pickle.load(open(os.path.join(run, "train_state.pkl"), "rb"))
    assert state.step == 30 and len(state.params) == 37
    # eval_pm_vae_uci.py on that run: imputation NRMSE + arbitrary-conditional log-likelihood (device-side masks)
    out = subprocess.run([sys.executable, os.path.join(root, "eval_pm_vae_uci.py"), "--run_dir", run, "--dataset", "gas",
                          "--num_instances", "128", "--batch_size", "32", "--num_samples", "16", "--num_trials", "2"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    nrmse = np.load(os.path.join(run, "uci_results", "nrmse.npy"))
    ac = np.load(os.path.join(run, "uci_results", "ac_lls.npy"))
    assert nrmse.shape == (2,) and ac.shape == (2,) and np.isfinite(nrmse).all() and np.isfinite(ac).all()


# ----------------------------------------------------------------------------------------------
# masked convolutions (PixelCNN, reference pixel_cnn.py:392-422): sub-kernel form of the index rule
# ----------------------------------------------------------------------------------------------
MASKED_CASES = [
    # B, H, Cin, Cout, full kernel, valid rows, valid cols     (pixel_cnn.py:392-422 with receptive field 3x3)
    (3, 7, 256, 128, (3, 3), 2, 3),    # vertical stack
    (3, 7, 256, 256, (3, 3), 2, 2),    # horizontal stack, 2F -> 2F
    (2, 7, 128, 128, (5, 3), 2, 3),    # vertical_stack_init
    (2, 7, 128, 128, (3, 3), 1, 3),    # horizontal_stack_up
    (2, 7, 128, 128, (3, 3), 2, 1),    # horizontal_stack_left
    (5, 16, 64, 32, (3, 3), 2, 2),     # CelebA-sized grid
    (2, 7, 24, 40, (3, 3), 2, 2),      # C not a multiple of 32: f32 gather loaders
    # the shapes and batches bench / tools run (pm_vqvae_mnist at B = 256): from B = 128 on these dispatch to the
    # image-resident kernels (image_conv_bf16 with kws > KW) instead of direct_gemm_bf16
    (256, 7, 256, 256, (3, 3), 2, 2),  # horizontal stack conv2
    (128, 7, 256, 256, (3, 3), 2, 2),
    (256, 7, 256, 128, (3, 3), 2, 3),  # vertical stack conv1 / horizontal conv1 (2F -> F)
    (128, 7, 128, 128, (5, 3), 2, 3),  # vertical_stack_init
    (256, 7, 128, 128, (3, 3), 1, 3),  # horizontal_stack_up
    (256, 7, 128, 128, (3, 3), 2, 1),  # horizontal_stack_left
    (16, 16, 256, 256, (3, 3), 2, 2),  # pm_vqvae_celeb_a at its per-GPU batch
    # the channel-pass patch form (patch_conv_cp_bf16: deep layers whose patch does not fit LDS at full depth): 2 x 3 and 2 x 2
    # masks (64-channel passes), N = 128, a ragged grid (13: half-empty tiles) and 512 channels (8 passes)
    (16, 16, 256, 128, (3, 3), 2, 3), (3, 13, 256, 64, (3, 3), 2, 2), (2, 16, 512, 96, (3, 3), 2, 3),
]


@pytest.mark.parametrize("B,H,ci,co,full,vr,vc", MASKED_CASES)
@pytest.mark.parametrize("bf16x3", [False, True])
def test_masked_conv_fwd_dgrad_wgrad(B, H, ci, co, full, vr, vc, bf16x3):
    from posterior_matching_amd import ops
    from posterior_matching_amd._lib import SplitJob
    from posterior_matching_amd.ops import ACT_NONE, LayerGeom

    gen = torch.Generator().manual_seed(B * 1000 + ci + vr * 10 + vc)
    geom = LayerGeom.masked_conv(H, H, ci, co, full[0], full[1], vr, vc)
    assert geom.weight_shape == (full[0], full[1], ci, co)
    x = g32((B, H, H, ci), gen)
    w = g32(geom.weight_shape, gen, 1.0 / math.sqrt(full[0] * full[1] * ci))
    bias, dy = g32((co,), gen, 0.1), g32((B, H, H, co), gen)
    mask = torch.zeros(full + (1, 1), dtype=F64)
    mask[:vr, :vc] = 1.0
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y = O.conv2d(xr, wr * mask, br, 1, "SAME")
    y.backward(dy)

    d = dev()
    xd, wd, bd, dyd = x.float().to(d), w.float().to(d), bias.float().to(d), dy.float().to(d)
    ws_f = ws_d = None
    if bf16x3 and ci % 32 == 0 and co % 32 == 0:
        def split(mode):
            desc = geom._desc(1, mode)
            npad = (desc.N + 31) // 32 * 32
            plane = desc.KH * desc.KW * desc.C * npad
            j = SplitJob()
            j.src_off, j.dst_off, j.plane = 0, 0, plane
            j.taps, j.C, j.N, j.npad = desc.KH * desc.KW, desc.C, desc.N, npad
            j.wts, j.wcs, j.wns, j.kw, j.kws = desc.wts, desc.wcs, desc.wns, desc.KW, desc.kws
            j.first_block, j.num_blocks = 0, plane // 1024
            out = torch.zeros(2 * plane, dtype=torch.bfloat16, device=d)
            jobs = torch.frombuffer(bytearray(bytes(j)), dtype=torch.uint8).to(d)
            ops.split_weights(wd.reshape(-1), out, jobs, 1, j.num_blocks)
            return out
        ws_f, ws_d = split("fwd"), split("dgrad")
    tol = 3e-5 if ws_f is not None else 2e-6
    yd = torch.empty((B, H, H, co), device=d)
    ops.layer_forward(geom, xd, wd, bd, yd, wsplit=ws_f)
    assert rel_err(yd, y) < tol
    dxd = torch.empty((B, H, H, ci), device=d)
    ops.layer_dgrad(geom, dyd, wd, dxd, wsplit=ws_d)
    assert rel_err(dxd, xr.grad) < tol
    dwd, dbd = torch.zeros(geom.weight_shape, device=d), torch.zeros(co, device=d)
    ops.layer_wgrad(geom, xd, dyd, dwd, dbd, bf16=bf16x3)
    assert rel_err(dwd, wr.grad) < tol                    # masked taps: exactly zero on both sides
    assert torch.equal(dwd.cpu()[vr:], torch.zeros_like(dwd.cpu()[vr:]))
    assert torch.equal(dwd.cpu()[:, vc:], torch.zeros_like(dwd.cpu()[:, vc:]))
    assert rel_err(dbd, br.grad) < 1e-5
    if bf16x3 and ws_f is not None:
        # the grouped form the train steps use (ops.WgradBatch -> pm_gather_wgrad_table): two layers of this geometry in one
        # launch, the second with its operands swapped in sign so that the groups cannot be confused
        from posterior_matching_amd.ops import WgradBatch

        batch = WgradBatch()
        buf_x, buf_dy = torch.stack([xd, -xd]).contiguous(), torch.stack([dyd, 0.5 * dyd]).contiguous()
        dw2, db2 = torch.zeros((2,) + geom.weight_shape, device=d), torch.zeros((2, co), device=d)
        for i in range(2):
            batch.add(geom, buf_x[i], buf_dy[i], dw2[i], db2[i], True)
        batch.flush()
        assert rel_err(dw2[0], wr.grad) < tol and rel_err(dw2[1], -0.5 * wr.grad) < tol
        assert rel_err(db2[0], br.grad) < 1e-5 and rel_err(db2[1], 0.5 * br.grad) < 1e-5
        assert torch.equal(dw2.cpu()[:, vr:], torch.zeros_like(dw2.cpu()[:, vr:]))


def test_diagonal_gaussian_heads_and_model():
    """DiagonalGaussian (reference distributions.py:58-84) as posterior (sample + KL) and as partial posterior
    (log_prob): kernels in isolation, then a whole PM-VAE (gas networks) with both heads diagonal."""
    from posterior_matching_amd import ops
    from posterior_matching_amd.engine import loss_cfg_from_config

    gen = torch.Generator().manual_seed(8)
    B, k = 37, 16
    prm, eps, dz, g = g32((B, 2 * k), gen), g32((B, k), gen), g32((B, k), gen), g32((B,), gen)
    pr = prm.clone().requires_grad_(True)
    loc, sc = pr[:, :k], O.softplus(pr[:, k:]) + 1e-5
    z = loc + sc * eps
    kl = O.mvn_tril_kl_to_std_normal(loc, torch.diag_embed(sc))
    ((z * dz).sum() + (kl * g).sum()).backward()
    d = dev()
    zd, kld, dp = torch.empty((B, k), device=d), torch.empty(B, device=d), torch.empty((B, 2 * k), device=d)
    ops.diag_gaussian_sample_kl_fwd(prm.float().to(d), eps.float().to(d), zd, kld)
    assert rel_err(zd, z) < 1e-6 and rel_err(kld, kl) < 2e-6
    ops.diag_gaussian_sample_kl_bwd(prm.float().to(d), eps.float().to(d), dz.float().to(d), g.float().to(d), dp)
    assert rel_err(dp, pr.grad) < 2e-6
    zz = g32((B, k), gen)
    pr2, zr = prm.clone().requires_grad_(True), zz.clone().requires_grad_(True)
    lp = O.mvn_tril_log_prob(zr, pr2[:, :k], torch.diag_embed(O.softplus(pr2[:, k:]) + 1e-5))
    (lp * g).sum().backward()
    lpd, dzo = torch.empty(B, device=d), torch.empty((B, k), device=d)
    ops.diag_gaussian_logprob_fwd(prm.float().to(d), zz.float().to(d), lpd)
    assert rel_err(lpd, lp) < 2e-6
    ops.diag_gaussian_logprob_bwd(prm.float().to(d), zz.float().to(d), g.float().to(d), dp, dzo)
    assert rel_err(dp, pr2.grad) < 2e-6 and rel_err(dzo, zr.grad) < 2e-6

    cfg = pm_vae_gas()
    cfg["model"] = dict(cfg["model"], posterior_dist="DiagonalGaussian", partial_posterior_dist="DiagonalGaussian",
                        matching_ll_stop_gradients=False)
    rng = np.random.default_rng(3)
    x, b = torch.tensor(rng.normal(size=(B, 8))), torch.tensor((rng.uniform(size=(B, 8)) < 0.5) * 1.0)
    e = torch.tensor(rng.normal(size=(B, 16)))
    m = _product_model(cfg, (8,), bf16x3=False)
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    assert list(p64) == list(O.param_shapes(cfg["model"], (8,)))
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = O.pm_vae_loss(leaves, cfg, x, b, e, 30000)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    got = m(x.float().to(d), b.float().to(d), is_training=True, eps=e.float().to(d))
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(got[key], out[key]) < 1e-5, key
    step_dev, metrics = torch.tensor([30000], dtype=torch.int32, device=d), torch.zeros(8, device=d)
    gs = [torch.empty(B, device=d) for _ in range(3)]
    ops.pmvae_loss(got["reconstruction_ll"], got["kl"], got["matching_ll"], loss_cfg_from_config(cfg, B), step_dev, metrics, *gs)
    m.zero_grad()
    m.backward(*gs)
    torch.cuda.synchronize()
    for n, gt in m.grads_dict().items():
        assert rel_err(gt, grads[n]) < 5e-5, n


# ----------------------------------------------------------------------------------------------
# ResidualMLP(layer_norm=True, dropout > 0): reference networks.py:116-131, configs/pm_vae_miniboone.py:29-39
# ----------------------------------------------------------------------------------------------
def test_layernorm_and_relu_dropout_kernels():
    """pm_layernorm_{fwd,bwd}, pm_relu_mask_{fwd,bwd} vs float64 autograd (2e-6 / 5e-6)."""
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(3)
    for R, H in ((37, 256), (5, 100), (130, 1024)):
        x = torch.randn((R, H), generator=gen, dtype=F64) * 2 + 0.3
        res, g = torch.randn((R, H), generator=gen, dtype=F64), torch.randn((R, H), generator=gen, dtype=F64)
        xr = x.clone().requires_grad_(True)
        y = O.layer_norm(xr)
        (y * g).sum().backward()
        d = dev()
        yd, od, rs = torch.empty((R, H), device=d), torch.empty((R, H), device=d), torch.empty(R, device=d)
        ops.layernorm_fwd(x.float().to(d), res.float().to(d), yd, od, rs)
        assert rel_err(yd, y) < 2e-6 and rel_err(od, y + res) < 2e-6
        var = ((x - x.mean(-1, keepdim=True)) ** 2).mean(-1)
        assert rel_err(rs, 1.0 / torch.sqrt(var + 1e-5)) < 2e-6
        dx = torch.empty((R, H), device=d)
        ops.layernorm_bwd(yd, rs, g.float().to(d), dx)
        assert rel_err(dx, xr.grad) < 5e-6
        mask = (torch.rand((R, H), generator=gen) > 0.5).double() * 2.0
        out, dxx = torch.empty((R, H), device=d), torch.empty((R, H), device=d)
        ops.relu_mask_fwd(x.float().to(d), mask.float().to(d), out)
        assert torch.equal(out.cpu(), (torch.relu(x.float()) * mask.float()))
        ops.relu_mask_bwd(x.float().to(d), mask.float().to(d), g.float().to(d), dxx)
        assert torch.equal(dxx.cpu(), (g.float() * mask.float() * (x.float() > 0)))


@pytest.mark.parametrize("L,R", [(2, 64), (4, 192), (3, 128)])
def test_mlp_chain_kernel_matches_float64(L, R):
    """pm_mlp_chain_bf16 (csrc/pm_mlp.hip): L stacked 256 -> 256 layers with the blocks' residuals, the forward form
    (relu on load, biases) and the data-gradient form (relu'(aux) multipliers), against float64; and the two-layer
    launches it replaces (pm_mlp_pair_bf16) must give the same numbers to float32 rounding."""
    from posterior_matching_amd import ops
    from posterior_matching_amd.models.core import ParamStore
    from posterior_matching_amd.ops import ACT_NONE, ACT_RELU, LayerGeom

    d, Hd = dev(), 256
    gen = torch.Generator().manual_seed(11 + L)
    g = LayerGeom.dense(Hd, Hd)
    st = ParamStore()
    for j in range(L):
        st.add(f"w{j}", (Hd, Hd), fan_in=Hd)
    hf = [st.request_split(f"w{j}", g, "fwd") for j in range(L)]
    hd = [st.request_split(f"w{j}", g, "dgrad") for j in range(L)]
    st.allocate(d)
    W = [torch.randn((Hd, Hd), generator=gen, dtype=F64) / 16 for _ in range(L)]
    st.load_dict({f"w{j}": W[j] for j in range(L)})
    bias = [torch.randn(Hd, generator=gen, dtype=F64) * 0.1 for _ in range(L)]
    aux = [torch.randn((R, Hd), generator=gen, dtype=F64) for _ in range(L)]
    x = torch.randn((R, Hd), generator=gen, dtype=F64)
    f32 = lambda t: t.float().to(d).contiguous()   # noqa: E731

    # forward form
    ref, a, res = [], torch.relu(x), x
    for j in range(L):
        o = a @ W[j] + bias[j]
        if j & 1:
            o = o + res
            res = o
        ref.append(o)
        a = torch.relu(o)
    outs = [torch.empty((R, Hd), device=d) for _ in range(L)]
    ops.mlp_chain_bf16(f32(x), [st.split_view(h) for h in hf], [f32(b) for b in bias], None, outs, ACT_RELU, ACT_RELU, ACT_NONE)
    torch.cuda.synchronize()
    for j in range(L):
        assert rel_err(outs[j], ref[j]) < 2e-5, ("fwd", j)

    # data-gradient form: out_j = (A_j W_j^T) * relu'(aux_j) [+ residual], no activations between layers
    ref, a, res = [], x, x
    for j in range(L):
        o = (a @ W[j].T) * (aux[j] > 0)
        if j & 1:
            o = o + res
            res = o
        ref.append(o)
        a = o
    outs_b = [torch.empty((R, Hd), device=d) for _ in range(L)]
    auxd = [f32(t) for t in aux]
    ops.mlp_chain_bf16(f32(x), [st.split_view(h) for h in hd], None, auxd, outs_b, ACT_NONE, ACT_NONE, ACT_RELU)
    torch.cuda.synchronize()
    for j in range(L):
        assert rel_err(outs_b[j], ref[j]) < 2e-5, ("bwd", j)

    if L % 2 == 0:      # the same arithmetic as the pair kernel, block by block
        xin = f32(x)
        for k in range(L // 2):
            o1, o2 = torch.empty((R, Hd), device=d), torch.empty((R, Hd), device=d)
            ops.mlp_pair_bf16(xin, st.split_view(hf[2 * k]), st.split_view(hf[2 * k + 1]), f32(bias[2 * k]),
                              f32(bias[2 * k + 1]), None, None, o1, o2, ACT_RELU, ACT_RELU, ACT_NONE, ACT_NONE)
            torch.cuda.synchronize()
            assert rel_err(o1, outs[2 * k]) < 1e-6 and rel_err(o2, outs[2 * k + 1]) < 1e-6
            xin = o2


def _miniboone(B, seed):
    from tests.ref_configs import pm_vae_miniboone

    cfg, xs = pm_vae_miniboone(), (43,)
    rng = np.random.default_rng(seed)
    x = torch.tensor(rng.normal(size=(B,) + xs))
    b = torch.tensor((rng.uniform(size=(B,) + xs) < 0.5).astype(np.float64))
    eps = torch.tensor(rng.normal(size=(B, 32)))
    masks = {"encoder_net": [torch.tensor((rng.uniform(size=(B, 256)) > 0.5) * 2.0) for _ in range(5)],
             "decoder_net": [torch.tensor((rng.uniform(size=(B, 256)) > 0.5) * 2.0) for _ in range(2)],
             "partial_encoder_net": [torch.tensor((rng.uniform(size=(B, 256)) > 0.5) * 2.0) for _ in range(5)]}
    return cfg, xs, x, b, eps, masks


def _tame_tril_heads(m):
    """A 32-dimensional TriL head on LayerNorm-ed features with randomly perturbed weights has diagonal entries near
    softplus(-3): the triangular solve of log q(z | x_o) then reaches 1e38 and overflows float32.  Scale the head weights
    so that the test exercises arithmetic, not overflow."""
    vals = {}
    for n, t in m.params_dict().items():
        if n in ("posterior_dist/linear/w", "partial_posterior_dist/linear/w"):
            vals[n] = t.cpu() * 0.05
        elif n in ("posterior_dist/linear/b", "partial_posterior_dist/linear/b"):
            vals[n] = t.cpu() * 0.2 + 0.5
    m.load_params(vals)


def _set_masks(m, masks):
    for name, net in (("encoder_net", m.encoder_net), ("decoder_net", m.decoder_net),
                      ("partial_encoder_net", m.partial_encoder_net)):
        if net.dropout_masks is None:
            net.dropout_masks = [t.float().to(dev()).contiguous() for t in masks[name]]
        else:
            for dst, src in zip(net.dropout_masks, masks[name]):
                dst.copy_(src.float())


@pytest.mark.parametrize("bf16x3,training", [(False, True), (True, True), (False, False)])
def test_miniboone_layernorm_dropout_model(bf16x3, training):
    """configs/pm_vae_miniboone.py network (5 / 2 residual blocks with LayerNorm after every Linear and dropout 0.5,
    TriL posterior AND - the from_config key quirk - TriL partial posterior): outputs and all gradient tensors against the
    float64 oracle with explicit keep masks (f32: 1e-5 / max(5e-5, 10x f32-CPU error); bf16x3: 1e-4 / 1e-2)."""
    from posterior_matching_amd import ops
    from posterior_matching_amd.engine import loss_cfg_from_config

    B = 19
    cfg, xs, x, b, eps, masks = _miniboone(B, 5)
    m = _product_model(cfg, xs, bf16x3=bf16x3)
    _tame_tril_heads(m)
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    assert list(p64) == list(O.param_shapes(cfg["model"], xs))
    step = 4500                                                      # cyclic beta = 1.0 at 2000 + 2500
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = O.pm_vae_loss(leaves, cfg, x, b, eps, step, masks if training else None)
    d = dev()
    _set_masks(m, masks)
    got = m(x.float().to(d), b.float().to(d), is_training=training, eps=eps.float().to(d))
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(got[key], out[key]) < (1e-4 if bf16x3 else 1e-5), key
    if not training:
        return
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    step_dev = torch.tensor([step], dtype=torch.int32, device=d)
    metrics = torch.zeros(8, device=d)
    g = [torch.empty(B, device=d) for _ in range(3)]
    ops.pmvae_loss(got["reconstruction_ll"], got["kl"], got["matching_ll"], loss_cfg_from_config(cfg, B), step_dev,
                   metrics, *g)
    assert abs(metrics[0].item() - loss.item()) < (1e-4 if bf16x3 else 1e-5) * abs(loss.item())
    m.zero_grad()
    m.backward(*g)
    torch.cuda.synchronize()
    gd = m.grads_dict()
    l32 = {n: t.float().clone().requires_grad_(True) for n, t in p64.items()}
    m32 = {k: [t.float() for t in v] for k, v in masks.items()}
    loss32, _, _ = O.pm_vae_loss(l32, cfg, x.float(), b.float(), eps.float(), step, m32)
    g32_ = dict(zip(l32, torch.autograd.grad(loss32, list(l32.values()))))
    for n in grads:
        e, e32 = rel_err(gd[n], grads[n]), rel_err(g32_[n], grads[n])
        if not bf16x3:
            assert e < max(5e-5, 10 * e32) and e < 2e-4, (n, e, e32)
        else:
            assert e < 1e-2, (n, e, e32)


def test_miniboone_train_steps_and_device_dropout():
    """3 optimizer steps (launch plan replay from step 3) with explicit masks in static buffers track the float64 oracle
    (loss 1e-4 at step 0, 1e-3 after); then the device Philox dropout: keep masks in {0, 2} with mean 1, a different
    draw every step, the same draw for the same (seed, step)."""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVAETrainStep

    B = 32
    cfg, xs, *_ = _miniboone(B, 7)
    m = _product_model(cfg, xs, bf16x3=False)
    _tame_tril_heads(m)
    p = {n: t.cpu().double() for n, t in m.params_dict().items()}
    mo, vo = {n: torch.zeros_like(t) for n, t in p.items()}, {n: torch.zeros_like(t) for n, t in p.items()}
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg["weight_decay"]),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(m, cfg, opt, B, xs, external_eps=True)
    ts.step_dev.fill_(2400)                                          # cyclic beta = 0.16 .. : a schedule value in (0, 1)
    for step in range(3):
        _, _, x, b, eps, masks = _miniboone(B, 100 + step)
        _set_masks(m, masks)
        ts.set_batch(x.float().to(dev()), b.float().to(dev()), eps.float().to(dev()))
        ts.step()
        loss, aux, _ = O.train_step(p, mo, vo, cfg, x, b, eps, 2400 + step, masks)
        got = ts.read_metrics()
        _compared()
        assert got["beta"] == pytest.approx(aux["beta"], rel=1e-6)
        for key, want in (("loss", loss), ("kl", aux["kl"]), ("matching_ll", aux["matching_ll"])):
            # step 0 tests the kernels; after one Adam update (sign-like: |u| ~ lr whatever the gradient scale) two float32
            # implementations already differ at the 1e-4 level (measured 1.15e-4 on matching_ll at step 1)
            assert abs(got[key] - float(want)) <= (1e-4 if step < 1 else 1e-3) * abs(float(want)), (step, key)
    # device dropout
    for net in (m.encoder_net, m.decoder_net, m.partial_encoder_net):
        net.dropout_masks = None
    ts.invalidate_plan()
    drawn = []
    for _ in range(3):
        ts.step()
        ts.synchronize()
        drawn.append(m.encoder_net.buf("g/mask_0", (B, 256)).clone())
        assert np.isfinite(ts.read_metrics()["loss"])
    vals = set(torch.unique(drawn[0]).cpu().tolist())
    assert vals == {0.0, 2.0} and abs(drawn[0].mean().item() - 1.0) < 0.05
    assert not torch.equal(drawn[0], drawn[1]) and not torch.equal(drawn[1], drawn[2])
    assert not torch.equal(m.encoder_net.buf("g/mask_0", (B, 256)), m.encoder_net.buf("g/mask_1", (B, 256)))
    assert not torch.equal(m.encoder_net.buf("g/mask_0", (B, 256)), m.partial_encoder_net.buf("g/mask_0", (B, 256)))


@pytest.mark.parametrize("slots", [(1, 3), (8, 37), (2, 256)])
def test_partial_sums_reduce_and_fused_adam(slots):
    """pm_reduce_partials adds the slots of every run in a fixed order (== the float64 sum to f32 rounding, and bit-identical
    when repeated); pm_adam_step_jobs (the optimizer reading the partial sums itself) gives bit for bit the parameters /
    moments of pm_reduce_partials + pm_adam_step, for runs with few slots (a thread walks them), many (the four waves deal
    them) and none, ragged run lengths, and direct contributions already sitting in g."""
    from posterior_matching_amd import ops, optim
    from posterior_matching_amd.partials import PartialSums

    d = dev()
    gen = torch.Generator().manual_seed(sum(slots))
    n = 5000                                                       # a multiple of 4, like every ParamStore buffer
    runs = [(0, 1030, slots[0]), (1032, 257, slots[1]), (2000, 2048, slots[0] + 1)]       # (g_off, count, nslots); gaps between
    g0 = torch.randn(n, generator=gen).to(d)                       # direct contributions (biases, heads)
    p0, m0, v0 = (torch.randn(n, generator=gen).to(d), torch.randn(n, generator=gen).to(d) * 0.1,
                  torch.rand(n, generator=gen).to(d) * 0.01)
    cfg = optim.adam(1e-3).adam_cfg(grad_scale=0.5)
    cfg.zero_grad = 1
    step = torch.full((1,), 7, dtype=torch.int32, device=d)

    def fill(ps):
        want = g0.double().cpu().clone()
        for off, cnt, S in runs:
            buf, src = ps.arena(off, cnt, S)
            stride = ps.entries[(off, cnt)][2]
            vals = torch.randn((S, stride), generator=torch.Generator().manual_seed(off + S))
            vals[:, cnt:] = 0.0                                    # the pad of a slot stays zero, as the kernels leave it
            buf.view(S, stride).copy_(vals)
            want[off:off + cnt] += vals[:, :cnt].double().sum(0)
        return want

    outs = []
    for fused in (False, True, False):
        g, p, m, v = g0.clone(), p0.clone(), m0.clone(), v0.clone()
        ps = PartialSums(g)
        want = fill(ps)
        if fused:
            table, njobs, _ = ps.adam_table()
            ops.adam_step_jobs(table, njobs, p, g, m, v, 3000, step, cfg)
        else:
            ps.reduce()
            assert rel_err(g, want) < 1e-6
            ops.adam_step(p, g, m, v, 3000, step, cfg)
        torch.cuda.synchronize()
        assert not ps.pending and float(g.abs().max()) == 0.0      # consumed and zeroed
        outs.append((p, m, v))
    _compared()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                   # fused == reduce + Adam, bit for bit
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)                                   # and the same bits when repeated


SPLITK_CASES = [   # kind, B, H, ci, co, k, s, padding, direction
    ("conv", 16, 8, 256, 256, 3, 1, "SAME", "fwd"), ("conv", 16, 8, 256, 256, 3, 1, "SAME", "dgrad"),
    ("conv", 2, 14, 64, 64, 5, 2, "SAME", "fwd"), ("convT", 2, 7, 64, 64, 5, 2, "SAME", "dgrad"),
    ("dense", 64, 1, 2048, 512, 1, 1, "VALID", "fwd"), ("dense", 100, 1, 1100, 200, 1, 1, "VALID", "fwd"),
    ("conv", 5, 7, 64, 128, 7, 1, "VALID", "fwd"), ("conv", 1, 32, 64, 64, 5, 1, "SAME", "dgrad"),
]


@pytest.mark.parametrize("kind,B,H,ci,co,k,s,padding,direction", SPLITK_CASES)
def test_splitk_slabs_are_bit_identical_and_agree_with_the_atomic_form(kind, B, H, ci, co, k, s, padding, direction):
    """Short-grid GEMMs split K over workgroups.  Rounds 1-3 finished them with f32 atomics into a zero-filled output (the last
    source of run-to-run differences in the data path); now every K slice stores its tile into its own slab and the epilogue
    adds the slabs in slice order (pm_gather_gemm_sk / pm_gather_gemm_bf16_sk).  Checked: the problem really takes the
    split-K form (pm_gemm_splitk_floats > 0 on at least one arithmetic), both forms against float64, two slab runs bit for bit,
    an output buffer full of garbage beforehand (no zero-fill is needed any more), bias + leaky epilogue on the forward case."""
    import ctypes as C

    from posterior_matching_amd import _lib, ops
    from posterior_matching_amd.models.core import ParamStore
    from posterior_matching_amd.ops import ACT_LEAKY, LayerGeom

    gen = torch.Generator().manual_seed(77 + B + ci)
    geom = (LayerGeom.conv(H, H, ci, co, k, s, padding) if kind == "conv" else
            LayerGeom.conv_t(H, H, ci, co, k, s, padding) if kind == "convT" else LayerGeom.dense(ci, co))
    desc = geom._desc(B, direction)
    want = {}
    for bf16 in (0, 1):
        n = C.c_longlong(0)
        rc = _lib.load().pm_gemm_splitk_floats(C.byref(desc), bf16, 1, C.byref(n))
        want[bf16] = n.value if rc == 0 else 0
    assert max(want.values()) > 0, "this case was chosen because it splits K"

    x = g32((B, geom.IH, geom.IW, ci), gen)
    w = g32(geom.weight_shape, gen, 1.0 / math.sqrt(k * k * ci))
    bias = g32((co,), gen, 0.1)
    dy = g32((B, geom.OH, geom.OW, co), gen)
    xr = x.clone().requires_grad_(True)
    if kind == "conv":
        pre = O.conv2d(xr, w, bias, s, padding)
    elif kind == "convT":
        pre = O.conv2d_transpose(xr, w, bias, s, padding)
    else:
        pre = (xr.reshape(B, ci) @ w + bias).reshape(B, 1, 1, co)
    if direction == "fwd":
        ref = O.leaky_relu(pre).detach()
    else:
        pre.backward(dy)
        ref = xr.grad

    d = dev()
    st = ParamStore()
    st.add("w", geom.weight_shape, fan_in=1)
    h = st.request_split("w", geom, direction)
    st.allocate(d)
    st.load_dict({"w": w})
    xd, wd, bd, dyd = x.float().to(d), w.float().to(d), bias.float().to(d), dy.float().to(d)

    def run(wsplit):
        out = torch.full(tuple(ref.shape), float("nan"), device=d)       # garbage in the output: nothing may depend on it
        if direction == "fwd":
            ops.layer_forward(geom, xd, wd, bd, out, out_act=ACT_LEAKY, wsplit=wsplit)
        else:
            ops.layer_dgrad(geom, dyd, wd, out, wsplit=wsplit)
        return out

    for bf16, wsplit in ((0, None), (1, st.split_view(h))):
        if bf16 and wsplit is None:
            continue
        tol = 3e-5 if bf16 else 2e-6
        a, b = run(wsplit), run(wsplit)
        assert rel_err(a, ref) < tol
        assert torch.equal(a, b)
        ops.SPLITK_SLABS = False
        try:
            c = run(wsplit)                       # the atomic form: same value to rounding, not necessarily the same bits
        finally:
            ops.SPLITK_SLABS = True
        assert rel_err(c, ref) < tol
        assert rel_err(a, c) < 1e-6


def test_vq_dw_exact_and_normal_scale_gradient_fixed_order(monkeypatch):
    """The two remaining float-atomic reductions of the BASELINE configs' train steps, now in a fixed order:
    (1) VectorQuantizerEMA's dw[:, idx] += z (reference vqvae.py:66-72 -> hk.nets.VectorQuantizerEMA): pm_vq_dw_exact, one
        workgroup per code, rows in ascending order - against float64, against the atomic form, twice the same bits, with codes
        that take no row and one code that takes most rows;
    (2) the Normal decoder's d log_scale (reference distributions.py Normal with a learned scalar scale, UCI configs):
        pm_normal_ll_bwd_det, example terms in scratch, the last workgroup adds them in example order and makes ONE +=."""
    from posterior_matching_amd import ops

    d = dev()
    gen = torch.Generator().manual_seed(31)
    for N, D, K in ((1000, 64, 256), (12544, 64, 512), (77, 24, 40)):
        z = g32((N, D), gen)
        z[: N // 2] = z[0] + 0.01 * z[: N // 2]            # half the rows crowd around one code
        emb = g32((D, K), gen)
        emb[:, 3] = z[0]
        emb[:, K - 5:] = 100.0                             # codes no row is near
        zd, ed = z.float().to(d), emb.float().to(d)
        dots = (zd @ ed).contiguous()
        outs = []
        for mode in ("exact", "exact", "atomic"):
            if mode == "atomic":
                monkeypatch.setenv("PM_VQ_DW_ATOMIC", "1")
            e2, idx = torch.empty(K, device=d), torch.empty(N, dtype=torch.int32, device=d)
            quant, cg, sq = torch.empty_like(zd), torch.empty_like(zd), torch.empty(N, device=d)
            counts, dw = torch.empty(K, device=d), torch.full((D, K), float("nan"), device=d)
            ops.vq_select(zd, ed, dots, e2, idx, quant, cg, sq, counts, dw, 0.25)
            outs.append((idx.clone(), counts.clone(), dw.clone()))
            monkeypatch.delenv("PM_VQ_DW_ATOMIC", raising=False)
        idx = outs[0][0].long().cpu()
        ref = torch.zeros((K, D), dtype=F64).index_add_(0, idx, zd.double().cpu()).t()
        assert (outs[0][1].cpu() == torch.bincount(idx, minlength=K).float()).all()
        assert idx.bincount(minlength=K).max() >= N // 2 and (idx.bincount(minlength=K) == 0).any()
        scale = zd.abs().max().item() * max(1, int(idx.bincount().max()))
        assert (outs[0][2].cpu().double() - ref).abs().max().item() < 1e-6 * scale
        assert rel_err(outs[0][2], ref) < 5e-6 and rel_err(outs[2][2], ref) < 5e-6
        assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[2][0])

    for B, D in ((700, 8), (9, 21), (256, 43)):
        loc, xs, g = g32((B, D), gen), g32((B, D), gen), g32((B,), gen)
        ls = torch.tensor(0.3, dtype=F64)
        lr_, lsr = loc.clone().requires_grad_(True), ls.clone().requires_grad_(True)
        nl = O.normal_log_prob(xs, lr_, torch.exp(lsr)).sum(-1)
        (nl * g).sum().backward()
        lsd = ls.float().reshape(()).to(d)
        runs = []
        for _ in range(3):                                    # the third reuses the scratch (its ticket went back to zero)
            dloc, dls = torch.empty((B, D), device=d), torch.full((), 0.5, device=d)
            ops.normal_ll_bwd(loc.float().to(d), xs.float().to(d), lsd, g.float().to(d), dloc, dls)
            runs.append((dloc.clone(), dls.clone()))
        tscale = (g.abs() * D).sum().item()
        assert rel_err(runs[0][0], lr_.grad) < 2e-6
        assert abs(runs[0][1].item() - 0.5 - lsr.grad.item()) < 2e-6 * tscale
        assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][0], runs[2][0])
        monkeypatch.setenv("PM_NLL_ATOMIC", "1")
        dloc, dls = torch.empty((B, D), device=d), torch.full((), 0.5, device=d)
        ops.normal_ll_bwd(loc.float().to(d), xs.float().to(d), lsd, g.float().to(d), dloc, dls)
        monkeypatch.delenv("PM_NLL_ATOMIC")
        assert abs(dls.item() - 0.5 - lsr.grad.item()) < 2e-6 * tscale and torch.equal(dloc, runs[0][0])
