"""GPU parity of the Posterior-Matching VDVAE path: HIP kernels through the C ABI vs oracle/vdvae_oracle.py."""
import math

import numpy as np
import pytest

from tests import conftest as _conftest
import torch

from oracle import pm_vae_oracle as O
from oracle import vdvae_oracle as DO
from tests.ref_configs import pm_vdvae_mnist

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


def f32d(t):
    return t.float().to(dev()).contiguous()


# ----------------------------------------------------------------------------------------------
# row-wise kernels in isolation
# ----------------------------------------------------------------------------------------------
def test_gelu_pool_resize_affine():
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(0)
    B, H, C = 3, 7, 12
    a, b = torch.randn((B, H, H, C), generator=gen, dtype=F64) * 2, torch.randn((B, H, H, 5), generator=gen, dtype=F64) * 2
    dout = torch.randn((B, H, H, C + 5), generator=gen, dtype=F64)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = DO.gelu(torch.cat([ar, br], -1))
    want.backward(dout)
    out = torch.empty((B, H, H, C + 5), device=dev())
    ops.gelu_fwd(f32d(a), f32d(b), out)
    assert rel_err(out, want) < 1e-6
    da, db = torch.ones(a.shape, device=dev()), torch.ones(b.shape, device=dev())
    ops.gelu_bwd(f32d(a), f32d(b), f32d(dout), da, db, accumulate=True)
    assert rel_err(da, ar.grad + 1) < 2e-6 and rel_err(db, br.grad + 1) < 2e-6

    for k, size in ((2, 7), (2, 3), (2, 28)):
        x = torch.randn((B, size, size, C), generator=gen, dtype=F64)
        xr = x.clone().requires_grad_(True)
        y = DO.avg_pool(xr, k)
        g = torch.randn(y.shape, generator=gen, dtype=F64)
        y.backward(g)
        yd = torch.empty(tuple(y.shape), device=dev())
        ops.avgpool_fwd(f32d(x), yd, k)
        assert rel_err(yd, y) < 1e-6
        dx = torch.empty(tuple(x.shape), device=dev())
        ops.avgpool_bwd(f32d(g), dx, k)
        assert rel_err(dx, xr.grad) < 1e-6

    for (h, Hh) in ((3, 7), (1, 3), (7, 14), (14, 28)):
        src = torch.randn((B, h, h, C + 3), generator=gen, dtype=F64)
        dst = torch.randn((B, Hh, Hh, C), generator=gen, dtype=F64)
        sr = src.clone().requires_grad_(True)
        y = dst + DO.resize_nearest(sr[..., :C], (Hh, Hh))
        g = torch.randn(y.shape, generator=gen, dtype=F64)
        y.backward(g)
        dd = f32d(dst)
        ops.resize_nearest_add(f32d(src), dd)
        assert rel_err(dd, y) < 1e-6
        ds = torch.ones(tuple(src.shape), device=dev())
        ops.resize_nearest_add_bwd(f32d(g), ds)
        assert rel_err(ds, sr.grad + 1) < 1e-6
    assert DO.nearest_index(7, 3) == [0, 0, 1, 1, 1, 2, 2]

    x = torch.randn((B * 49, C), generator=gen, dtype=F64)
    gain, bias = torch.randn(C, generator=gen, dtype=F64), torch.randn(C, generator=gen, dtype=F64)
    g = torch.randn(x.shape, generator=gen, dtype=F64)
    xr, gr, br_ = x.clone().requires_grad_(True), gain.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    (xr * gr + br_).backward(g)
    o = torch.empty(tuple(x.shape), device=dev())
    ops.affine_fwd(f32d(x), f32d(gain), f32d(bias), o)
    assert rel_err(o, x * gain + bias) < 1e-6
    dx, dg, dbb = torch.empty(tuple(x.shape), device=dev()), torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    ops.affine_bwd(f32d(x), f32d(gain), f32d(g), dx, dg, dbb)
    assert rel_err(dx, xr.grad) < 1e-6 and rel_err(dg, gr.grad) < 1e-5 and rel_err(dbb, br_.grad) < 1e-5


@pytest.mark.parametrize("Z", [16, 4])
def test_diag_sample_kl_and_diag_tril_kl(Z):
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(Z)
    B, P, W = 3, 9, 8
    R = B * P
    post = torch.randn((R, 2 * Z), generator=gen, dtype=F64)
    prior = torch.randn((R, 2 * Z + W), generator=gen, dtype=F64)
    eps, dz = torch.randn((R, Z), generator=gen, dtype=F64), torch.randn((R, Z), generator=gen, dtype=F64)
    NP = Z + Z * (Z + 1) // 2
    mp = torch.randn((R, NP), generator=gen, dtype=F64) * 0.5
    pr_, qr = prior.clone().requires_grad_(True), post.clone().requires_grad_(True)
    sq, sp = O.softplus(qr[:, Z:]) + 1e-5, O.softplus(pr_[:, Z:2 * Z]) + 1e-5
    z = qr[:, :Z] + sq * eps
    kl = DO.mvn_diag_kl(qr[:, :Z], sq, pr_[:, :Z], sp).reshape(B, P).sum(1)
    g_kl = 0.37
    ((z * dz).sum() + g_kl * kl.sum()).backward()
    zd, kld = torch.empty((R, Z), device=dev()), torch.zeros(B, device=dev())
    ops.diag_sample_kl_fwd(f32d(post), f32d(prior), f32d(eps), zd, kld, P)
    assert rel_err(zd, z) < 1e-6 and rel_err(kld, kl) < 2e-6
    dpost, dprior = torch.empty((R, 2 * Z), device=dev()), torch.zeros((R, 2 * Z + W), device=dev())
    ops.diag_sample_kl_bwd(f32d(post), f32d(prior), f32d(eps), f32d(dz), g_kl, dpost, dprior)
    assert rel_err(dpost, qr.grad) < 2e-6 and rel_err(dprior, pr_.grad) < 2e-6

    mr = mp.clone().requires_grad_(True)
    sq0 = O.softplus(post[:, Z:]) + 1e-5
    pm = DO.mvn_diag_tril_kl(post[:, :Z], sq0, mr[:, :Z], O.fill_scale_tril(mr[:, Z:])).reshape(B, P).sum(1)
    # closed form against torch.distributions (SURVEY.md 8c iv)
    td = torch.distributions
    ref = td.kl_divergence(td.MultivariateNormal(post[:, :Z], scale_tril=torch.diag_embed(sq0)),
                           td.MultivariateNormal(mp[:, :Z], scale_tril=O.fill_scale_tril(mp[:, Z:]))).reshape(B, P).sum(1)
    assert torch.allclose(pm.detach(), ref, rtol=1e-9)
    (0.21 * pm.sum()).backward()
    pmd = torch.zeros(B, device=dev())
    ops.diag_tril_kl_fwd(f32d(post), f32d(mp), pmd, Z, P)
    assert rel_err(pmd, pm) < 1e-5
    dmp = torch.empty((R, NP), device=dev())
    ops.diag_tril_kl_bwd(f32d(post), f32d(mp), 0.21, dmp, Z, P)
    assert rel_err(dmp, mr.grad) < 2e-5


def test_dmol_log_prob_mean_and_grads():
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(2)
    B, P, nm = 4, 36, 10
    R = B * P
    params = torch.randn((B, 6, 6, 3 * nm), generator=gen, dtype=F64)
    params.view(B, 6, 6, nm, 3)[..., 2] -= 2.0                             # some sharp components
    x = torch.randint(0, 256, (B, 6, 6, 1), generator=gen).double()
    x[0, 0, :3] = 0.0
    x[1, 1, :3] = 255.0                                                    # the open edge bins
    prr = params.clone().requires_grad_(True)
    ll = DO.logistic_mixture_log_prob(prr, x, nm)
    (0.3 * ll.sum()).backward()
    lld = torch.empty(B, device=dev())
    ops.dmol_ll_fwd(f32d(params), f32d(x), lld, nm, P)
    assert rel_err(lld, ll) < 1e-5
    dp = torch.empty((B, 6, 6, 3 * nm), device=dev())
    ops.dmol_ll_bwd(f32d(params), f32d(x), 0.3, dp, nm, P)
    assert rel_err(dp, prr.grad) < 2e-5
    mean = torch.empty((B, 6, 6, 1), device=dev())
    ops.dmol_mean(f32d(params), mean, nm)
    assert torch.equal(mean.cpu().double(), DO.logistic_mixture_mean(params, nm))
    # normalisation over 0..255 on the device (SURVEY.md 8c viii)
    one = torch.randn((1, 1, 1, 3 * nm), generator=gen).repeat(256, 1, 1, 1).to(dev()).contiguous()
    vals = torch.arange(256, dtype=torch.float32, device=dev()).view(256, 1, 1, 1).contiguous()
    l1 = torch.empty(256, device=dev())
    ops.dmol_ll_fwd(one, vals, l1, nm, 1)
    assert abs(torch.exp(l1.double()).sum().item() - 1.0) < 1e-5


def test_clipped_adam_with_ema_and_nonfinite_skip():
    from posterior_matching_amd import ops
    from posterior_matching_amd._lib import AdamCfg

    gen = torch.Generator().manual_seed(3)
    n, n_decay = 5000, 3000
    p = {"w": torch.randn(n_decay, 1, generator=gen, dtype=F64) * 0.05, "b": torch.randn(n - n_decay, generator=gen, dtype=F64) * 0.05}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    ema = {k: t.clone() for k, t in p.items()}
    cfg = {"lr": 1.5e-4, "gradient_clip": 2.0, "ema_rate": 0.999, "weight_decay": 0.01}
    d = dev()
    flat = lambda dd: torch.cat([dd["w"].reshape(-1), dd["b"]]).float().to(d)   # noqa: E731
    pd, md, vd, ed = flat(p), flat(m), flat(v), flat(ema)
    c = AdamCfg()
    c.b1, c.b2, c.eps, c.weight_decay, c.lr_init, c.lr_decay_rate, c.lr_transition_steps, c.grad_scale = 0.9, 0.999, 1e-8, 0.01, 1.5e-4, 1.0, 1.0, 1.0
    count, gn = torch.zeros(1, dtype=torch.int32, device=d), torch.zeros(1, device=d)
    for step, scale in enumerate([1.0, 0.01, float("nan"), 1.0]):
        g = {"w": torch.randn(n_decay, 1, generator=gen, dtype=F64) * scale, "b": torch.randn(n - n_decay, generator=gen, dtype=F64) * scale}
        applied = DO.optimizer_update(p, g, m, v, ema, int(count.item()), cfg)
        gd = flat(g)
        ops.sumsq(gd, gn)
        ops.adam_step_clip_ema(pd, gd, md, vd, ed, n_decay, count, gn, c, 2.0, 0.999, True)
        torch.cuda.synchronize()
        assert applied == (step != 2)
        assert count.item() == (step + 1 if step < 2 else step)          # the skipped step does not advance optax's count
        assert rel_err(pd, flat(p)) < 2e-6 and rel_err(ed, flat(ema)) < 2e-6 and rel_err(md, flat(m)) < 1e-5
        assert torch.isfinite(pd).all()


def test_vdvae_lr_warm_up_schedule():
    """optax.linear_schedule(0, lr, warm_up) inside the fused clip + Adam + EMA kernel (train_pm_vdvae.py:128-132): 7 updates
    across the end of a 5-step warm-up against the oracle (2e-6), and the host-side schedule object's known answers."""
    from posterior_matching_amd import ops, optim
    from posterior_matching_amd._lib import AdamCfg

    sched = optim.linear_schedule(0, 1.5e-4, 5)
    assert [sched(c) for c in (0, 1, 5, 9)] == pytest.approx([0.0, 3e-5, 1.5e-4, 1.5e-4])
    chain = optim.chain(optim.clip_by_global_norm(2.0), optim.scale_by_adam(), optim.add_decayed_weights(0.01),
                        optim.scale_by_schedule(sched), optim.scale(-1.0))
    c = chain.adam_cfg()
    assert c.lr_kind == 1 and c.lr_end == pytest.approx(1.5e-4) and c.lr_transition_steps == 5.0
    gen = torch.Generator().manual_seed(4)
    n, n_decay = 3000, 2000
    p = {"w": torch.randn(n_decay, 1, generator=gen, dtype=F64) * 0.05, "b": torch.randn(n - n_decay, generator=gen, dtype=F64) * 0.05}
    m, v = {k: torch.zeros_like(t) for k, t in p.items()}, {k: torch.zeros_like(t) for k, t in p.items()}
    ema = {k: t.clone() for k, t in p.items()}
    cfg = {"lr": 1.5e-4, "gradient_clip": 2.0, "ema_rate": 0.999, "weight_decay": 0.01, "warm_up": 5}
    d = dev()
    flat = lambda dd: torch.cat([dd["w"].reshape(-1), dd["b"]]).float().to(d)   # noqa: E731
    pd, md, vd, ed = flat(p), flat(m), flat(v), flat(ema)
    count, gn = torch.zeros(1, dtype=torch.int32, device=d), torch.zeros(1, device=d)
    for step in range(7):
        g = {"w": torch.randn(n_decay, 1, generator=gen, dtype=F64) * 0.02, "b": torch.randn(n - n_decay, generator=gen, dtype=F64) * 0.02}
        before = pd.clone()
        DO.optimizer_update(p, g, m, v, ema, step, cfg)
        gd = flat(g)
        ops.sumsq(gd, gn)
        ops.adam_step_clip_ema(pd, gd, md, vd, ed, n_decay, count, gn, c, 2.0, 0.999, True)
        torch.cuda.synchronize()
        if step == 0:
            assert torch.equal(pd, before)                                  # lr(0) = 0: the first update moves nothing
        assert rel_err(pd, flat(p)) < 2e-6 and rel_err(ed, flat(ema)) < 2e-6
    assert count.item() == 7


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
TINY = {"model": {"image_shape": (7, 7, 1), "encoder_blocks": "7x2,7d2,3x1,3d2,1x1", "decoder_blocks": "1x1,3m1,3x2,7m3,7x2",
                  "latent_dim": 4, "width": 32, "bottleneck_multiple": 0.25, "no_bias_above": 64, "num_mixtures": 10,
                  "custom_width_string": None},
        "ema_rate": 0.999, "gradient_clip": 200.0, "lr": 0.00015}


def _setup(cfg, B, seed=5, bf16x3=False, perturb=True):
    from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE

    m = PosteriorMatchingVDVAE(**cfg["model"], device="cuda:0", seed=seed)
    m.init()
    m.store.use_bf16 = bf16x3
    if perturb:      # zero-initialised pieces (prior c4, biases, x_bias) and the unit gain: move them so every path matters
        gen = torch.Generator().manual_seed(seed)
        m.load_params({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    rng = np.random.default_rng(seed)
    H = cfg["model"]["image_shape"][0]
    x = torch.tensor(np.round(rng.uniform(size=(B, H, H, 1)) * 255.0 * (rng.uniform(size=(B, H, H, 1)) < 0.4)))
    b = torch.tensor((rng.uniform(size=(B, H, H, 1)) < 0.5).astype(np.float64))
    eps = [torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)]
    return m, p64, x, b, eps


@pytest.mark.parametrize("B,H,cin,cout,mid,residual", [
    (3, 28, 192, 192, 48, True),      # encoder / resnet block at the top resolution: 7 bands of 4 rows per image
    (2, 28, 384, 32, 48, False),      # posterior block (two concatenated sources): 2 Z = 32 outputs
    (5, 14, 384, 152, 48, False),     # masked posterior: Z + Z(Z+1)/2 = 152 outputs (not a multiple of 32)
    (4, 14, 192, 224, 48, False),     # prior: 2 Z + width
    (7, 7, 192, 192, 48, True),
    (9, 3, 192, 192, 48, True),       # 3x3 grid: every tap but the centre hits padding somewhere
    (6, 1, 192, 192, 48, True),       # resolution 1: the middle convolutions are 1x1
    (2, 12, 64, 40, 24, True),        # other widths: mid < 48 leaves an n-tile half empty
])
def test_fused_block_matches_oracle_and_layerwise_path(B, H, cin, cout, mid, residual):
    """pm_vdvae_block_fwd / _bwd (a Block's four convolutions, and its four data gradients, in one launch each with the
    bottleneck activations in LDS) against the float64 oracle's Block (reference vdvae.py:263-299) and against the
    layer-by-layer path of the same engine: out, h1..h3, g1..g3, dx, dh1..dh3 within 2e-5 relative (bf16x3 forward
    values carry 5e-6), all eight weight / bias gradients within 2e-4."""
    import os

    from posterior_matching_amd import _lib, ops
    from posterior_matching_amd.models.core import ParamStore, Workspace
    from posterior_matching_amd.models.vdvae import Block

    _lib.load()
    store, ws = ParamStore(), Workspace(dev())
    blk = Block(store, ws, "blk", H, H, cin, mid, cout, H > 2)
    store.allocate(dev(), 3)
    gen = torch.Generator().manual_seed(H * 1000 + cin + cout)
    store.load_dict({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in store.to_dict("p").items()})
    p64 = {n: t.cpu().double() for n, t in store.to_dict("p").items()}
    x = torch.randn((B, H, H, cin), generator=gen, dtype=F64)
    dout = torch.randn((B, H, H, cout), generator=gen, dtype=F64)
    xr = x.clone().requires_grad_(True)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    want = DO.block(leaves, "blk", xr, H > 2, residual and cin == cout)
    grads = torch.autograd.grad((want * dout).sum(), [xr] + list(leaves.values()))
    want_dx, want_g = grads[0], dict(zip(leaves, grads[1:]))

    two_sources = cin == 384                      # posterior / masked posterior: gelu([x | activations])

    def run():
        xd = f32d(x)
        res_f = xd if residual and cin == cout else None
        if two_sources:                           # the raw-input form: gelu inside the kernel, gelu(x) written out
            out = blk.forward_raw(xd[..., :192].contiguous(), xd[..., 192:].contiguous(), res=res_f).clone()
        else:
            out = blk.forward_raw(xd, None, res=res_f).clone()
        xg_want = torch.empty_like(xd)
        ops.gelu_fwd(xd, None, xg_want)
        torch.cuda.synchronize()
        assert rel_err(blk._xg, xg_want) < 1e-6
        inter = [t.clone() for t in blk._h + blk._g]
        ops.fill_zero(store.flat_g)
        dx = torch.empty_like(xd)
        dd = f32d(dout)
        if two_sources:                           # gradient w.r.t. gelu([x | acts]); the caller's gelu_bwd splits it
            blk.backward(dd, dx)
        else:
            blk.backward(dd, dx, x_pre=xd, res=dd if residual and cin == cout else None)
        torch.cuda.synchronize()
        dhs = [ws.get(f"blk/dh{i}", (B, H, H, mid)).clone() for i in (1, 2, 3)]
        return out, inter, dx.clone(), dhs, {n: t.clone() for n, t in store.to_dict("g").items()}

    assert blk._fused() is not None
    fused = run()
    os.environ["PM_NO_VDVAE_FUSED"] = "1"
    try:
        assert blk._fused() is None
        plain = run()
    finally:
        del os.environ["PM_NO_VDVAE_FUSED"]
    assert rel_err(fused[0], want) < 2e-5 and (two_sources or rel_err(fused[2], want_dx) < 2e-5)
    if two_sources:                               # chain rule by hand: d/dx = d/dgelu(x) * gelu'(x)
        xr2 = x.clone().requires_grad_(True)
        (DO.gelu(xr2) * fused[2].double().cpu()).sum().backward()
        assert rel_err(xr2.grad, want_dx) < 2e-5
    assert rel_err(fused[0], plain[0]) < 2e-5 and rel_err(fused[2], plain[2]) < 2e-5
    for a, b_ in zip(fused[1] + fused[3], plain[1] + plain[3]):
        assert rel_err(a, b_) < 2e-5
    for n in want_g:
        err = rel_err(fused[4][n], want_g[n])
        where = ""
        if err >= 2e-4 and want_g[n].dim() == 4:      # which taps / 16-channel blocks are off
            e = (fused[4][n].double().cpu() - want_g[n]).abs() / want_g[n].abs().max()
            where = [[round(float(e[ky, kx].max()), 4) for kx in range(e.shape[1])] for ky in range(e.shape[0])]
            where = (where, [round(float(e[:, :, c:c + 16].max()), 4) for c in range(0, e.shape[2], 16)],
                     round(float(rel_err(plain[4][n], want_g[n])), 6))
        assert err < 2e-4, (n, err, where)


def test_vdvae_param_names_and_init_match_oracle():
    cfg = pm_vdvae_mnist()
    m, p64, x, b, eps = _setup(cfg, 1, perturb=False)
    shapes = DO.param_shapes(cfg["model"])
    assert {n: tuple(t.shape) for n, t in p64.items()} == shapes
    assert m.num_params == 7266942
    assert p64["decoder/gain"].eq(1).all() and p64["decoder/block_3/prior/c4/w"].eq(0).all()
    # init scale of the sqrt(1/N)-damped layers (vdvae.py:193-205): stddev = 0.88 * (1/sqrt(fan_in)) / sqrt(N)
    w = p64["decoder/block_5/resnet/c4/w"]
    assert abs(w.std().item() / (0.8796 / math.sqrt(48) / math.sqrt(20)) - 1) < 0.05
    # at init the prior's last conv is zero: prior loc = 0, scale = softplus(0) + 1e-5, h = 0 (SURVEY.md 8c ix)
    out = m(f32d(x), f32d(b), [f32d(e) for e in eps])
    pr = m.dec_blocks[0]._pr
    torch.cuda.synchronize()
    assert pr.abs().max().item() == 0.0


@pytest.mark.parametrize("bf16x3", [False, True])
def test_vdvae_forward_and_grads(bf16x3):
    B = 3
    m, p64, x, b, eps = _setup(TINY, B, bf16x3=bf16x3)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = DO.vdvae_loss(leaves, TINY, x, b, eps)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    got = m(f32d(x), f32d(b), [f32d(e) for e in eps])
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    tol = 2e-5 if not bf16x3 else 2e-4
    for k in ("reconstruction_ll", "kl", "pm_kl"):
        assert rel_err(got[k], out[k]) < tol, k
    met = m.metrics.cpu().double()
    assert abs(met[0].item() - loss.item()) < tol * abs(loss.item())
    assert abs(met[4].item() - aux["bpd"].item()) < tol * abs(aux["bpd"].item())
    assert torch.equal(m.reconstruction().cpu().double(), out["reconstruction"]) or bf16x3
    gd = m.grads_dict()
    worst = max((rel_err(gd[n], grads[n]), n) for n in grads)
    assert worst[0] < (1e-4 if not bf16x3 else 1e-2), worst


@pytest.mark.parametrize("fused,B", [(True, 5), (True, 37), (False, 5)])
def test_vdvae_fused_sample_projection_path(monkeypatch, fused, B):
    """`x += h`, sample + KL and `x += z_proj(z)` as one launch, and their gradients as one launch (pm_sample_project_fwd /
    _bwd, reference vdvae.py:558-562; the default since round 4's one-row-per-wave form) - and the three separate launches
    (PM_VDVAE_NO_SAMPLE_PROJECT=1) - against the float64 oracle.  B = 37: workgroups of 16 rows that straddle examples at the
    small resolutions, ragged last workgroups."""
    if not fused:
        monkeypatch.setenv("PM_VDVAE_NO_SAMPLE_PROJECT", "1")
    m, p64, x, b, eps = _setup(TINY, B, bf16x3=False)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = DO.vdvae_loss(leaves, TINY, x, b, eps)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    got = m(f32d(x), f32d(b), [f32d(e) for e in eps])
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    for k in ("reconstruction_ll", "kl", "pm_kl"):
        assert rel_err(got[k], out[k]) < 2e-5, k
    gd = m.grads_dict()
    worst = max((rel_err(gd[n], grads[n]), n) for n in grads)
    assert worst[0] < 1e-4, worst


def test_vdvae_train_steps_match_oracle():
    from posterior_matching_amd.engine import VDVAETrainStep

    B = 4
    cfg = dict(TINY, gradient_clip=30.0)                                    # small enough to clip at this size
    m, p64, x, b, eps = _setup(cfg, B)
    ts = VDVAETrainStep(m, cfg["lr"], B, gradient_clip=cfg["gradient_clip"], ema_rate=cfg["ema_rate"], external_eps=True)
    mo = {k: torch.zeros_like(v) for k, v in p64.items()}
    vo = {k: torch.zeros_like(v) for k, v in p64.items()}
    ema = {k: v.clone() for k, v in p64.items()}
    p32 = {k: v.float().clone() for k, v in p64.items()}
    m32, v32 = {k: torch.zeros_like(v) for k, v in p32.items()}, {k: torch.zeros_like(v) for k, v in p32.items()}
    rng = np.random.default_rng(11)
    clipped = 0
    for step in range(3):
        xb = torch.tensor(np.round(rng.uniform(size=(B, 7, 7, 1)) * 255.0))
        bb = torch.tensor((rng.uniform(size=(B, 7, 7, 1)) < 0.5).astype(np.float64))
        ee = [torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)]
        loss, aux, g = DO.train_step(p64, mo, vo, ema, cfg, xb, bb, ee, step)
        DO.train_step(p32, m32, v32, None, cfg, xb.float(), bb.float(), [e.float() for e in ee], step)
        gn = math.sqrt(sum(float((t ** 2).sum()) for t in g.values()))
        clipped += gn >= cfg["gradient_clip"]
        ts.set_batch(f32d(xb), f32d(bb), [f32d(e) for e in ee])
        ts.step()
        met = ts.read_metrics()
        _compared()
        assert abs(met["loss"] - loss.item()) < 1e-4 * abs(loss.item()), (step, met)
        assert abs(met["grad_norm"] - gn) < 1e-3 * gn
        pd, ed = m.params_dict(), ts.ema_params()
        for n in p64:
            e, e32 = rel_err(pd[n], p64[n]), rel_err(p32[n], p64[n])
            assert e < max(3e-4, 20 * e32) and e < 5e-3, (step, n, e, e32)
            assert rel_err(ed[n], ema[n]) < max(3e-4, 20 * e32), (step, n)
    assert clipped >= 1 and ts.opt_count.item() == 3 and ts.step_dev.item() == 3


@pytest.mark.parametrize("size", ["tiny", "reference"])
def test_vdvae_default_mode_trajectory_within_1e3(size):
    """Default arithmetic (bf16x3 GEMMs, companion streams, launch-plan replay from step 3): 4 optimizer steps; ELBO
    (= reconstruction_ll - kl), pm_kl and bpd of every step within 1e-3 relative of the float64 oracle trajectory - at the
    tiny sizes (B = 4) and at configs/pm_vdvae_mnist.py (7.27 M parameters, B = 2)."""
    from posterior_matching_amd.engine import VDVAETrainStep

    cfg, B = (TINY, 4) if size == "tiny" else (pm_vdvae_mnist(), 2)
    m, p64, x, b, eps = _setup(cfg, B, seed=13, bf16x3=True)
    H = cfg["model"]["image_shape"][0]
    ts = VDVAETrainStep(m, cfg["lr"], B, gradient_clip=cfg["gradient_clip"], ema_rate=cfg["ema_rate"], external_eps=True)
    mo, vo = {k: torch.zeros_like(t) for k, t in p64.items()}, {k: torch.zeros_like(t) for k, t in p64.items()}
    ema = {k: t.clone() for k, t in p64.items()}
    rng = np.random.default_rng(17)
    for step in range(4):
        xb = torch.tensor(np.round(rng.uniform(size=(B, H, H, 1)) * 255.0 * (rng.uniform(size=(B, H, H, 1)) < 0.4)))
        bb = torch.tensor((rng.uniform(size=(B, H, H, 1)) < 0.5).astype(np.float64))
        ee = [torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)]
        loss, aux, _ = DO.train_step(p64, mo, vo, ema, cfg, xb, bb, ee, step)
        ts.set_batch(f32d(xb), f32d(bb), [f32d(e) for e in ee])
        ts.step()
        met = ts.read_metrics()
        _compared()
        elbo = float(aux["reconstruction_ll"] - aux["kl"])
        assert abs((met["reconstruction_ll"] - met["kl"]) - elbo) <= 1e-3 * abs(elbo), (step, met, elbo)
        assert abs(met["pm_kl"] - float(aux["pm_kl"])) <= 1e-3 * abs(float(aux["pm_kl"])), (step, met, float(aux["pm_kl"]))
        assert abs(met["bpd"] - float(aux["bpd"])) <= 1e-3 * abs(float(aux["bpd"])), (step, met)
    assert ts.step_dev.item() == 4


def test_vdvae_reference_config_small_batch():
    """configs/pm_vdvae_mnist.py (7.27 M parameters, 20 + 20 + 20 blocks) at batch 2: outputs and sampled gradients."""
    cfg = pm_vdvae_mnist()
    B = 2
    m, p64, x, b, eps = _setup(cfg, B, seed=8)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out = DO.vdvae_loss(leaves, cfg, x, b, eps)
    names = ["encoder/stem/w", "encoder/block_3/c2/w", "masked_encoder/block_19/c4/w", "decoder/block_0/posterior/c1/w",
             "decoder/block_7/masked_posterior/c4/w", "decoder/block_12/prior/c3/w", "decoder/block_19/resnet/c4/w",
             "decoder/block_2/z_proj/w", "decoder/x_bias_3", "decoder/x_bias_28", "decoder/gain", "decoder/out_net/b"]
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names])))
    got = m(f32d(x), f32d(b), [f32d(e) for e in eps])
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    for k in ("reconstruction_ll", "kl", "pm_kl"):
        assert rel_err(got[k], out[k]) < 5e-5, k
    gd = m.grads_dict()
    for n in names:
        assert rel_err(gd[n], grads[n]) < 5e-4, (n, rel_err(gd[n], grads[n]))


def test_vdvae_impute_and_psnr_match_oracle():
    """PosteriorMatchingVDVAE.impute (reference vdvae.py:161-186) with explicit noise and the held-out PSNR
    of eval_pm_vdvae_imputation.py:123-128."""
    from posterior_matching_amd.models.vdvae import vdvae_imputation_psnr

    B, S = 3, 2
    m, p64, x, b, _ = _setup(TINY, B, seed=12)
    rng = np.random.default_rng(4)
    eps = [[torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)] for _ in range(S)]
    want = DO.vdvae_impute(p64, TINY["model"], x, b, eps)
    got = m.impute(f32d(x), f32d(b), num_samples=S, eps=[[f32d(e) for e in es] for es in eps])
    psnr = vdvae_imputation_psnr(got, f32d(x))
    torch.cuda.synchronize()
    assert got.shape == (B, S, 7, 7, 1)
    # the mean is rounded to an integer pixel value: identical unless the oracle's pre-rounding value sits on a .5 tie
    diff = (got.cpu().double() - want).abs()
    assert (diff > 0).float().mean().item() < 0.01 and diff.max().item() <= 1.0
    assert rel_err(psnr, DO.imputation_psnr(want, x)) < 1e-2
    obs = b.bool().expand(B, 7, 7, 1)
    assert torch.equal(got.cpu()[:, 0][obs], x.float()[obs])
    a = m.impute(f32d(x), f32d(b), num_samples=2, seed=3)
    c = m.impute(f32d(x), f32d(b), num_samples=2, seed=3)
    assert torch.equal(a, c) and a.min() >= 0 and a.max() <= 255


@_conftest.compares
def test_vdvae_is_log_probs_and_sample_match_oracle():
    """PosteriorMatchingVDVAE.is_log_probs (reference vdvae.py:96-146, forward_lls / sample_lls :609-660,725-754) and
    .sample (:148-159, forward_prior) with explicit noise.  The estimator sums O(100)-magnitude log terms over every
    latent site before exponentiating: the yardstick is the float32-CPU oracle's own distance from float64."""
    B, S = 3, 4
    m, p64, x, b, _ = _setup(TINY, B, seed=14)
    rng = np.random.default_rng(5)
    eps = [[torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)] for _ in range(S)]
    eps_m = [[torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)] for _ in range(S)]
    want_px, want_pxu = DO.vdvae_is_log_probs(p64, TINY["model"], x, b, eps, eps_m)
    p32 = {n: t.float() for n, t in p64.items()}
    f32_px, f32_pxu = DO.vdvae_is_log_probs(p32, TINY["model"], x.float(), b.float(), [[e.float() for e in es] for es in eps],
                                            [[e.float() for e in es] for es in eps_m])
    got_px, got_pxu = m.is_log_probs(f32d(x), f32d(b), num_samples=S, eps=[[f32d(e) for e in es] for es in eps],
                                     eps_masked=[[f32d(e) for e in es] for es in eps_m])
    torch.cuda.synchronize()
    scale = max(1.0, want_px.abs().max().item())
    tol_px = max(1e-4 * scale, 20 * (f32_px.double() - want_px).abs().max().item())
    tol_pxu = max(1e-4 * scale, 20 * (f32_pxu.double() - want_pxu).abs().max().item())
    assert (got_px.cpu().double() - want_px).abs().max().item() < tol_px
    assert (got_pxu.cpu().double() - want_pxu).abs().max().item() < tol_pxu
    # device-noise path: finite and reproducible for a seed
    a1 = [t.clone() for t in m.is_log_probs(f32d(x), f32d(b), num_samples=3, seed=9)]
    a2 = m.is_log_probs(f32d(x), f32d(b), num_samples=3, seed=9)
    # same noise for a seed; the per-example sums are float atomics, so the order of additions may differ between calls
    assert all(torch.isfinite(t).all() for t in a1)
    assert torch.allclose(a1[0], a2[0], atol=5e-3) and torch.allclose(a1[1], a2[1], atol=5e-3)
    # unconditional samples
    N = 5
    eps_p = [torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(N)]
    want = DO.vdvae_sample(p64, TINY["model"], eps_p)
    got = m.sample(N, eps=[f32d(e) for e in eps_p])
    diff = (got.cpu().double() - want).abs()
    assert got.shape == (N, 7, 7, 1) and (diff > 0).float().mean().item() < 0.02 and diff.max().item() <= 1.0
    s1, s2 = m.sample(4, seed=2), m.sample(4, seed=2)
    assert torch.equal(s1, s2) and s1.min() >= 0 and s1.max() <= 255


def test_train_and_eval_vdvae_scripts_end_to_end(tmp_path):
    """train_pm_vdvae.py for a few steps on a small network, then eval_pm_vdvae_imputation.py on its checkpoint."""
    import json
    import os
    import pickle
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(script, *argv):
        out = subprocess.run([sys.executable, os.path.join(root, script), *argv], cwd=tmp_path, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        return out.stdout

    run("train_pm_vdvae.py", "--config", os.path.join(root, "configs", "pm_vdvae_mnist.py"), "--config.steps=6",
        "--device_masks", "--config.validation_freq=3", "--config.seed=2", "--config.model.width=32", "--config.model.latent_dim=4",
        "--config.data.train_batch_size=4", "--config.data.val_batch_size=4",
        "--config.model.encoder_blocks=28x1,28d2,14x1,14d2,7x1,7d2,3x1,3d2,1x1",
        "--config.model.decoder_blocks=1x1,3m1,3x1,7m3,7x1,14m7,14x1,28m14,28x1")
    rd = os.path.join(tmp_path, "runs", os.listdir(os.path.join(tmp_path, "runs"))[0])
    lines = [json.loads(l) for l in open(os.path.join(rd, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [3, 6]
    assert all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) and l["learning_rate"] == 0.00015 for l in lines)
    imp = np.load(os.path.join(rd, "tb", "imputations_6.npy"))
    assert imp.dtype == np.uint8 and imp.shape == (4, 28, 28 * 10, 1)
    sys.path.insert(0, root)
    st = pickle.load(open(os.path.join(rd, "train_state.pkl"), "rb"))
    assert st.step == 6 and st.opt_state["count"] == 6 and st.ema_params is not None
    k = "decoder/block_0/resnet/c1/w"
    assert not torch.equal(st.ema_params[k], st.params[k])
    out = run("eval_pm_vdvae_imputation.py", "--run_dir", rd, "--num_instances", "8", "--batch_size", "4", "--num_samples", "2")
    res = json.loads(out.strip().splitlines()[-1])
    assert res["num_instances"] == 8 and np.isfinite(res["mean_psnr"]) and 0.0 < res["mean_psnr"] < 60.0
    out = run("eval_pm_vdvae_likelihood.py", "--run_dir", rd, "--num_instances", "8", "--batch_size", "4", "--num_samples", "3",
              "--num_trials", "2")
    assert "BPD:" in out and "AC LL:" in out
    lk = os.path.join(rd, "likelihood_results")
    x_lls, bpd = np.load(os.path.join(lk, "x_lls.npy")), np.load(os.path.join(lk, "bpd.npy"))
    assert x_lls.shape == (2, 8) and np.isfinite(x_lls).all() and (bpd > 0).all()
    assert np.load(os.path.join(lk, "xo_lls.npy")).shape == (2, 8)
