"""GPU parity of the PixelCNN partial posterior and the PM-VQVAE (stage 2) step: HIP path through
the C ABI vs oracle/pixel_cnn_oracle.py (float64).  Tolerances stated per test."""
import math

import numpy as np
import pytest

from tests import conftest as _conftest
import torch

from oracle import pixel_cnn_oracle as PO
from oracle import vqvae_oracle as VO
from tests.ref_configs import pm_vqvae_mnist, vqvae_mnist

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
def test_concat_elu_gate_rows_sum_elu():
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(0)
    R, Ca, Cb, P = 3 * 49, 24, 40, 49
    a, b = torch.randn((R, Ca), generator=gen, dtype=F64), torch.randn((R, Cb), generator=gen, dtype=F64)
    drop = (torch.rand((R, 2 * (Ca + Cb)), generator=gen) > 0.5).double() * 2.0
    dout = torch.randn((R, 2 * (Ca + Cb)), generator=gen, dtype=F64)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = PO.concat_elu(torch.cat([ar, br], -1)) * drop
    want.backward(dout)
    out = torch.empty((R, 2 * (Ca + Cb)), device=dev())
    ops.concat_elu_fwd(f32d(a), f32d(b), f32d(drop), out)
    assert rel_err(out, want) < 1e-6
    da, db = torch.ones((R, Ca), device=dev()), torch.ones((R, Cb), device=dev())
    ops.concat_elu_bwd(f32d(a), f32d(b), f32d(drop), f32d(dout), da, db, accumulate=True)
    assert rel_err(da, ar.grad + 1.0) < 1e-6 and rel_err(db, br.grad + 1.0) < 1e-6
    ops.concat_elu_bwd(f32d(a), None, None, f32d(dout[:, :2 * Ca]), da, None, accumulate=False)
    a2 = a.clone().requires_grad_(True)
    PO.concat_elu(a2).backward(dout[:, :2 * Ca])
    assert rel_err(da, a2.grad) < 1e-6

    F, B = 32, 3
    y, h = torch.randn((R, 2 * F), generator=gen, dtype=F64), torch.randn((B, 2 * F), generator=gen, dtype=F64)
    inp, g = torch.randn((R, F), generator=gen, dtype=F64), torch.randn((R, F), generator=gen, dtype=F64)
    yr, hr = y.clone().requires_grad_(True), h.clone().requires_grad_(True)
    x = yr.reshape(B, P, 2 * F) + hr[:, None, :]
    wantg = inp.reshape(B, P, F) + torch.sigmoid(x[..., F:]) * x[..., :F]
    wantg.backward(g.reshape(B, P, F))
    og = torch.empty((R, F), device=dev())
    ops.gate_fwd(f32d(y), f32d(h), f32d(inp), og, P)
    assert rel_err(og, wantg.reshape(R, F)) < 1e-6
    dy = torch.empty((R, 2 * F), device=dev())
    ops.gate_bwd(f32d(y), f32d(h), f32d(g), dy, P)
    assert rel_err(dy, yr.grad) < 2e-6
    dh = torch.empty((B, 2 * F), device=dev())
    ops.rows_sum(dy, dh, P)
    assert rel_err(dh, hr.grad) < 2e-6
    gs = torch.ones((B, 2 * F), device=dev())
    ops.groups_sum(dy.view(P, B * 2 * F)[:0 + P].contiguous(), gs.view(-1), P, accumulate=True)
    assert rel_err(gs.view(-1), dy.view(P, -1).double().sum(0) + 1.0) < 1e-6

    e = torch.empty((R, Ca), device=dev())
    ops.elu_fwd(f32d(a), e)
    assert rel_err(e, PO.elu(a)) < 1e-6
    de = torch.empty((R, Ca), device=dev())
    ops.elu_bwd(f32d(a), f32d(dout[:, :Ca]), de)
    a3 = a.clone().requires_grad_(True)
    PO.elu(a3).backward(dout[:, :Ca])
    assert rel_err(de, a3.grad) < 1e-6


@pytest.mark.parametrize("B,N,P", [(256, 256, 49), (16, 512, 256), (128, 128, 7), (3, 64, 5), (16, 256, 256), (130, 132, 3)])
def test_rows_sum_forms(B, N, P):
    """pm_rows_sum at the shapes the benchmarked steps run: (256, 256, 49) = pm_vqvae_mnist at B = 256 and (16, 512, 256)
    (the one-workgroup-per-example rows_sum_v4 form, taken when N >= 128 and B * ceil(N / 256) >= 128), (16, 256, 256) =
    pm_vqvae_celeb_a at its per-GPU batch (the 32-column form), and ragged sizes on both sides of the switch; vs float64."""
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(B + N + P)
    x = torch.randn((B, P, N), generator=gen, dtype=F64)
    out = torch.full((B, N), 7.0, device=dev())
    ops.rows_sum(f32d(x).view(B * P, N), out, P)
    assert rel_err(out, x.sum(1)) < 2e-6


def test_embed_categorical_and_dropout_mask(monkeypatch):
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(1)
    # (256, 49) = pm_vqvae_mnist at B = 256, (16, 256, K = 512) = pm_vqvae_celeb_a; K = 7: hot entries, F not a power of two
    for B, P, K, F in ((5, 49, 256, 128), (256, 49, 256, 128), (16, 256, 512, 128), (40, 30, 7, 300)):
        R = B * P
        idx = torch.randint(0, K, (R,), generator=gen)
        table = torch.randn((K, F), generator=gen, dtype=F64)
        out = torch.empty((R, F), device=dev())
        idd = idx.to(torch.int32).to(dev())
        ops.embed_fwd(idd, f32d(table), out)
        assert torch.equal(out.cpu(), table.float()[idx])
        dout = torch.randn((R, F), generator=gen, dtype=F64)
        dt = torch.ones((K, F), device=dev())                           # accumulates
        ops.embed_bwd(idd, f32d(dout), dt)
        want = torch.ones((K, F), dtype=F64).index_add_(0, idx, dout)
        assert rel_err(dt, want) < 1e-6, (B, P, K, F)
        # the fixed-order form (pm_embed_bwd_sorted, the default) twice: the same bits; the fixed-point atomic form
        # (pm_embed_bwd_exact) and the f32-atomic one agree with it to rounding
        dt2 = torch.ones((K, F), device=dev())
        ops.embed_bwd(idd, f32d(dout), dt2)
        assert torch.equal(dt, dt2)
        for knob in ("PM_EMBED_FIXEDPOINT", "PM_EMBED_ATOMIC"):
            monkeypatch.setenv(knob, "1")
            dt3 = torch.ones((K, F), device=dev())
            ops.embed_bwd(idd, f32d(dout), dt3)
            monkeypatch.delenv(knob)
            assert rel_err(dt3, want) < 1e-6 and rel_err(dt3, dt) < 1e-6, (knob, B, P, K, F)
    B, P, K, F = 5, 49, 256, 128
    R = B * P
    idx = torch.randint(0, K, (R,), generator=gen)
    idd = idx.to(torch.int32).to(dev())

    logits = torch.randn((R, K), generator=gen, dtype=F64) * 3
    g = torch.randn((B,), generator=gen, dtype=F64)
    lr = logits.clone().requires_grad_(True)
    ll = torch.log_softmax(lr, -1).gather(-1, idx[:, None]).reshape(B, P).sum(1)
    (ll * g).sum().backward()
    lse, lld = torch.empty(R, device=dev()), torch.empty(B, device=dev())
    ops.categorical_ll_fwd(f32d(logits), idd, lse, lld, P)
    assert rel_err(lld, ll) < 1e-6
    dl = torch.empty((R, K), device=dev())
    ops.categorical_ll_bwd(f32d(logits), idd, lse, f32d(g), dl, P)
    assert rel_err(dl, lr.grad) < 2e-6
    met, gl = torch.zeros(8, device=dev()), torch.empty(B, device=dev())
    ops.neg_mean_loss(lld, 1.0 / B, met, gl)
    assert abs(met[0].item() + ll.mean().item()) < 1e-5 * abs(ll.mean().item()) and torch.allclose(gl.cpu(), torch.full((B,), -1.0 / B))

    m = torch.empty(1 << 20, device=dev())
    step = torch.tensor([3], dtype=torch.int32, device=dev())
    ops.dropout_mask(m, 0.5, 1234, step, stream_id=2)
    vals = set(torch.unique(m).cpu().tolist())
    assert vals == {0.0, 2.0} and abs(m.mean().item() - 1.0) < 0.01      # E[mask] = 1
    m2 = torch.empty_like(m)
    ops.dropout_mask(m2, 0.5, 1234, step, stream_id=2)
    assert torch.equal(m, m2)                                              # counter-based: reproducible
    ops.dropout_mask(m2, 0.5, 1234, step, stream_id=3)
    assert not torch.equal(m, m2)


@pytest.mark.parametrize("B,P,F,cond", [(256, 49, 128, True), (130, 9, 32, True), (128, 49, 128, False), (129, 5, 256, True)])
def test_gate_bwd_with_rows_sum(B, P, F, cond):
    """pm_gate_bwd_rows_sum (the train step's form at B >= 128) vs float64: dy of the gated residual (pixel_cnn.py:455-460)
    and dh = its sum over the positions of each image (the conditional projection's gradient, :565-569)."""
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(B + P + F)
    R = B * P
    y, h = torch.randn((R, 2 * F), generator=gen, dtype=F64), torch.randn((B, 2 * F), generator=gen, dtype=F64)
    g = torch.randn((R, F), generator=gen, dtype=F64)
    yr, hr = y.clone().requires_grad_(True), h.clone().requires_grad_(True)
    x = yr.reshape(B, P, 2 * F) + (hr[:, None, :] if cond else 0.0)
    (torch.sigmoid(x[..., F:]) * x[..., :F]).backward(g.reshape(B, P, F))
    assert ops.gate_bwd_rows_sum_ok(f32d(g), B)
    dy, dh = torch.empty((R, 2 * F), device=dev()), torch.full((B, 2 * F), 7.0, device=dev())
    ops.gate_bwd_rows_sum(f32d(y), f32d(h) if cond else None, f32d(g), dy, dh, P)
    assert rel_err(dy, yr.grad) < 2e-6
    assert rel_err(dh, hr.grad if cond else yr.grad.reshape(B, P, 2 * F).sum(1)) < 2e-6


@pytest.mark.parametrize("B,P,F,cond", [(256, 49, 128, True), (16, 256, 128, True), (5, 9, 12, False)])
@_conftest.compares
def test_gate_forward_with_the_next_blocks_concat_elu(B, P, F, cond):
    """pm_gate_fwd_ce (the train step's form) against the two launches it replaces - pm_gate_fwd, then pm_concat_elu_fwd of its
    output, both checked against the oracle by the PixelCNN parity tests: BIT-identical `out` and `ce`; the whole-network tests
    run it as well (their logits / gradients are compared with the oracle's)."""
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(B + F)
    d = dev()
    y = torch.randn((B * P, 2 * F), generator=gen).to(d)
    h = torch.randn((B, 2 * F), generator=gen).to(d) if cond else None
    x = torch.randn((B * P, F), generator=gen).to(d)
    out_w, ce_w = torch.empty_like(x), torch.empty((B * P, 2 * F), device=d)
    ops.gate_fwd(y, h, x, out_w, P)
    ops.concat_elu_fwd(out_w, None, None, ce_w)
    out_g, ce_g = torch.empty_like(x), torch.empty((B * P, 2 * F), device=d)
    assert ops.gate_fwd_ce_ok(y, h, x, out_g, ce_g)
    ops.gate_fwd_ce(y, h, x, out_g, ce_g, P)
    assert torch.equal(out_g, out_w) and torch.equal(ce_g, ce_w)


@pytest.mark.parametrize("R,Ca,Cb", [(12544, 128, 0), (245, 64, 32), (4096, 128, 128)])
@_conftest.compares
def test_concat_elu_with_the_keep_mask_drawn_in_place(R, Ca, Cb):
    """pm_concat_elu_{fwd,bwd}_philox (the train step's form: hk.dropout's keep mask never exists in HBM) against the
    two-launch form it replaces - pm_dropout_mask into a tensor, then pm_concat_elu_{fwd,bwd} with that tensor, which the
    PixelCNN parity tests check against the oracle with explicit masks: BIT-identical outputs and gradients, for several
    steps / stream ids (same Philox counters), with and without the second input, accumulate / add_a."""
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(R + Ca)
    d = dev()
    a = torch.randn((R, Ca), generator=gen).to(d)
    b = torch.randn((R, Cb), generator=gen).to(d) if Cb else None
    dout = torch.randn((R, 2 * (Ca + Cb)), generator=gen).to(d)
    add_a = torch.randn((R, Ca), generator=gen).to(d)
    for step, sid, rate in ((0, 0, 0.5), (7, 3, 0.5), (123456, 31, 0.1)):
        step_dev = torch.tensor([step], dtype=torch.int32, device=d)
        mask = torch.empty((R, 2 * (Ca + Cb)), device=d)
        ops.dropout_mask(mask, rate, 99, step_dev, stream_id=sid)
        pd = ops.PhiloxDrop(rate, 99, step_dev, sid)
        assert ops.PhiloxDrop.usable(a, b)
        want, got = torch.empty_like(dout), torch.empty_like(dout)
        ops.concat_elu_fwd(a, b, mask, want)
        ops.concat_elu_fwd(a, b, pd, got)
        assert torch.equal(got, want) and 0.0 < (got == 0).float().mean() < 1.0
        for acc, add in ((False, None), (True, add_a)):
            da_w, da_g = torch.ones((R, Ca), device=d), torch.ones((R, Ca), device=d)
            db_w = torch.ones((R, Cb), device=d) if Cb else None
            db_g = torch.ones((R, Cb), device=d) if Cb else None
            ops.concat_elu_bwd(a, b, mask, dout, da_w, db_w, accumulate=acc, add_a=add)
            ops.concat_elu_bwd(a, b, pd, dout, da_g, db_g, accumulate=acc, add_a=add)
            assert torch.equal(da_g, da_w) and (not Cb or torch.equal(db_g, db_w))
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------------------------
# the network
# ----------------------------------------------------------------------------------------------
SMALL = {"num_indices": 24, "image_shape": (5, 5), "num_resnet": 2, "num_hierarchies": 1, "num_filters": 32, "dropout": 0.5}


def _build_pixelcnn(cfg, cond_dim, seed=3, bf16x3=False):
    from posterior_matching_amd import _lib
    from posterior_matching_amd.models.core import ParamStore, Workspace
    from posterior_matching_amd.models.pixel_cnn import PixelCNN

    _lib.load()
    store, ws = ParamStore(), Workspace(dev())
    pc = PixelCNN(**cfg)
    pc.ws = ws
    pc.build(store, "pc", cond_dim)
    store.allocate(dev(), seed)
    store.use_bf16 = bf16x3
    gen = torch.Generator().manual_seed(seed)
    vals = {}
    for n, t in store.to_dict("p").items():           # biases are zero at init: move them; tame the N(0,1) projections
        v = t.cpu()
        if n.endswith("/b"):
            v = 0.05 * torch.randn(v.shape, generator=gen)
        if n.endswith("/cond/w") or n.endswith("/embeddings"):
            v = v * 0.3
        vals[n] = v
    store.load_dict(vals)
    return pc, store


@pytest.mark.parametrize("training,bf16x3", [(False, False), (True, False), (True, True)])
def test_pixelcnn_log_prob_and_grads(training, bf16x3):
    cfg, cd, B = SMALL, 16, 3
    pc, store = _build_pixelcnn(cfg, cd, bf16x3=bf16x3)
    p64 = {n: t.cpu().double() for n, t in store.to_dict("p").items()}
    assert {n: tuple(t.shape) for n, t in p64.items()} == PO.pixel_cnn_param_shapes("pc", cfg, cd)
    gen = torch.Generator().manual_seed(5)
    H, W = cfg["image_shape"]
    idx = torch.randint(0, cfg["num_indices"], (B, H, W), generator=gen)
    cond = torch.randn((B, cd), generator=gen, dtype=F64)
    g = torch.randn((B,), generator=gen, dtype=F64)
    F = cfg["num_filters"]
    masks = [((torch.rand((B, H, W, 2 * F), generator=gen) > 0.5).double() * 2.0) for _ in range(4 * cfg["num_resnet"])] \
        if training else None
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    cr = cond.clone().requires_grad_(True)
    lp = PO.pixel_cnn_log_prob(leaves, "pc", idx, cfg, cr, masks)
    grads = torch.autograd.grad((lp * g).sum(), list(leaves.values()) + [cr])
    gd_want = dict(zip(leaves, grads[:-1]))

    got = pc.log_prob(idx.to(torch.int32).to(dev()), training=training, conditional_input=f32d(cond),
                      dropout_masks=[f32d(m) for m in masks] if masks else None)
    tol = 1e-5 if not bf16x3 else 1e-4
    assert rel_err(got, lp) < tol
    from posterior_matching_amd import ops
    ops.fill_zero(store.flat_g)
    dcond = pc.backward(f32d(g))
    torch.cuda.synchronize()
    assert rel_err(dcond, grads[-1]) < (5e-5 if not bf16x3 else 5e-3)
    for n, gt in store.to_dict("g").items():
        e = rel_err(gt, gd_want[n])
        assert e < (5e-5 if not bf16x3 else 5e-3), (n, e)
    # masked taps never receive gradient
    gw = store.to_dict("g")["pc/down_0/horizontal/conv2/w"].cpu()
    assert torch.equal(gw[2], torch.zeros_like(gw[2])) and torch.equal(gw[:, 2], torch.zeros_like(gw[:, 2]))


@_conftest.compares
def test_pixelcnn_is_autoregressive_on_device():
    """bit-exact: logits at raster positions <= (r, c) do not depend on the index at (r, c)."""
    cfg, cd, B = dict(SMALL, image_shape=(7, 7)), 16, 2
    pc, store = _build_pixelcnn(cfg, cd)
    gen = torch.Generator().manual_seed(9)
    idx = torch.randint(0, 24, (B, 7, 7), generator=gen).to(torch.int32).to(dev())
    cond = torch.randn((B, cd), generator=gen).to(dev())
    base = pc.logits(idx, False, cond).clone()
    for (r, c) in [(0, 0), (3, 3), (6, 6), (2, 5)]:
        idx2 = idx.clone()
        idx2[:, r, c] = (idx2[:, r, c] + 1) % 24
        l2 = pc.logits(idx2, False, cond)
        torch.cuda.synchronize()
        same = (l2 == base).all(-1)[0].cpu()
        for rr in range(7):
            for cc in range(7):
                if (rr, cc) <= (r, c):
                    assert same[rr, cc], ((r, c), (rr, cc))
        if (r, c) != (6, 6):
            assert not same.all()


@_conftest.compares
def test_pixelcnn_unconditional_sampling_matches_oracle():
    """PixelCNN.sample without conditional_input (reference pixel_cnn.py:82-100): exact index grids under explicit Gumbel
    noise; device noise: reproducible per seed; a network built WITH a conditional_dim refuses to sample without input."""
    cfg = dict(SMALL, image_shape=(4, 4))
    pc, store = _build_pixelcnn(cfg, None)
    p64 = {n: t.cpu().double() for n, t in store.to_dict("p").items()}
    assert not any("/cond/" in n for n in p64)
    n, K, P = 5, cfg["num_indices"], 16
    rng = np.random.default_rng(11)
    gumbel = torch.tensor(-np.log(-np.log(rng.uniform(1e-6, 1 - 1e-6, size=(P, n, K)))))
    want = PO.pixel_cnn_sample_unconditional(p64, "pc", cfg, n, gumbel)
    got = pc.sample(seed=0, sample_shape=n, gumbel=f32d(gumbel))
    torch.cuda.synchronize()
    assert got.shape == (n, 4, 4) and got.dtype == torch.int32 and torch.equal(got.cpu().long(), want)
    s1, s2, s3 = pc.sample(seed=3, sample_shape=(2, 3)), pc.sample(seed=3, sample_shape=(2, 3)), pc.sample(seed=4, sample_shape=(2, 3))
    assert s1.shape == (2, 3, 4, 4) and torch.equal(s1, s2) and not torch.equal(s1, s3)
    assert pc.sample(seed=1).shape == (4, 4) and int(s1.min()) >= 0 and int(s1.max()) < K
    pc2, _ = _build_pixelcnn(SMALL, 16)
    with pytest.raises(ValueError):
        pc2.sample(seed=0, sample_shape=2)


# ----------------------------------------------------------------------------------------------
# stage 2 of PM-VQVAE: frozen VQ-VAE + partial encoder + PixelCNN
# ----------------------------------------------------------------------------------------------
TINY_VQ = {"embedding_dim": 32, "num_embeddings": 24, "hidden_units": 32, "residual_hidden_units": 32,
           "residual_blocks": 1, "decay": 0.99, "use_ema": True, "commitment_cost": 0.25, "output_channels": 1}
TINY_CFG = {"pixel_cnn": {"image_shape": (3, 3), "num_resnet": 2, "num_hierarchies": 1, "num_filters": 32, "dropout": 0.5},
            "conditional_dim": 64, "lr_schedule": {"init_value": 3e-4, "decay_rate": 0.999995, "transition_steps": 1}}


def _stage2(cfg, vq_cfg, xs, B, seed=4, bf16x3=False):
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVQVAETrainStep
    from posterior_matching_amd.models.pixel_cnn import PixelCNN
    from posterior_matching_amd.models.vqvae import VQVAE, VQVAEPartialEncoder

    vq = VQVAE(**vq_cfg, device="cuda:0", seed=seed)
    vq.init(xs)
    vq.store.use_bf16 = False           # frozen encoder: indices must match the oracle's
    pc_cfg = dict(cfg["pixel_cnn"], num_indices=vq_cfg["num_embeddings"])
    penc, pcnn = VQVAEPartialEncoder(cfg["conditional_dim"], vq_cfg), PixelCNN(**pc_cfg)
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVQVAETrainStep(vq, penc, pcnn, opt, B, xs, seed=seed, external_dropout=True)
    ts.store.use_bf16 = bf16x3
    gen = torch.Generator().manual_seed(seed)
    vals = {}
    for n, t in ts.store.to_dict("p").items():
        v = t.cpu()
        if n.endswith("/b"):
            v = 0.05 * torch.randn(v.shape, generator=gen)
        if n.endswith("/cond/w") or n.endswith("/embeddings"):
            v = v * 0.3
        vals[n] = v
    ts.store.load_dict(vals)
    p64 = {n: t.cpu().double() for n, t in ts.store.to_dict("p").items()}
    vq64 = {n: t.cpu().double() for n, t in vq.params_dict().items()}
    sd = vq.state_dict()
    st64 = {"vq/embeddings": sd["embeddings"].cpu().double()}
    for name in ("ema_cluster_size", "ema_dw"):
        st64[f"vq/{name}/hidden"] = sd[f"{name}/hidden"].cpu().double()
        st64[f"vq/{name}/average"] = sd[f"{name}/average"].cpu().double()
        st64[f"vq/{name}/counter"] = torch.tensor(0)
    return ts, p64, vq64, st64


def _batch(rng, B, xs):
    x = rng.uniform(size=(B,) + xs) * (rng.uniform(size=(B,) + xs) < 0.3)
    b = (rng.uniform(size=(B,) + xs[:-1] + (1,)) < 0.5).astype(np.float64)
    return torch.tensor(x), torch.tensor(b)


def _masks(rng, cfg, B):
    H, W = cfg["pixel_cnn"]["image_shape"]
    F, R = cfg["pixel_cnn"]["num_filters"], cfg["pixel_cnn"]["num_resnet"]
    return [torch.tensor((rng.uniform(size=(B, H, W, 2 * F)) > 0.5) * 2.0) for _ in range(4 * R)]


def test_stage2_param_names_match_oracle():
    ts, p64, vq64, st64 = _stage2(TINY_CFG, TINY_VQ, (12, 12, 1), 2)
    want = PO.init_params(TINY_CFG, TINY_VQ, 2)
    assert {n: tuple(t.shape) for n, t in p64.items()} == {n: tuple(t.shape) for n, t in want.items()}


@pytest.mark.parametrize("bf16x3", [False, True])
def test_stage2_loss_and_grads(bf16x3):
    B, xs = 4, (12, 12, 1)
    ts, p64, vq64, st64 = _stage2(TINY_CFG, TINY_VQ, xs, B, bf16x3=bf16x3)
    rng = np.random.default_rng(0)
    x, b = _batch(rng, B, xs)
    masks = _masks(rng, TINY_CFG, B)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, idx, lp = PO.pm_vqvae_loss(leaves, vq64, st64, TINY_CFG, TINY_VQ, x, b, True, masks)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    ts.dropout_masks = [f32d(m) for m in masks]
    ts.set_batch(f32d(x), f32d(b))
    with torch.cuda.stream(ts.stream):
        ll = ts.forward(True)
        from posterior_matching_amd import ops
        ops.fill_zero(ts.store.flat_g)
        ts.penc.backward(ts.pcnn.backward(ts.g_ll))
    ts.synchronize()
    assert torch.equal(ts._idx.cpu().long(), idx)
    tol = 1e-5 if not bf16x3 else 1e-4
    _compared()
    assert rel_err(ll, lp) < tol and abs(ts.read_metrics()["loss"] - loss.item()) < tol * abs(loss.item())
    for n, gt in ts.store.to_dict("g").items():
        e = rel_err(gt, grads[n])
        assert e < (1e-4 if not bf16x3 else 1e-2), (n, e)


def test_stage2_train_steps_match_oracle():
    B, xs = 4, (12, 12, 1)
    ts, p64, vq64, st64 = _stage2(TINY_CFG, TINY_VQ, xs, B)
    m = {k: torch.zeros_like(v) for k, v in p64.items()}
    v = {k: torch.zeros_like(t) for k, t in p64.items()}
    p32 = {k: t.float().clone() for k, t in p64.items()}
    m32, v32 = {k: torch.zeros_like(t) for k, t in p32.items()}, {k: torch.zeros_like(t) for k, t in p32.items()}
    vq32 = {k: t.float() for k, t in vq64.items()}
    st32 = {k: (t.float() if t.is_floating_point() else t) for k, t in st64.items()}
    rng = np.random.default_rng(1)
    frozen_before = {n: t.clone() for n, t in ts.vqvae.params_dict().items()}
    for step in range(3):
        x, b = _batch(rng, B, xs)
        masks = _masks(rng, TINY_CFG, B)
        loss, _ = PO.train_step(p64, vq64, st64, m, v, TINY_CFG, TINY_VQ, x, b, step, masks)
        PO.train_step(p32, vq32, st32, m32, v32, TINY_CFG, TINY_VQ, x.float(), b.float(), step, [t.float() for t in masks])
        ts.dropout_masks = [f32d(t) for t in masks]
        ts.set_batch(f32d(x), f32d(b))
        ts.step()
        _compared()
        assert abs(ts.read_metrics()["loss"] - loss.item()) < 1e-4 * abs(loss.item()), step
        pd = ts.store.to_dict("p")
        for n in p64:
            e, e32 = rel_err(pd[n], p64[n]), rel_err(p32[n], p64[n])
            assert e < max(3e-4, 20 * e32) and e < 5e-3, (step, n, e, e32)
    assert ts.step_dev.item() == 3
    for n, t in ts.vqvae.params_dict().items():                     # trainable_predicate: "vqvae/" stays frozen
        assert torch.equal(t, frozen_before[n]), n
    assert int(ts.vqvae.state["counter"]) == 0


@pytest.mark.parametrize("size", ["tiny", "mnist"])
def test_stage2_default_mode_trajectory_within_1e3(size):
    """Default arithmetic (bf16x3 GEMMs in the partial encoder and the PixelCNN, launch-plan style static dropout-mask
    buffers): 4 optimizer steps; loss = -mean log p(codes | x_o) of every step within 1e-3 relative of the float64 oracle
    trajectory - at the tiny sizes and at the configs/pm_vqvae_mnist.py sizes (35.0 M trainable parameters, batch 4).
    The frozen VQ-VAE encoder stays on the strict path so that both sides see the same code indices."""
    if size == "tiny":
        cfg, vq_cfg, xs, B = TINY_CFG, TINY_VQ, (12, 12, 1), 4
    else:
        cfg, vq_cfg, xs, B = pm_vqvae_mnist(), vqvae_mnist()["model"], (28, 28, 1), 4
    ts, p64, vq64, st64 = _stage2(cfg, vq_cfg, xs, B, seed=9, bf16x3=True)
    m = {k: torch.zeros_like(v) for k, v in p64.items()}
    v = {k: torch.zeros_like(t) for k, t in p64.items()}
    rng = np.random.default_rng(4)
    for step in range(4):
        x, b = _batch(rng, B, xs)
        masks = _masks(rng, cfg, B)
        loss, _ = PO.train_step(p64, vq64, st64, m, v, cfg, vq_cfg, x, b, step, masks)
        ts.dropout_masks = [f32d(t) for t in masks]
        ts.set_batch(f32d(x), f32d(b))
        ts.step()
        _compared()
        assert abs(ts.read_metrics()["loss"] - loss.item()) < 1e-3 * abs(loss.item()), (step, ts.read_metrics(), loss.item())
    assert ts.step_dev.item() == 4


def test_stage2_reference_config_small_batch():
    """configs/pm_vqvae_mnist.py network sizes (34.2 M PixelCNN params) at batch 2: loss and a sample of gradients."""
    cfg, vq_cfg = pm_vqvae_mnist(), vqvae_mnist()["model"]
    B, xs = 2, (28, 28, 1)
    ts, p64, vq64, st64 = _stage2(cfg, vq_cfg, xs, B, seed=6)
    assert ts.num_trainable_params == 35026768                       # 841 936 + 34 184 832
    rng = np.random.default_rng(2)
    x, b = _batch(rng, B, xs)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, idx, lp = PO.pm_vqvae_loss(leaves, vq64, st64, cfg, vq_cfg, x, b, False)
    names = ["pixel_cnn/embed/embeddings", "pixel_cnn/down_3/horizontal/conv2/w", "pixel_cnn/up_7/horizontal/linear/w",
             "pixel_cnn/up_0/vertical/cond/w", "partial_encoder/linear/w", "partial_encoder/encoder/enc_1/w",
             "pixel_cnn/out_conv/b"]
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names])))
    ts.set_batch(f32d(x), f32d(b))
    with torch.cuda.stream(ts.stream):
        ll = ts.forward(False)
        from posterior_matching_amd import ops
        ops.neg_mean_loss(ll, 1.0 / B, ts.metrics, ts.g_ll)
        ops.fill_zero(ts.store.flat_g)
        ts.penc.backward(ts.pcnn.backward(ts.g_ll))
    ts.synchronize()
    assert torch.equal(ts._idx.cpu().long(), idx)
    assert rel_err(ll, lp) < 2e-5
    gd = ts.store.to_dict("g")
    for n in names:
        assert rel_err(gd[n], grads[n]) < 2e-4, (n, rel_err(gd[n], grads[n]))


def _stage2_fwd_bwd_as_the_train_step(ts, x, b, masks):
    ts.dropout_masks = [f32d(m) for m in masks]
    ts.set_batch(f32d(x), f32d(b))
    with torch.cuda.stream(ts.stream):
        ts._forward_backward()          # grouped weight gradients (ops.WgradBatch), two chains: what ts.step() runs
    ts.synchronize()
    return ts.read_metrics()["loss"], {n: t.clone() for n, t in ts.store.to_dict("g").items()}


STAGE2_SAMPLED = ["pixel_cnn/embed/embeddings", "pixel_cnn/down_3/horizontal/conv2/w", "pixel_cnn/up_7/horizontal/linear/w",
                  "pixel_cnn/up_0/vertical/cond/w", "pixel_cnn/down_0/vertical/conv1/w", "pixel_cnn/up_5/vertical/conv2/w",
                  "pixel_cnn/down_6/horizontal/conv1/w", "pixel_cnn/vertical_init/w", "pixel_cnn/horizontal_up/w", "pixel_cnn/horizontal_left/w",
                  "partial_encoder/linear/w", "partial_encoder/encoder/enc_1/w", "pixel_cnn/out_conv/w", "pixel_cnn/out_conv/b",
                  "pixel_cnn/down_2/horizontal/conv2/b"]


def test_stage2_reference_config_at_batch_128_matches_oracle():
    """configs/pm_vqvae_mnist.py (35.0 M trainable) in TRAINING mode with explicit dropout masks at B = 128 - the batch from
    which the masked sub-kernels run on image_conv_bf16 and the conditional row sums on rows_sum_v4, i.e. the kernels of
    the B = 256 benchmark - through the train step's own forward / backward (grouped weight gradients, two chains), default
    arithmetic: code indices exact, loss 1e-4, sampled gradient tensors 1e-2 against the float64 oracle (the bars of
    test_stage2_loss_and_grads[bf16x3])."""
    cfg, vq_cfg = pm_vqvae_mnist(), vqvae_mnist()["model"]
    B, xs = 128, (28, 28, 1)
    ts, p64, vq64, st64 = _stage2(cfg, vq_cfg, xs, B, seed=6, bf16x3=True)
    rng = np.random.default_rng(12)
    x, b = _batch(rng, B, xs)
    masks = _masks(rng, cfg, B)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, idx, lp = PO.pm_vqvae_loss(leaves, vq64, st64, cfg, vq_cfg, x, b, True, masks)
    grads = dict(zip(STAGE2_SAMPLED, torch.autograd.grad(loss, [leaves[n] for n in STAGE2_SAMPLED])))
    got_loss, gd = _stage2_fwd_bwd_as_the_train_step(ts, x, b, masks)
    assert torch.equal(ts._idx.cpu().long(), idx)
    assert abs(got_loss - loss.item()) < 1e-4 * abs(loss.item()), (got_loss, loss.item())
    for n in STAGE2_SAMPLED:
        e = rel_err(gd[n], grads[n])
        assert e < 1e-2, (n, e)


def test_stage2_reference_config_at_batch_256_default_vs_strict():
    """B = 256 (BASELINE's batch for pm_vqvae_mnist; the float64 CPU pass would need ~30 GB): the default arithmetic on
    the train step's path against the strict-f32 path (f32 MFMA kernels, per-layer weight gradients - a disjoint set of
    kernels, itself checked against the oracle at B <= 128): loss 1e-4, EVERY gradient tensor 1e-2."""
    cfg, vq_cfg = pm_vqvae_mnist(), vqvae_mnist()["model"]
    B, xs = 256, (28, 28, 1)
    ts, p64, vq64, st64 = _stage2(cfg, vq_cfg, xs, B, seed=6, bf16x3=True)
    rng = np.random.default_rng(13)
    x, b = _batch(rng, B, xs)
    masks = _masks(rng, cfg, B)
    loss_fast, g_fast = _stage2_fwd_bwd_as_the_train_step(ts, x, b, masks)
    ts.store.use_bf16 = False
    loss_strict, g_strict = _stage2_fwd_bwd_as_the_train_step(ts, x, b, masks)
    assert abs(loss_fast - loss_strict) < 1e-4 * abs(loss_strict), (loss_fast, loss_strict)
    worst = max((rel_err(g_fast[n], g_strict[n]), n) for n in g_fast)
    assert worst[0] < 1e-2, worst


def test_sampling_imputation_and_psnr_match_oracle():
    """PixelCNN ancestral sampling with explicit Gumbel noise, vqvae_impute and the held-out PSNR
    (reference pixel_cnn.py:102-124, vqvae.py:269-312, eval_pm_vqvae.py:133-136)."""
    from posterior_matching_amd.models.vqvae import imputation_psnr, vqvae_impute

    B, S, xs = 3, 2, (12, 12, 1)
    ts, p64, vq64, st64 = _stage2(TINY_CFG, TINY_VQ, xs, B)
    rng = np.random.default_rng(5)
    x, b = _batch(rng, B, xs)
    K, P = TINY_VQ["num_embeddings"], 9
    u = rng.uniform(1e-6, 1 - 1e-6, size=(P, B * S, K))
    gumbel = torch.tensor(-np.log(-np.log(u)))
    want = PO.vqvae_impute(p64, vq64, st64, TINY_CFG, TINY_VQ, x, b, S, gumbel)
    want_psnr = PO.imputation_psnr(want, x)
    got = vqvae_impute(ts.vqvae, ts.penc, ts.pcnn, f32d(x), f32d(b), num_samples=S, gumbel=f32d(gumbel))
    psnr = imputation_psnr(got, f32d(x))
    torch.cuda.synchronize()
    assert got.shape == (B, S) + xs
    assert rel_err(got, want) < 1e-5                 # identical code draws (no Gumbel near-tie at this seed)
    assert rel_err(psnr, want_psnr) < 1e-5
    obs = b.bool().expand(B, *xs)
    assert torch.equal(got.cpu()[:, 0][obs], x.float()[obs]) and got.min() >= 0 and got.max() <= 1
    # device noise path: reproducible for a seed, different across seeds, indices in range
    cond = torch.randn((B, TINY_CFG["conditional_dim"]), device=dev())
    s1 = ts.pcnn.sample(seed=7, sample_shape=S, conditional_input=cond)
    s2 = ts.pcnn.sample(seed=7, sample_shape=S, conditional_input=cond)
    s3 = ts.pcnn.sample(seed=8, sample_shape=S, conditional_input=cond)
    assert s1.shape == (S, B, 3, 3) and torch.equal(s1, s2) and not torch.equal(s1, s3)
    assert int(s1.min()) >= 0 and int(s1.max()) < K
    assert ts.pcnn.sample(seed=1, conditional_input=cond).shape == (B, 3, 3)
    # PSNR known answers (SURVEY.md 8c xii): uniform error e -> -20 log10 e ; identical images -> +inf
    xi = torch.rand((2, 12, 12, 1), device=dev())
    imp = (xi + 0.1).view(2, 1, 12, 12, 1).repeat(1, 3, 1, 1, 1).contiguous()
    assert torch.allclose(imputation_psnr(imp, xi).cpu(), torch.full((2,), 20.0), atol=1e-3)
    assert torch.isinf(imputation_psnr(xi.view(2, 1, 12, 12, 1).contiguous(), xi)).all()


def test_two_stage_scripts_end_to_end(tmp_path):
    """train_vqvae.py -> train_pm_vqvae.py -> eval_pm_vqvae.py on small network sizes: the stage-1 run
    directory feeds stage 2, stage 2 checkpoints both trees, the PSNR script reads them back."""
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

    run("train_vqvae.py", "--config", os.path.join(root, "configs", "vqvae_mnist.py"), "--config.steps=20",
        "--config.validation_freq=20", "--config.seed=1")
    stage1 = os.path.join(tmp_path, "runs", [d for d in os.listdir(os.path.join(tmp_path, "runs")) if d.startswith("vqvae-")][0])
    run("train_pm_vqvae.py", "--config", os.path.join(root, "configs", "pm_vqvae_mnist.py"), f"--config.vqvae_dir={stage1}",
        "--config.steps=6", "--config.validation_freq=3", "--config.seed=2", "--config.pixel_cnn.num_resnet=1",
        "--config.pixel_cnn.num_filters=32", "--config.conditional_dim=64", "--config.data.train_batch_size=8",
        "--config.data.val_batch_size=8")
    stage2 = os.path.join(tmp_path, "runs", [d for d in os.listdir(os.path.join(tmp_path, "runs")) if d.startswith("pm-vqvae-")][0])
    lines = [json.loads(l) for l in open(os.path.join(stage2, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [3, 6] and all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) for l in lines)
    imp = np.load(os.path.join(stage2, "tb", "imputations_6.npy"))
    assert imp.shape == (3, 28, 28 * 7, 1) and imp.min() >= 0.0 and imp.max() <= 1.0
    sys.path.insert(0, root)
    s1 = pickle.load(open(os.path.join(stage1, "train_state.pkl"), "rb"))
    s2 = pickle.load(open(os.path.join(stage2, "train_state.pkl"), "rb"))
    assert s2.step == 6 and int(s2.state["vqvae/counter"]) == 20
    for k, v in s1.params.items():                                  # stage 2 never touches the VQ-VAE
        assert torch.equal(s2.params["vqvae/" + k], v), k
    assert any(k.startswith("pixel_cnn/") for k in s2.params) and any(k.startswith("partial_encoder/") for k in s2.params)
    out = run("eval_pm_vqvae.py", "--run_dir", stage2, "--num_instances", "16", "--batch_size", "8", "--num_samples", "2")
    res = json.loads(out.strip().splitlines()[-1])
    assert res["num_instances"] == 16 and np.isfinite(res["mean_psnr"]) and 0.0 < res["mean_psnr"] < 60.0
