"""Run-to-run repeatability of the gradients at the reference network sizes, with every companion stream active.

Round 4: every weight / bias gradient leaves its kernel as per-split PARTIAL SUMS (plain stores) that pm_reduce_partials adds in
a fixed order, and no split-K data gradient runs in the PM-VAE step - its gradients are asserted BIT-IDENTICAL between runs
(what jax.grad gives the reference, train_pm_vae.py:58-72).  The VDVAE's too (its last atomics - the gain / bias sums of the
final affine and the global gradient norm - became partial sums / a fixed-order sum), step and parameter EMA included.  The
PixelCNN's as well: its embedding scatter runs in order-independent 64-bit fixed point (pm_embed_bwd_exact) and the split-K data
gradients of short grids keep their K slices in slabs added in slice order (pm_gather_gemm_bf16_sk).  The last three sources -
VectorQuantizerEMA's dw (pm_vq_dw_exact), the Normal decoder's d log_scale (pm_normal_ll_bwd_det) and split-K at small batches -
went the same way, so a train step of EVERY BASELINE.json config is asserted bit-reproducible below.  (TOL is what _compare
falls back to for a tensor a test does not declare exact; the hazard this file was first written for - the packed-FP32
instability of the thin weight-gradient kernel, DESIGN.md section 6 - produced 1e-4 ... 1e-2.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _compare(runs, exact=lambda name: False):
    ref = runs[0]
    inexact = []
    for other in runs[1:]:
        for n in ref:
            if torch.equal(ref[n], other[n]):
                continue
            scale = max(ref[n].abs().max().item(), 1e-30)
            err = (ref[n] - other[n]).abs().max().item() / scale
            assert not exact(n), (n, err, "expected bit-identical gradients")
            assert err < TOL, (n, err)
            inexact.append(n)
    return sorted(set(inexact))


def test_pm_vae_gradients_repeat():
    from tests.test_gpu_parity import _inputs, _product_model

    cfg, xs, x, b, eps = _inputs("mnist", 256, 9)
    m = _product_model(cfg, xs)
    xd, bd, ed = x.float().cuda(), b.float().cuda(), eps.float().cuda()
    m(xd, bd, True, eps=ed)
    g = [torch.full((256,), v, device="cuda") for v in (-1 / 256, 1 / 256, -1 / 256)]
    runs = []
    for _ in range(4):
        m.zero_grad()
        m.backward(*g)
        torch.cuda.synchronize()
        runs.append({n: t.clone() for n, t in m.grads_dict().items()})
    assert _compare(runs, exact=lambda name: True) == []


def test_pm_vae_train_step_is_bit_reproducible():
    """two PMVAETrainStep objects from the same seed, launch-plan replay on two streams: parameters equal bit for bit after 6
    optimizer steps (bf16x3 default arithmetic)"""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVAETrainStep
    from tests.test_gpu_parity import _inputs, _product_model

    cfg, xs, x, b, eps = _inputs("mnist", 256, 9)
    finals = []
    for _ in range(2):
        m = _product_model(cfg, xs)
        ts = PMVAETrainStep(m, cfg, optim.adam(1e-3), 256, xs, seed=3)
        ts.set_batch(x.float().cuda(), b.float().cuda())
        for _ in range(6):
            ts.step()
        ts.synchronize()
        finals.append(m.store.flat_p.clone())
    assert torch.equal(finals[0], finals[1])


def test_vdvae_gradients_repeat():
    from tests.ref_configs import pm_vdvae_mnist
    from tests.test_gpu_vdvae import _setup, f32d

    m, _, x, b, eps = _setup(pm_vdvae_mnist(), 8, seed=8, bf16x3=True)
    runs = []
    for _ in range(3):
        m(f32d(x), f32d(b), [f32d(e) for e in eps])
        m.zero_grad()
        m.backward()
        torch.cuda.synchronize()
        runs.append({n: t.clone() for n, t in m.grads_dict().items()})
    assert _compare(runs, exact=lambda name: True) == []          # round 4: affine_bwd's gain / bias sums were the last atomics


def test_vdvae_train_step_is_bit_reproducible():
    """two VDVAETrainStep objects from the same seed (fused Blocks, grouped weight gradients on side streams, global-norm
    clip through pm_sumsq_det, parameter EMA): parameters AND EMA equal bit for bit after 4 optimizer steps"""
    from posterior_matching_amd.engine import VDVAETrainStep
    from tests.ref_configs import pm_vdvae_mnist
    from tests.test_gpu_vdvae import _setup, f32d

    finals = []
    for _ in range(2):
        cfg = pm_vdvae_mnist()
        m, _, x, b, eps = _setup(cfg, 8, seed=8, bf16x3=True)
        ts = VDVAETrainStep(m, cfg["lr"], 8, gradient_clip=cfg["gradient_clip"], ema_rate=cfg["ema_rate"], seed=5,
                            external_eps=True)
        ts.set_batch(f32d(x), f32d(b), [f32d(e) for e in eps])
        for _ in range(4):
            ts.step()
        ts.synchronize()
        finals.append((m.store.flat_p.clone(), ts.ema.clone(), ts.read_metrics()["loss"]))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])
    # (the reported loss sums per-example KL / log-likelihood values with atomics: a metric, equal to ~1e-7 relative only)
    assert abs(finals[0][2] - finals[1][2]) <= 1e-5 * abs(finals[0][2])


@pytest.mark.parametrize("B", [32, 256])
def test_pm_vqvae_gradients_repeat(B):
    from posterior_matching_amd import ops
    from tests.ref_configs import pm_vqvae_mnist, vqvae_mnist
    from tests.test_gpu_pixelcnn import _batch, _stage2, f32d

    cfg, vq_cfg = pm_vqvae_mnist(), vqvae_mnist()["model"]
    xs = (28, 28, 1)
    ts, _, _, _ = _stage2(cfg, vq_cfg, xs, B, seed=6, bf16x3=True)
    x, b = _batch(np.random.default_rng(2), B, xs)
    ts.set_batch(f32d(x), f32d(b))
    runs = []
    for _ in range(3):
        with torch.cuda.stream(ts.stream):
            ll = ts.forward(False)
            ops.neg_mean_loss(ll, 1.0 / B, ts.metrics, ts.g_ll)
            ops.fill_zero(ts.store.flat_g)
            ts.penc.backward(ts.pcnn.backward(ts.g_ll))
        ts.synchronize()
        runs.append({n: t.clone() for n, t in ts.store.to_dict("g").items()})
    inexact = _compare(runs, exact=lambda name: True)
    # every gradient bit-identical at both batches - weight gradients as partial sums, the embedding scatter in
    # order-independent fixed point, and (B = 32) the short grids' split-K data gradients as slabs added in slice order
    assert not inexact


def test_pm_vqvae_train_step_is_bit_reproducible():
    """two PMVQVAETrainStep objects (35.0 M trainable parameters, in-place Philox dropout, two chains, grouped weight gradients)
    from the same seeds at the benchmarked batch: parameters equal bit for bit after 3 optimizer steps"""
    from tools.workloads import build

    finals = []
    for _ in range(2):
        w = build("pm_vqvae_mnist", 256)
        w.feed()
        for _ in range(3):
            w.step()
        w.synchronize()
        finals.append(w.ts.store.flat_p.clone())
        del w
        torch.cuda.empty_cache()
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize("name,B", [("pm_vqvae_celeb_a", 16), ("vqvae_mnist", 256), ("pm_vae_gas", 128)])
def test_remaining_baseline_configs_train_steps_are_bit_reproducible(name, B):
    """the BASELINE.json configurations the tests above do not cover, at their bench batch sizes (tools/workloads.py): two
    objects from the same seeds, 3 optimizer steps, parameters (and the VQ-VAE's EMA codebook state) equal bit for bit"""
    from tools.workloads import build

    finals = []
    for _ in range(2):
        w = build(name, B)
        w.feed()
        for _ in range(3):
            w.step()
        w.synchronize()
        st = w.ts.store if hasattr(w.ts, "store") else w.ts.model.store
        state = [st.flat_p.clone()]
        vq = getattr(getattr(w.ts, "model", None), "state", None)          # VQVAE: the EMA codebook and its statistics
        if isinstance(vq, dict):
            state += [t.clone() for _, t in vq.items() if torch.is_tensor(t)]
        if name == "vqvae_mnist":
            assert len(state) > 1
        finals.append(state)
        del w
        torch.cuda.empty_cache()
    assert len(finals[0]) == len(finals[1])
    for a, b in zip(finals[0], finals[1]):
        assert torch.equal(a, b), (name, (a.float() - b.float()).abs().max().item())
