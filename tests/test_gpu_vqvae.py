"""GPU parity of the VQ-VAE (stage 1) path: HIP kernels through the C ABI vs oracle/vqvae_oracle.py.

Code indices are integer work: they must match the oracle exactly, except where the oracle's own
best / second-best distance gap is below float32 resolution (a genuine tie for f32 arithmetic,
which the reference computes in too).  Floating-point outputs: tolerances stated per test.
"""
import math

import numpy as np
import pytest

from tests import conftest as _conftest
import torch

from oracle import pm_vae_oracle as O
from oracle import vqvae_oracle as VO
from tests.ref_configs import vqvae_mnist

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


def _indices_match(idx_gpu, flat64, emb64, tol=1e-5):
    """exact match, or the oracle's gap between chosen and GPU-chosen code is a float32 tie"""
    dist = VO.vq_distances(flat64, emb64)
    want = torch.argmax(-dist, 1)
    got = idx_gpu.reshape(-1).long().cpu()
    bad = (want != got).nonzero().reshape(-1)
    for r in bad.tolist():
        gap = (dist[r, got[r]] - dist[r, want[r]]).item()
        assert 0 <= gap < tol * max(1.0, dist[r, want[r]].abs().item()), (r, gap)
    return len(bad)


# ----------------------------------------------------------------------------------------------
# kernels in isolation
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,D,K", [(98, 64, 256), (1000, 64, 512), (37, 16, 24), (4096, 64, 256)])
def test_vq_select_and_ema(N, D, K):
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import LayerGeom

    gen = torch.Generator().manual_seed(N + K)
    z = torch.randn((N, D), generator=gen, dtype=F64) * 0.3
    emb = (torch.rand((D, K), generator=gen, dtype=F64) * 2 - 1) * math.sqrt(3.0 / D)
    state = {"vq/embeddings": emb}
    for name, shp in (("vq/ema_cluster_size", (K,)), ("vq/ema_dw", (D, K))):
        state[f"{name}/hidden"] = torch.rand(shp, generator=gen, dtype=F64)
        state[f"{name}/average"] = torch.zeros(shp, dtype=F64)
        state[f"{name}/counter"] = torch.tensor(6)
    cc, decay = 0.25, 0.99
    zr = z.clone().requires_grad_(True)
    out, new_state = VO.vector_quantizer_ema(state, zr, cc, decay, True)
    out["loss"].backward()

    d = dev()
    zd, ed = z.float().to(d), emb.float().to(d)
    dots = torch.empty((N, K), device=d)
    ops.gather_gemm(LayerGeom.dense(D, K)._desc(N, "fwd"), zd, ed, None, None, None, dots)
    e2, idx = torch.empty(K, device=d), torch.empty(N, dtype=torch.int32, device=d)
    quant, cgrad = torch.empty((N, D), device=d), torch.empty((N, D), device=d)
    sqerr, counts, dw = torch.empty(N, device=d), torch.empty(K, device=d), torch.empty((D, K), device=d)
    ops.vq_select(zd, ed, dots, e2, idx, quant, cgrad, sqerr, counts, dw, 2.0 * cc / (N * D))
    torch.cuda.synchronize()
    nbad = _indices_match(idx, z, emb)
    assert nbad <= 1
    if nbad == 0:
        assert rel_err(quant, out["quantize"]) < 1e-6
        assert torch.equal(counts.cpu().double(), out["encodings"].sum(0))
        assert rel_err(dw, z.t() @ out["encodings"]) < 2e-6
        assert rel_err(cgrad, zr.grad) < 2e-6                     # commitment gradient
        assert abs(sqerr.sum().item() / (N * D) * cc - out["loss"].item()) < 2e-6 * out["loss"].item()
        # rows of quantize are codebook columns (KAT x)
        assert torch.equal(quant.cpu(), ed.cpu().t()[idx.long().cpu()])
        csh, dwh = state["vq/ema_cluster_size/hidden"].float().to(d), state["vq/ema_dw/hidden"].float().to(d)
        csa, dwa = torch.empty(K, device=d), torch.empty((D, K), device=d)
        counter = torch.tensor([6], dtype=torch.int32, device=d)
        ops.vq_ema_update(counts, dw, csh, csa, dwh, dwa, ed, counter, decay, VO.VQ_EPSILON)
        torch.cuda.synchronize()
        assert counter.item() == 7
        # (1 - decay) in float32 carries 1e-6 relative error (the reference computes it in f32 too)
        assert rel_err(csh, new_state["vq/ema_cluster_size/hidden"]) < 5e-6
        assert rel_err(dwa, new_state["vq/ema_dw/average"]) < 5e-6
        assert rel_err(ed, new_state["vq/embeddings"]) < 5e-6
        lk = torch.empty((N, D), device=d)
        ops.vq_lookup(idx, ed, lk)
        assert torch.equal(lk.cpu(), ed.cpu().t()[idx.long().cpu()])


@_conftest.compares
def test_vq_tie_breaks_to_lowest_index():
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import LayerGeom

    d = dev()
    N, D, K = 8, 64, 256
    emb = torch.randn((D, K), device=d)
    emb[:, 200] = emb[:, 3]                  # duplicate code: both are equally near
    emb[:, 77] = emb[:, 3]
    z = emb[:, 3].reshape(1, D).repeat(N, 1).contiguous()
    dots = torch.empty((N, K), device=d)
    ops.gather_gemm(LayerGeom.dense(D, K)._desc(N, "fwd"), z, emb, None, None, None, dots)
    e2, idx = torch.empty(K, device=d), torch.empty(N, dtype=torch.int32, device=d)
    quant, sqerr, counts = torch.empty((N, D), device=d), torch.empty(N, device=d), torch.empty(K, device=d)
    ops.vq_select(z, emb, dots, e2, idx, quant, None, sqerr, counts, None, 0.0)
    assert idx.cpu().tolist() == [3] * N and counts[3].item() == N and counts.sum().item() == N


def test_normal_ll_with_scale_eps():
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(2)
    B, D = 7, 784
    loc, x, g = torch.randn((B, D), generator=gen, dtype=F64), torch.rand((B, D), generator=gen, dtype=F64), \
        torch.randn((B,), generator=gen, dtype=F64)
    ls = torch.tensor(-0.7, dtype=F64)
    lr_, lsr = loc.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    ll = O.normal_log_prob(x, lr_, torch.exp(lsr) + 1e-5).sum(-1)
    (ll * g).sum().backward()
    d = dev()
    lld, lsd = torch.empty(B, device=d), ls.float().to(d)
    ops.normal_ll_fwd(loc.float().to(d), x.float().to(d), lsd, lld, 1e-5)
    assert rel_err(lld, ll) < 2e-6
    dloc, dls = torch.empty((B, D), device=d), torch.zeros((), device=d)
    ops.normal_ll_bwd(loc.float().to(d), x.float().to(d), lsd, g.float().to(d), dloc, dls, 1e-5)
    assert rel_err(dloc, lr_.grad) < 2e-6 and rel_err(dls, lsr.grad) < 5e-6


def test_epilogue_aux_after_res():
    """PM_AUX_AFTER_RES: out = (acc + res) * relu'(aux)"""
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import ACT_RELU, AUX_AFTER_RES, LayerGeom

    gen = torch.Generator().manual_seed(4)
    for geom, B in ((LayerGeom.conv(7, 7, 32, 32, 3, 1, "SAME"), 5), (LayerGeom.dense(48, 40), 70)):
        x = torch.randn((B, geom.IH, geom.IW, geom.CI), generator=gen, dtype=F64)
        w = torch.randn(geom.weight_shape, generator=gen, dtype=F64) * 0.1
        dy = torch.randn((B, geom.OH, geom.OW, geom.CO), generator=gen, dtype=F64)
        xr = x.clone().requires_grad_(True)
        y = O.conv2d(xr, w, None, 1, "SAME") if geom.kind == "conv" else (xr.reshape(B, -1) @ w).reshape(dy.shape)
        (y * dy).sum().backward()
        aux, res = torch.randn(x.shape, generator=gen, dtype=F64), torch.randn(x.shape, generator=gen, dtype=F64)
        want = (xr.grad + res) * (aux > 0).double()
        d = dev()
        out = torch.empty(x.shape, device=d)
        ops.layer_dgrad(geom, dy.float().to(d), w.float().to(d), out, aux=aux.float().to(d),
                        aux_act=ACT_RELU | AUX_AFTER_RES, res=res.float().to(d))
        assert rel_err(out, want) < 2e-6


# ----------------------------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------------------------
def _setup(B, seed=7, bf16x3=False, cfg=None, perturb=True):
    from posterior_matching_amd.models.vqvae import VQVAE

    cfg = cfg or vqvae_mnist()
    rng = np.random.default_rng(seed)
    x = torch.tensor(rng.uniform(size=(B, 28, 28, 1)) * (rng.uniform(size=(B, 28, 28, 1)) < 0.3))
    m = VQVAE(**cfg["model"], device="cuda:0", seed=seed)
    m.init((28, 28, 1))
    m.store.use_bf16 = bf16x3
    if perturb:
        gen = torch.Generator().manual_seed(seed)
        m.load_params({n: t.cpu() + 0.05 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    st64 = _oracle_state(m.state_dict())
    return cfg, x, m, p64, st64


def _oracle_state(sd):
    st = {"vq/embeddings": sd["embeddings"].cpu().double()}
    for name in ("ema_cluster_size", "ema_dw"):
        st[f"vq/{name}/hidden"] = sd[f"{name}/hidden"].cpu().double()
        st[f"vq/{name}/average"] = sd[f"{name}/average"].cpu().double()
        st[f"vq/{name}/counter"] = torch.tensor(int(sd["counter"].item()))
    return st


def test_param_names_and_count_match_oracle():
    cfg, x, m, p64, st64 = _setup(2, perturb=False)
    shapes = VO.param_shapes(cfg["model"], 1)
    assert {n: tuple(t.shape) for n, t in p64.items()} == shapes
    assert m.num_params == 88002                                   # 40 464 + 47 538
    lim = math.sqrt(3.0 / 64)
    assert st64["vq/embeddings"].abs().max() <= lim and st64["vq/embeddings"].std() > 0.4 * lim


@pytest.mark.parametrize("B,bf16x3,seed", [(6, False, 7), (6, True, 7), (32, False, 8)])
def test_vqvae_forward_and_grads(B, bf16x3, seed):
    """Seeds are chosen free of relu-kink ties: with seed 7 at B=32 one encoder pre-activation is
    +1.5e-8 in float32 and <= 0 in float64, which switches that element's relu' (a tie for float32
    arithmetic - tools/vq_grad_errors.py shows it - not a kernel difference)."""
    cfg, x, m, p64, st64 = _setup(B, seed=seed, bf16x3=bf16x3)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out, new_state = VO.vqvae_loss(leaves, st64, cfg, x, True)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))

    got = m(x.float().to(dev()), is_training=True)
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    tol = 2e-5 if not bf16x3 else 2e-4
    D = cfg["model"]["embedding_dim"]
    assert rel_err(got["z"], out["z"]) < tol
    nbad = _indices_match(got["vq_output"]["encoding_indices"], out["z"].detach().reshape(-1, D), st64["vq/embeddings"],
                          tol=1e-5 if not bf16x3 else 1e-4)
    assert nbad == 0 or bf16x3
    if nbad:
        pytest.skip("bf16x3 forward flipped a near-tie code; remaining comparisons need identical codes")
    assert torch.equal(got["vq_output"]["encoding_indices"].cpu().long(), out["vq_output"]["encoding_indices"])
    assert torch.equal(got["vq_output"]["encodings"].cpu().double(), out["vq_output"]["encodings"])
    assert rel_err(got["vq_output"]["quantize"], out["vq_output"]["quantize"]) < 1e-6
    assert rel_err(got["reconstruction"], out["reconstruction"]) < tol
    assert abs(got["loss"].item() - loss.item()) < tol * abs(loss.item())
    assert abs(got["reconstruction_loss"].item() - aux["reconstruction_loss"].item()) < tol * abs(loss.item())
    assert abs(got["vq_output"]["loss"].item() - aux["vq_loss"].item()) < 10 * tol * aux["vq_loss"].item()
    assert abs(got["vq_output"]["perplexity"].item() - aux["perplexity"].item()) < 1e-4 * aux["perplexity"].item()
    assert rel_err(got["scale"], torch.exp(p64["decoder/log_scale"]) + 1e-5) < 1e-6
    gd = m.grads_dict()
    l32 = {n: t.float().clone().requires_grad_(True) for n, t in p64.items()}
    st32 = {k: (v.float() if v.is_floating_point() else v) for k, v in st64.items()}
    loss32, _, _, _ = VO.vqvae_loss(l32, st32, cfg, x.float(), True)
    g32 = dict(zip(l32, torch.autograd.grad(loss32, list(l32.values()))))
    for n in grads:
        e, e32 = rel_err(gd[n], grads[n]), rel_err(g32[n], grads[n])
        if not bf16x3:
            assert e < max(5e-5, 10 * e32) and e < max(2e-4, 3 * e32), (n, e, e32)
        else:
            assert e < 5e-3, (n, e, e32)
    # haiku state after the forward (EMA update) - f32 (1 - decay) carries 1e-6
    sd = _oracle_state(m.state_dict())
    for k, v in new_state.items():
        if k.endswith("counter"):
            assert int(sd[k]) == int(v)
        else:
            assert rel_err(sd[k], v) < (2e-5 if not bf16x3 else 2e-4), k


@_conftest.compares
def test_vqvae_eval_mode_leaves_state_untouched():
    cfg, x, m, p64, st64 = _setup(5)
    before = m.state_dict()
    got = m(x.float().to(dev()), is_training=False)
    loss, aux, out, _ = VO.vqvae_loss(p64, st64, cfg, x, False)
    torch.cuda.synchronize()
    assert abs(got["loss"].item() - loss.item()) < 2e-5 * abs(loss.item())
    after = m.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_vqvae_train_steps_match_oracle(use_graph):
    """4 optimizer steps of train_vqvae.py's loop (f32 mode): parameters, Adam moments and the haiku
    state (codebook + EMAs) track the float64 oracle."""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import VQVAETrainStep

    B = 16
    cfg, x, m, p64, st64 = _setup(B, seed=9)
    ts = VQVAETrainStep(m, optim.adam(cfg["learning_rate"]), B, (28, 28, 1), use_graph=use_graph)
    mo = {k: torch.zeros_like(v) for k, v in p64.items()}
    vo = {k: torch.zeros_like(v) for k, v in p64.items()}
    p32 = {k: v.float().clone() for k, v in p64.items()}
    st32 = {k: (v.float() if v.is_floating_point() else v) for k, v in st64.items()}
    m32, v32 = {k: torch.zeros_like(v) for k, v in p32.items()}, {k: torch.zeros_like(v) for k, v in p32.items()}
    rng = np.random.default_rng(3)
    for step in range(4):
        xb = torch.tensor(rng.uniform(size=(B, 28, 28, 1)) * (rng.uniform(size=(B, 28, 28, 1)) < 0.3))
        loss, aux, g, st64 = VO.train_step(p64, st64, mo, vo, cfg, xb, step)
        _, _, _, st32 = VO.train_step(p32, st32, m32, v32, cfg, xb.float(), step)
        ts.set_batch(xb.float().to(dev()))
        ts.step()
        met = ts.read_metrics()
        _compared()
        if not abs(met["loss"] - loss.item()) < 1e-4 * abs(loss.item()):     # diagnostics: which buffer went wrong
            bad = [(k[0], t.float().abs().max().item()) for k, t in m.ws._bufs.items()
                   if not (t.float().abs().max().item() < 1e6)]
            detail = []
            for k, t in m.ws._bufs.items():
                if k[0] == "decoder/dec_1_out":
                    nz = (t.float().abs() > 1e6).nonzero()
                    detail = [len(nz), nz[:6].tolist(), nz[-3:].tolist(), t[tuple(nz[0].tolist())].item()]
            raise AssertionError((step, met, loss.item(), bad, detail))
        assert abs(met["perplexity"] - aux["perplexity"].item()) < 1e-3 * aux["perplexity"].item()
        pd = m.params_dict()
        # Adam's first updates are sign-like (|u| ~ lr whatever the gradient scale): yardstick is
        # the same restatement stepped in float32 on the CPU
        for n in p64:
            e, e32 = rel_err(pd[n], p64[n]), rel_err(p32[n], p64[n])
            assert e < max(1e-4, 20 * e32) and e < 5e-3, (step, n, e, e32)
        sd = _oracle_state(m.state_dict())
        for k, v in st64.items():
            if k.endswith("counter"):
                assert int(sd[k]) == int(v) == step + 1
            else:
                assert rel_err(sd[k], v) < max(1e-4, 20 * rel_err(st32[k], v)), (step, k)
    assert ts.step_dev.item() == 4


@_conftest.compares
def test_vqvae_full_batch_properties():
    """B = 256 (BASELINE config size): size-independent properties."""
    cfg, x, m, p64, st64 = _setup(256, seed=21)
    d = dev()
    xd = x.float().to(d)
    got = m(xd, is_training=False)
    torch.cuda.synchronize()
    idx = got["vq_output"]["encoding_indices"].clone()
    q = got["vq_output"]["quantize"].clone()
    emb = m.state["embeddings"]
    # quantised rows are codebook rows; lookup(idx) reproduces them; idempotence: quantising the
    # quantised tensor returns the same codes
    assert torch.equal(q.reshape(-1, 64), emb.t()[idx.reshape(-1).long()])
    assert torch.equal(m.vq.quantize(idx), q)
    again = m.vq(q.clone(), is_training=False)
    assert torch.equal(again["encoding_indices"].reshape(-1), idx.reshape(-1))
    assert again["sqerr"].abs().max().item() < 1e-9
    perp = got["vq_output"]["perplexity"].item()
    assert 1.0 <= perp <= 256.0
    # batch-permutation invariance of the loss
    loss = got["loss"].item()
    perm = torch.randperm(256, device=d)
    loss_p = m(xd[perm].contiguous(), is_training=False)["loss"].item()
    assert abs(loss - loss_p) < 1e-5 * abs(loss)


def test_train_vqvae_script_end_to_end(tmp_path):
    """train_vqvae.py --config configs/vqvae_mnist.py for a few steps: runs, logs, checkpoints state."""
    import json
    import os
    import pickle
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "train_vqvae.py"), "--config",
                          os.path.join(root, "configs", "vqvae_mnist.py"), "--config.steps=30",
                          "--config.validation_freq=15", "--config.seed=3"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    run = os.path.join(tmp_path, "runs", os.listdir(os.path.join(tmp_path, "runs"))[0])
    assert json.load(open(os.path.join(run, "model_config.json")))["num_embeddings"] == 256
    lines = [json.loads(l) for l in open(os.path.join(run, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [15, 30]
    assert all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) for l in lines)
    assert lines[1]["train_loss"] < lines[0]["train_loss"]
    rec = np.load(os.path.join(run, "tb", "reconstructions_30.npy"))
    assert rec.shape == (3, 28, 56, 1) and rec.min() >= 0.0 and rec.max() <= 1.0
    sys.path.insert(0, root)
    state = pickle.load(open(os.path.join(run, "train_state.pkl"), "rb"))
    assert state.step == 30 and len(state.params) == 31 and int(state.state["counter"]) == 30
    assert state.state["embeddings"].shape == (64, 256)


def test_vqvae_golden_fixture():
    """tests/golden/vqvae_tiny.npz (self-generated by the float64 oracle): indices exact, the rest to f32 accuracy."""
    import os

    from posterior_matching_amd.models.vqvae import VQVAE
    from tests.golden.make_golden_vqvae import CFG

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vqvae_tiny.npz"))
    m = VQVAE(**CFG["model"], device="cuda:0")
    m.init((12, 12, 1))
    m.store.use_bf16 = False
    m.load_params({k[len("param/"):]: z[k] for k in z.files if k.startswith("param/")})
    st = {k[len("state/vq/"):]: z[k] for k in z.files if k.startswith("state/")}
    m.load_state({"embeddings": st["embeddings"], "ema_cluster_size/hidden": st["ema_cluster_size/hidden"],
                  "ema_dw/hidden": st["ema_dw/hidden"], "counter": np.array([int(st["ema_dw/counter"])])})
    got = m(torch.tensor(z["x"]).float().to(dev()), is_training=True)
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    assert np.array_equal(got["vq_output"]["encoding_indices"].cpu().numpy(), z["encoding_indices"])
    assert got["loss"].item() == pytest.approx(float(z["loss"]), rel=2e-5)
    assert got["vq_output"]["perplexity"].item() == pytest.approx(float(z["perplexity"]), rel=1e-5)
    assert rel_err(got["reconstruction"], torch.tensor(z["reconstruction"])) < 2e-5
    for name, g in m.grads_dict().items():
        assert rel_err(g, torch.tensor(z["grad/" + name])) < 1e-4, name
    sd = m.state_dict()
    assert rel_err(sd["embeddings"], torch.tensor(z["new_state/vq/embeddings"])) < 2e-5
    assert rel_err(sd["ema_dw/average"], torch.tensor(z["new_state/vq/ema_dw/average"])) < 2e-5
    assert int(sd["counter"]) == 4
