"""VaDE / PM-VaDE on the device (reference posterior_matching/models/vade.py, train_vade.py, train_pm_vade.py)
against oracle/vade_oracle.py: the mixture kernels, the model's outputs and EVERY gradient tensor, the three train steps, the
two scripts end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vade_oracle as V
from tests.ref_configs import pm_vade_mnist, vade_mnist

pytestmark = pytest.mark.gpu
F64 = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DENSE = {"encoder_net": "ResidualMLP", "decoder_net": "ResidualMLP", "decoder_dist": "IdentityGaussian",
         "decoder_dist_config": {"event_size": 8}, "latent_dim": 4, "num_components": 3,
         "encoder_net_config": {"residual_blocks": 1, "hidden_units": 32},
         "decoder_net_config": {"residual_blocks": 1, "hidden_units": 32},
         "partial_posterior_dist": "TriLGaussian", "partial_posterior_dist_config": {}}     # (AutoregressiveGMM: the mnist cases)


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


@pytest.mark.parametrize("B,k,C", [(37, 10, 10), (5, 4, 3), (130, 16, 64)])
def test_mixture_prior_kernels(B, k, C):
    """pm_vade_prior_fwd / _bwd / pm_vade_cluster_probs vs float64 autograd (2e-6 / 1e-5)"""
    from posterior_matching_amd import ops

    gen = torch.Generator().manual_seed(B + k)
    p = {"vade/mu": torch.randn((C, k), generator=gen, dtype=F64), "vade/log_scale": 0.3 * torch.randn((C, k), generator=gen, dtype=F64),
         "vade/logits": torch.randn((C,), generator=gen, dtype=F64)}
    z, g = torch.randn((B, k), generator=gen, dtype=F64), torch.randn((B,), generator=gen, dtype=F64)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p.items()}
    zr = z.clone().requires_grad_(True)
    lp = torch.logsumexp(V.component_log_probs(leaves, zr) + V.log_pi(leaves), -1)
    grads = torch.autograd.grad((lp * g).sum(), [zr] + list(leaves.values()))
    d = dev()
    mu, ls, lg = f32d(p["vade/mu"]), f32d(p["vade/log_scale"]), f32d(p["vade/logits"])
    out = torch.empty(B, device=d)
    ops.vade_prior_fwd(f32d(z), mu, ls, lg, out)
    assert rel_err(out, lp) < 2e-6
    dz, dmu, dls, dlg = torch.empty((B, k), device=d), torch.zeros((C, k), device=d), torch.zeros((C, k), device=d), torch.zeros(C, device=d)
    ops.vade_prior_bwd(f32d(z), mu, ls, lg, f32d(g), dz, dmu, dls, dlg)
    for got, want in zip((dz, dmu, dls, dlg), grads):
        assert rel_err(got, want) < 1e-5
    S = 7
    zs = torch.randn((B * S, k), generator=gen, dtype=F64)
    want = torch.softmax(V.component_log_probs(p, zs) + V.log_pi(p), -1).reshape(B, S, C).mean(1)
    probs = torch.empty((B, C), device=d)
    ops.vade_cluster_probs(f32d(zs), mu, ls, lg, probs, S)
    assert rel_err(probs, want) < 2e-6


def _model(cfg, xs, partial, seed=3, bf16x3=False):
    from posterior_matching_amd.models.vade import VADE, PosteriorMatchingVADE

    m = (PosteriorMatchingVADE if partial else VADE).from_config(cfg, device="cuda:0", seed=seed)
    m.init(xs)
    m.store.use_bf16 = bf16x3
    gen = torch.Generator().manual_seed(seed)
    vals = {n: t.cpu() + (0.05 * torch.randn(t.shape, generator=gen) if n.endswith("/b") or n == "vade/logits" else 0.0)
            for n, t in m.params_dict().items()}
    vals["vade/log_scale"] = 0.3 * vals["vade/log_scale"]
    m.store.load_dict(vals)
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    if partial:
        m.partial_store.use_bf16 = bf16x3
        p64.update({n: t.cpu().double() for n, t in m.partial_params_dict().items()})
    return m, p64


def _inputs(cfg, xs, B, seed):
    rng = np.random.default_rng(seed)
    if len(xs) == 3:
        x = torch.tensor(rng.uniform(size=(B,) + xs) * (rng.uniform(size=(B,) + xs) < 0.3))
        b = torch.tensor((rng.uniform(size=(B,) + xs[:-1] + (1,)) < 0.5) * 1.0)
    else:
        x = torch.tensor(rng.normal(size=(B,) + xs))
        b = torch.tensor((rng.uniform(size=(B,) + xs) < 0.5) * 1.0)
    return x, b, torch.tensor(rng.normal(size=(B, cfg["latent_dim"])))


@pytest.mark.parametrize("name,B,bf16x3", [("dense", 9, False), ("mnist", 4, False), ("mnist", 4, True)])
def test_vade_elbo_pretrain_and_every_gradient(name, B, bf16x3):
    """VADE.elbo (vade.py:117-150), the pre-training loss (train_vade.py:45-49) and predict_cluster (:96-115): outputs 1e-5
    (bf16x3 1e-4), every gradient tensor 5e-5 (bf16x3 1e-2) against the float64 oracle; parameter names / shapes = the oracle's"""
    from posterior_matching_amd import ops

    cfg, xs = (DENSE, (8,)) if name == "dense" else (vade_mnist()["model"], (28, 28, 1))
    m, p64 = _model(cfg, xs, False, bf16x3=bf16x3)
    assert {n: tuple(t.shape) for n, t in p64.items()} == V.param_shapes(cfg, xs)
    x, _, eps = _inputs(cfg, xs, B, 1)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    want = V.elbo(leaves, cfg, x, eps)
    g = torch.tensor(np.random.default_rng(2).normal(size=(B,)))
    grads = dict(zip(leaves, torch.autograd.grad((want * g).sum(), list(leaves.values()))))
    got = m.elbo(f32d(x), f32d(eps), is_training=True)
    otol, gtol = (1e-4, 1e-2) if bf16x3 else (1e-5, 5e-5)
    assert rel_err(got, want) < otol
    m.zero_grad()
    m.backward_elbo(f32d(g))
    torch.cuda.synchronize()
    for n, gt in m.grads_dict().items():
        assert rel_err(gt, grads[n]) < gtol, (n, rel_err(gt, grads[n]))
    # pre-training autoencoder
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss = V.pretrain_loss(leaves, cfg, x)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)))
    rec = m.reconstruction_ll_at_mean(f32d(x), is_training=True)
    assert abs(-rec.mean().item() - loss.item()) < otol * abs(loss.item())
    m.zero_grad()
    m.backward_reconstruction_at_mean(torch.full((B,), -1.0 / B, device=dev()))
    torch.cuda.synchronize()
    for n, gt in m.grads_dict().items():
        if grads[n] is None:                      # the mixture and the scale half of the head do not enter this loss
            assert float(gt.abs().max()) == 0.0, n
        else:
            assert rel_err(gt, grads[n]) < gtol, (n, rel_err(gt, grads[n]))
    # cluster probabilities under explicit noise
    S = 6
    e = torch.tensor(np.random.default_rng(3).normal(size=(S, B, cfg["latent_dim"])))
    want_p = V.predict_cluster(p64, cfg, x, e)
    got_p = m.predict_cluster(f32d(x), S, eps=f32d(e.permute(1, 0, 2)))
    assert rel_err(got_p, want_p) < (1e-3 if bf16x3 else 1e-5)
    assert torch.allclose(got_p.sum(-1).cpu(), torch.ones(B), atol=1e-5)
    assert rel_err(m.encode_mean(f32d(x)), V.encoder_params(p64, cfg, x)[0]) < otol
    ops.fill_zero(m.store.flat_g)


@pytest.mark.parametrize("name,B", [("dense", 7), ("mnist", 3)])
def test_pm_vade_matching_ll_gradients_and_partial_clusters(name, B):
    """PosteriorMatchingVADE.posterior_matching_ll (vade.py:247-265) and its gradient w.r.t. the partial encoder (the only
    trainable modules, train_pm_vade.py:59-60); partial_predict_cluster (:225-245) under explicit noise"""
    cfg, xs = (DENSE, (8,)) if name == "dense" else (pm_vade_mnist()["model"], (28, 28, 1))
    m, p64 = _model(cfg, xs, True)
    assert {n: tuple(t.shape) for n, t in p64.items()} == V.param_shapes(cfg, xs, partial=True)
    x, b, eps = _inputs(cfg, xs, B, 4)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    want = V.posterior_matching_ll(leaves, cfg, x, b, eps)
    g = torch.tensor(np.random.default_rng(5).normal(size=(B,)))
    names = [n for n in leaves if n.startswith("partial_")]
    grads = dict(zip(names, torch.autograd.grad((want * g).sum(), [leaves[n] for n in names])))
    got = m.posterior_matching_ll(f32d(x), f32d(b), f32d(eps), is_training=True)
    assert rel_err(got, want) < 1e-5
    from posterior_matching_amd import ops
    ops.fill_zero(m.partial_store.flat_g)
    before = m.store.flat_g.clone()
    m.backward_posterior_matching_ll(f32d(g))
    torch.cuda.synchronize()
    for n, gt in m.partial_store.to_dict("g").items():
        assert rel_err(gt, grads[n]) < 1e-4, (n, rel_err(gt, grads[n]))
    assert torch.equal(m.store.flat_g, before)                     # nothing reaches the VaDE's own parameters
    S, k, nc = 5, cfg["latent_dim"], cfg["partial_posterior_dist_config"].get("num_components", 10)
    rng = np.random.default_rng(6)
    noise = {"eps": torch.tensor(rng.normal(size=(B, S, k))),
             "gumbel": torch.tensor(-np.log(-np.log(rng.uniform(1e-6, 1 - 1e-6, size=(B, S, k, nc)))))}
    want_p = V.partial_predict_cluster(p64, cfg, x, b, noise)
    got_p = m.partial_predict_cluster(f32d(x), f32d(b), S, noise={n: f32d(t) for n, t in noise.items()})
    assert rel_err(got_p, want_p) < 1e-4


def test_vade_train_steps_match_oracle():
    """engine.VADETrainStep, both modes, three optimizer steps each on the dense model against the oracle's Adam (optax.adam /
    chain(scale_by_adam(eps), scale_by_schedule(exponential_decay), scale(-1))): losses 1e-4, parameters 1e-3"""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import VADETrainStep

    cfg, xs, B = DENSE, (8,), 16
    m, p64 = _model(cfg, xs, False, seed=5)
    mo, vo = {n: torch.zeros_like(t) for n, t in p64.items()}, {n: torch.zeros_like(t) for n, t in p64.items()}
    ts = VADETrainStep(m, optim.adam(0.002), B, xs, mode="pretrain")
    for step in range(3):
        x, _, _ = _inputs(cfg, xs, B, 10 + step)
        leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
        loss = V.pretrain_loss(leaves, cfg, x)
        gr = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        V.adam_update(p64, {n: (g if g is not None else torch.zeros_like(p64[n])) for n, g in zip(leaves, gr)}, mo, vo, step, 0.002)
        ts.set_batch(f32d(x))
        ts.step()
        _compared()
        assert abs(ts.read_metrics()["loss"] - loss.item()) < 1e-4 * abs(loss.item()), step
    sched = {"init_value": 0.002, "decay_rate": 0.9, "transition_steps": 4}
    opt = optim.chain(optim.scale_by_adam(eps=1e-4), optim.scale_by_schedule(optim.exponential_decay(**sched, staircase=False)),
                      optim.scale(-1.0))
    mo, vo = {n: torch.zeros_like(t) for n, t in p64.items()}, {n: torch.zeros_like(t) for n, t in p64.items()}
    ts2 = VADETrainStep(m, opt, B, xs, mode="elbo", external_eps=True)
    for step in range(3):
        x, _, eps = _inputs(cfg, xs, B, 20 + step)
        leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
        loss = V.vade_loss(leaves, cfg, x, eps)
        gr = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
        V.adam_update(p64, gr, mo, vo, step, V.lr_value(sched, step), eps=1e-4)
        ts2.set_batch(f32d(x), f32d(eps))
        ts2.step()
        _compared()
        assert abs(ts2.read_metrics()["loss"] - loss.item()) < 1e-4 * abs(loss.item()), step
    after = m.params_dict()
    worst = max((rel_err(after[n], p64[n]), n) for n in p64)
    assert worst[0] < 1e-3, worst


def test_pm_vade_train_steps_match_oracle():
    """engine.PMVADETrainStep: three steps; only the partial encoder moves (train_pm_vade.py:59-60)"""
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVADETrainStep

    cfg, xs, B = DENSE, (8,), 12
    m, p64 = _model(cfg, xs, True, seed=6)
    frozen = {n: t.clone() for n, t in m.params_dict().items()}
    mo, vo = {n: torch.zeros_like(t) for n, t in p64.items()}, {n: torch.zeros_like(t) for n, t in p64.items()}
    sched = {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5}
    opt = optim.chain(optim.scale_by_adam(), optim.scale_by_schedule(optim.exponential_decay(**sched)), optim.scale(-1.0))
    ts = PMVADETrainStep(m, opt, B, xs, external_eps=True)
    for step in range(3):
        x, b, eps = _inputs(cfg, xs, B, 30 + step)
        leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
        loss = V.pm_vade_loss(leaves, cfg, x, b, eps)
        names = [n for n in leaves if n.startswith("partial_")]
        gr = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names])))
        V.adam_update(p64, gr, mo, vo, step, V.lr_value(sched, step), trainable=lambda n: n.startswith("partial_"))
        ts.set_batch(f32d(x), f32d(b), f32d(eps))
        ts.step()
        _compared()
        assert abs(ts.read_metrics()["loss"] - loss.item()) < 1e-4 * abs(loss.item()), step
    for n, t in m.params_dict().items():
        assert torch.equal(t, frozen[n]), n
    worst = max((rel_err(t, p64[n]), n) for n, t in m.partial_params_dict().items())
    assert worst[0] < 1e-3, worst


def test_vade_scripts_end_to_end(tmp_path):
    """train_vade.py (pre-training, GMM initialisation, ELBO training with the clustering-accuracy callback) and
    train_pm_vade.py on its run directory, a few steps each"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train_vade.py"), "--config", os.path.join(ROOT, "configs", "vade_mnist.py"),
                          "--config.pretrain_steps=6", "--config.steps=8", "--config.validation_freq=4", "--config.seed=3",
                          "--config.data.train_batch_size=16", "--config.data.val_batch_size=16",
                          "--config.cluster_pred_num_samples=4"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    run = os.path.join(tmp_path, "runs", [d for d in os.listdir(os.path.join(tmp_path, "runs")) if d.startswith("vade-")][0])
    lines = [json.loads(l) for l in open(os.path.join(run, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [4, 8]
    assert all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) and 0.0 <= l["val_clustering_accuracy"] <= 1.0 for l in lines)
    assert os.path.exists(os.path.join(run, "pretrain_state.pkl")) and os.path.exists(os.path.join(run, "model_config.json"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train_pm_vade.py"), "--config", os.path.join(ROOT, "configs", "pm_vade_mnist.py"),
                          f"--config.vade_dir={run}", "--config.steps=6", "--config.validation_freq=3", "--config.seed=4",
                          "--config.data.train_batch_size=16", "--config.data.val_batch_size=16"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    run2 = os.path.join(tmp_path, "runs", [d for d in os.listdir(os.path.join(tmp_path, "runs")) if d.startswith("pm-vade-")][0])
    lines = [json.loads(l) for l in open(os.path.join(run2, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [3, 6] and all(np.isfinite(l["train_loss"]) for l in lines)
