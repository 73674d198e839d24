"""Lookahead posteriors on the device (reference posterior_matching/models/lookahead.py,
train_lookahead_posterior.py) against oracle/lookahead_oracle.py: the model-specific kernels, the one-step-ahead latent samples,
lookahead_lls and EVERY gradient tensor of the lookahead encoder, expected_info_gains, the train step, the scripts end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import conftest as _conftest
import torch

from oracle import lookahead_oracle as L
from oracle import pm_vae_oracle as O
from tests.ref_configs import lookahead_mnist16, pm_vae_mnist16

pytestmark = pytest.mark.gpu
F64 = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a small PM-VAE of the mnist16 kind (8 x 8 images) and the reference's configuration itself
SMALL = {"latent_dim": 4, "encoder_net": "ConvEncoder", "decoder_net": "ConvDecoder", "posterior_dist": "TriLGaussian",
         "decoder_dist": "Bernoulli", "encoder_net_config": {"conv_layers": [(8, 3, 1), (8, 3, 2), (16, 4, 1)]},
         "decoder_net_config": {"conv_layers": [(16, 4, 1), (8, 3, 2), (1, 3, 1)]}}
CASES = {"small": (SMALL, (8, 8, 1), {"lookahead_subsample": 5, "model_samples": 3}),
         "mnist16": (pm_vae_mnist16()["model"], (16, 16, 1), {"lookahead_subsample": 16, "model_samples": 4})}


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


def _setup(name, seed=3):
    from posterior_matching_amd.models.lookahead import LookaheadPosterior

    pm_cfg, xs, look = CASES[name]
    look = dict(look, num_features=xs[0] * xs[1])
    pv, pl = O.init_params(pm_cfg, xs, seed=seed), L.init_params(look, pm_cfg, xs, seed=seed + 1)
    rng = np.random.default_rng(seed)
    pl["lookahead_block/linear/b"] = torch.tensor(0.3 * rng.normal(size=pl["lookahead_block/linear/b"].shape))
    m = LookaheadPosterior.from_config(look, pm_cfg, device=dev(), seed=seed)
    m.init(xs)
    m.load_params({**pv, **pl})
    return m, pm_cfg, look, xs, pv, pl


def _inputs(xs, look, k, B, seed):
    rng = np.random.default_rng(seed)
    Z, S, F = look["model_samples"], look["lookahead_subsample"], look["num_features"]
    x = torch.tensor(rng.uniform(size=(B,) + xs) * (rng.uniform(size=(B,) + xs) < 0.4))
    b = torch.tensor((rng.uniform(size=(B,) + xs[:-1] + (1,)) < 0.3).astype(np.float64))
    noise = {"eps": torch.tensor(rng.normal(size=(B, Z, k))), "eps_look": torch.tensor(rng.normal(size=(B, Z, S, k)))}
    inds = rng.choice(F, size=S, replace=False)
    return x, b, noise, inds


@pytest.mark.parametrize("B,F,Z,S,k", [(5, 64, 3, 5, 4), (32, 256, 64, 16, 10), (3, 9, 2, 9, 64)])
def test_lookahead_ll_kernels(B, F, Z, S, k):
    """pm_lookahead_ll_fwd / _bwd vs float64 autograd of oracle.masked_mean_ll (2e-6 / 1e-5), incl. an example with no valid
    feature (ll = 0, zero gradient) and rows of dparams outside the subsample (zero)"""
    from posterior_matching_amd import ops

    rng = np.random.default_rng(B + F)
    params = torch.tensor(rng.normal(size=(B, F, 2 * k)))
    zs, g = torch.tensor(rng.normal(size=(B, Z, S, k))), torch.tensor(rng.normal(size=(B,)))
    inds = rng.choice(F, size=S, replace=False)
    b = torch.tensor((rng.uniform(size=(B, F)) < 0.4).astype(np.float64))
    b[0, inds] = 1.0                                                         # example 0: every subsampled feature observed
    valid = b[:, list(inds)] + 1.0 < 2.0
    pr = params.clone().requires_grad_(True)
    want = L.masked_mean_ll(pr, zs, valid, inds)
    (dwant,) = torch.autograd.grad((want * g).sum(), pr)
    d = dev()
    it = torch.tensor(inds.astype(np.int32), device=d)
    ll, dp = torch.empty(B, device=d), torch.full((B, F, 2 * k), 7.0, device=d)
    ops.lookahead_ll_fwd(f32d(params), it, f32d(zs), f32d(b), ll)
    assert rel_err(ll, want) < 2e-6 and ll[0].item() == 0.0
    ops.lookahead_ll_bwd(f32d(params), it, f32d(zs), f32d(b), f32d(g), dp)
    assert rel_err(dp, dwant) < 1e-5
    rest = np.setdiff1d(np.arange(F), inds)
    assert dp[0].abs().max().item() == 0.0 and (rest.size == 0 or dp[:, rest].abs().max().item() == 0.0)


@_conftest.compares
def test_lookahead_inputs_kernel():
    """pm_lookahead_inputs: rows (b, z, s) = [samples * max(b, one-hot) | max(b, one-hot)], bit-exact (lookahead.py:154-176)"""
    from posterior_matching_amd import ops

    rng = np.random.default_rng(2)
    B, Z, S, H, W, C = 3, 4, 5, 4, 6, 2
    imp = torch.tensor(rng.uniform(size=(B, Z, H, W, C)), dtype=torch.float32)
    b = torch.tensor((rng.uniform(size=(B, H, W, 1)) < 0.3).astype(np.float32))
    inds = rng.choice(H * W, size=S, replace=False)
    one = torch.eye(H * W)[list(inds)].reshape(S, H, W, 1)
    bl = torch.maximum(b[:, None], one[None])                                # [B, S, H, W, 1]
    want = torch.cat([imp[:, :, None] * bl[:, None], bl[:, None].expand(B, Z, S, H, W, 1)], -1).reshape(B * Z * S, H, W, C + 1)
    out = torch.empty((B * Z * S, H, W, C + 1), device=dev())
    ops.lookahead_inputs(imp.to(dev()), b.to(dev()), torch.tensor(inds.astype(np.int32), device=dev()), out)
    assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("name,B", [("small", 5), ("mnist16", 3)])
def test_lookahead_lls_and_every_gradient(name, B):
    """LookaheadPosterior.__call__ with explicit draws: the one-step-ahead samples (3e-5), lookahead_lls (1e-4) and the gradient
    of every lookahead-encoder parameter (1e-3 of the largest tensor norm) vs the oracle; the PM-VAE receives no gradient"""
    m, pm_cfg, look, xs, pv, pl = _setup(name)
    k = pm_cfg["latent_dim"]
    x, b, noise, inds = _inputs(xs, look, k, B, 11)
    leaves = {n: t.clone().requires_grad_(True) for n, t in pl.items()}
    zs_want, _ = L.model_one_step_z(pv, pm_cfg, x, b, noise, inds)
    want = L.lookahead_lls(leaves, pv, look, pm_cfg, x, b, noise, inds)
    g = torch.tensor(np.random.default_rng(1).normal(size=(B,)))
    grads = dict(zip(leaves, torch.autograd.grad((want * g).sum(), list(leaves.values()))))
    from posterior_matching_amd import ops

    it = torch.tensor(inds.astype(np.int32), device=dev())
    nd = {n: f32d(t) for n, t in noise.items()}
    got = m(f32d(x), f32d(b), is_training=True, noise=nd, inds=it)
    assert rel_err(m._saved[2], zs_want) < 3e-5
    assert rel_err(got, want) < 1e-4
    ops.fill_zero(m.store.flat_g)
    ops.fill_zero(m.pm_vae.store.flat_g)
    m.backward(f32d(g))
    torch.cuda.synchronize()
    gd = m.store.to_dict("g")
    scale = max(t.norm().item() for t in grads.values())
    for n, t in grads.items():
        err = (gd[n].double().cpu() - t).norm().item()
        assert err < 1e-3 * max(t.norm().item(), 1e-3 * scale), (n, err, t.norm().item())
    assert m.pm_vae.store.flat_g.abs().max().item() == 0.0


@_conftest.compares
def test_lookahead_expected_info_gains():
    m, pm_cfg, look, xs, pv, pl = _setup("small")
    x, b, _, _ = _inputs(xs, look, pm_cfg["latent_dim"], 1, 5)
    want = L.expected_info_gains(pl, pv, look, pm_cfg, x[0], b[0])
    got = m.expected_info_gains(f32d(x[0]), f32d(b[0])).cpu().double()
    obs = b[0].reshape(-1) == 1
    assert torch.equal(torch.isinf(got), obs) and (got[obs] < 0).all()
    assert (got[~obs] - want[~obs]).abs().max().item() < 2e-4


def test_lookahead_train_steps_match_oracle():
    """engine.LookaheadTrainStep: three optimizer steps against the oracle's Adam + exponential decay (loss 1e-4, parameters
    1e-3); the PM-VAE's parameters do not move (train_lookahead_posterior.py:61-62)"""
    from oracle import vade_oracle as V
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import LookaheadTrainStep

    m, pm_cfg, look, xs, pv, pl = _setup("small", seed=8)
    k, B = pm_cfg["latent_dim"], 6
    frozen = {n: t.clone() for n, t in m.pm_vae.params_dict().items()}
    mo, vo = {n: torch.zeros_like(t) for n, t in pl.items()}, {n: torch.zeros_like(t) for n, t in pl.items()}
    sched = {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5}
    opt = optim.chain(optim.scale_by_adam(), optim.scale_by_schedule(optim.exponential_decay(**sched)), optim.scale(-1.0))
    ts = LookaheadTrainStep(m, opt, B, xs, external_noise=True)
    for step in range(3):
        x, b, noise, inds = _inputs(xs, look, k, B, 40 + step)
        leaves = {n: t.clone().requires_grad_(True) for n, t in pl.items()}
        loss = L.loss(leaves, pv, look, pm_cfg, x, b, noise, inds)
        gr = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
        V.adam_update(pl, gr, mo, vo, step, V.lr_value(sched, step))
        ts.set_batch(f32d(x), f32d(b), {n: f32d(t) for n, t in noise.items()}, torch.tensor(inds.astype(np.int32), device=dev()))
        ts.step()
        _compared()
        assert abs(ts.read_metrics()["loss"] - loss.item()) < 1e-4 * abs(loss.item()), step
    for n, t in m.pm_vae.params_dict().items():
        assert torch.equal(t, frozen[n]), n
    after = m.store.to_dict("p")
    worst = max((rel_err(after[n], pl[n]), n) for n in pl)
    assert worst[0] < 1e-3, worst


@_conftest.compares
def test_acquisition_kernels():
    """pm_acquisition_policy (argmax with ties and -inf entries, softmax of the -1e10-filled logits) and pm_reconstruction_rmse
    vs numpy"""
    from posterior_matching_amd import ops

    rng = np.random.default_rng(9)
    for F in (9, 256, 700):
        g = rng.normal(size=F).astype(np.float32)
        g[rng.uniform(size=F) < 0.3] = -np.inf
        g[[F // 2, F // 3]] = 5.0                                              # a tie: the lower index wins
        gd, probs, act = torch.tensor(g, device=dev()), torch.empty(F, device=dev()), torch.empty(1, dtype=torch.int32, device=dev())
        ops.acquisition_policy(gd, probs, act)
        logits = np.where(np.isinf(g), -1e10, g).astype(np.float64)
        want = np.exp(logits - logits.max())
        want /= want.sum()
        assert act.item() == F // 3 and np.allclose(probs.cpu().numpy(), want, rtol=2e-6, atol=1e-12)
    allobs = torch.full((6,), -np.inf, device=dev())
    probs, act = torch.empty(6, device=dev()), torch.empty(1, dtype=torch.int32, device=dev())
    ops.acquisition_policy(allobs, probs, act)
    assert act.item() == 0 and np.allclose(probs.cpu().numpy(), 1.0 / 6)
    for shape, mshape in (((5, 7, 3), (5, 7, 1)), ((11,), (11,))):
        S = 4
        imp, x = rng.uniform(size=(S,) + shape).astype(np.float32), rng.uniform(size=shape).astype(np.float32)
        b = (rng.uniform(size=mshape) < 0.4).astype(np.float32)
        rec, err = torch.empty(shape, device=dev()), torch.empty(1, device=dev())
        ops.reconstruction_rmse(torch.tensor(imp, device=dev()), torch.tensor(x, device=dev()), torch.tensor(b, device=dev()), rec, err)
        m = imp.astype(np.float64).mean(0)
        assert np.allclose(rec.cpu().numpy(), m, rtol=1e-6) and abs(err.item() - np.sqrt(np.mean((x - m) ** 2 * (1 - b)))) < 1e-6


def test_acquisition_eval_fn_and_trajectories():
    """acquisition.make_acquisition_eval_fn on one instance with explicit draws vs the oracle (sampling-based and lookahead
    information gains -> policies, mean imputation), then two 3-step episodes: masks grow by the chosen one-hots"""
    from posterior_matching_amd.acquisition import make_acquisition_eval_fn, make_collect_trajectory_fn, rmse

    m, pm_cfg, look, xs, pv, pl = _setup("small", seed=12)
    k, S = pm_cfg["latent_dim"], 6
    x, b, _, _ = _inputs(xs, look, k, 1, 21)
    x, b = x[0], b[0]
    noise = {"eps": torch.tensor(np.random.default_rng(4).normal(size=(1, S, k)))}
    eval_fn = make_acquisition_eval_fn(look, pm_cfg, S, model=m)
    out = eval_fn(f32d(x * b), f32d(b), noise={n: f32d(t) for n, t in noise.items()})
    gs = O.pm_vae_expected_info_gains(pv, pm_cfg, x * b, b, noise)
    gl = L.expected_info_gains(pl, pv, look, pm_cfg, x * b, b)
    for name, g in (("sampling", gs), ("lookahead", gl)):
        logits = torch.where(torch.isinf(g), torch.full_like(g, -1e10), g)
        assert rel_err(out[f"{name}_probs"], torch.softmax(logits, 0)) < 2e-4, name
        top = torch.sort(logits, descending=True).values
        if (top[0] - top[1]).item() > 1e-4:
            assert out[f"{name}_action"].item() == int(torch.argmax(logits)), name
    want_rec = O.pm_vae_impute(pv, pm_cfg, (x * b)[None], b[None], noise).mean(0)[0]
    assert rel_err(out["reconstruction"], want_rec) < 1e-4
    want_rmse = torch.sqrt(((x - want_rec) ** 2 * (1 - b)).mean())
    assert abs(rmse(f32d(x), out["reconstruction"], f32d(b)).item() - want_rmse.item()) < 1e-4
    T = 3
    samp, lk = make_collect_trajectory_fn(eval_fn, T)(f32d(x))
    for traj, which in ((samp, "sampling"), (lk, "lookahead")):
        assert traj["mask"].shape == (T,) + tuple(b.shape) and traj["reconstruction"].shape == (T,) + tuple(x.shape)
        assert traj["sampling_probs"].shape == (T, look["num_features"]) and np.isfinite(traj["rmse"]).all()
        assert traj["mask"][0].sum() == 0
        for t in range(T - 1):
            one = np.zeros(look["num_features"], np.float32)
            one[traj[f"{which}_action"][t]] = 1.0
            assert np.array_equal(traj["mask"][t + 1], traj["mask"][t] + one.reshape(b.shape))
        assert len(set(traj[f"{which}_action"].tolist())) == T                      # never re-acquires an observed feature


def test_lookahead_scripts_end_to_end(tmp_path):
    """train_pm_vae.py on configs/pm_vae_mnist16.py, then train_lookahead_posterior.py on its run directory with the reference's
    configuration (batch 32, 64 model samples, 16 lookahead features), a few steps each; device Philox noise"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train_pm_vae.py"), "--config", os.path.join(ROOT, "configs", "pm_vae_mnist16.py"),
                          "--config.steps=4", "--config.validation_freq=4", "--config.seed=3", "--config.data.train_batch_size=32",
                          "--config.data.val_batch_size=32"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    run = os.path.join(tmp_path, "runs", [d for d in os.listdir(os.path.join(tmp_path, "runs")) if d.startswith("pm-vae-")][0])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train_lookahead_posterior.py"), "--config",
                          os.path.join(ROOT, "configs", "lookahead_mnist16.py"), f"--config.pm_vae_dir={run}", "--config.steps=6",
                          "--config.validation_freq=3", "--config.seed=4"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    run2 = os.path.join(tmp_path, "runs", [d for d in os.listdir(os.path.join(tmp_path, "runs")) if d.startswith("lookahead-")][0])
    lines = [json.loads(l) for l in open(os.path.join(run2, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [3, 6]
    assert all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) for l in lines)
    assert os.path.exists(os.path.join(run2, "lookahead_config.json")) and os.path.exists(os.path.join(run2, "pm_vae_config.json"))
    assert json.load(open(os.path.join(run2, "lookahead_config.json")))["num_features"] == 256
    out = subprocess.run([sys.executable, os.path.join(ROOT, "eval_greedy_acquisition.py"), "--run_dir", run2, "--dataset", "mnist16",
                          "--num_instances", "2", "--num_samples", "3", "--episode_length", "3"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    import pickle

    for kind in ("sampling", "lookahead"):
        trajs = pickle.load(open(os.path.join(run2, "trajectories", f"{kind}_trajectories.pkl"), "rb"))
        assert len(trajs) == 2 and trajs[0]["truth"].shape == (16, 16, 1) and trajs[0]["mask"].shape == (3, 16, 16, 1)
        assert trajs[0]["lookahead_probs"].shape == (3, 256) and np.isfinite(trajs[1]["rmse"]).all()
