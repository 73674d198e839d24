"""GPU parity at the CelebA sizes of BASELINE config 5 (reference configs/vqvae_celeb_a.py:14-27 and
configs/pm_vqvae_celeb_a.py:9-38): stage-1 VQ-VAE on 64x64x3 (hidden 128, K = 512, 16x16 code grid) and the
stage-2 step (partial encoder 17.1 M parameters on [x*b | b], 12-resnet PixelCNN 51.0 M parameters) through
the C ABI against the float64 oracles.  Tolerances are written in each test; code indices must be exact."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import pixel_cnn_oracle as PO
from oracle import vqvae_oracle as VO
from tests.ref_configs import pm_vqvae_celeb_a, vqvae_celeb_a
from tests.test_gpu_pixelcnn import _masks, _stage2
from tests.test_gpu_vqvae import _indices_match, _oracle_state

pytestmark = pytest.mark.gpu
XS = (64, 64, 3)


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


def _batch(rng, B):
    """SURVEY.md 8(d) workload C: U[0,1] RGB, one random rectangle (30-100 % of the area) missing per example - the
    dominant component of CelebAMaskGenerator (reference masking.py:107-140,317-325)"""
    x = rng.uniform(size=(B,) + XS)
    b = np.ones((B, 64, 64, 1))
    for i in range(B):
        while True:
            x1, x2 = np.sort(rng.integers(0, 64, 2))
            y1, y2 = np.sort(rng.integers(0, 64, 2))
            if 0.3 * 4096 <= (x2 - x1 + 1) * (y2 - y1 + 1):
                break
        b[i, y1:y2 + 1, x1:x2 + 1] = 0
    return torch.tensor(x), torch.tensor(b)


@pytest.mark.parametrize("bf16x3", [False, True])
def test_celeba_vqvae_forward_and_grads(bf16x3):
    """stage 1 at the reference size: every output of VQVAE.__call__ (vqvae.py:78-96), all 31 gradient tensors and the
    haiku state after the EMA update.  f32 mode: outputs 2e-5, gradients max(5e-5, 10x the float32-CPU oracle's own
    error); bf16x3: outputs 2e-4, gradients 5e-3 (d loss / d activation is ill-conditioned)."""
    from posterior_matching_amd.models.vqvae import VQVAE

    cfg, B, seed = vqvae_celeb_a(), 4, 11
    rng = np.random.default_rng(seed)
    x, _ = _batch(rng, B)
    m = VQVAE(**cfg["model"], device="cuda:0", seed=seed)
    m.init(XS)
    m.store.use_bf16 = bf16x3
    assert m.num_params == 662724
    gen = torch.Generator().manual_seed(seed)
    m.load_params({n: t.cpu() + 0.02 * torch.randn(t.shape, generator=gen) for n, t in m.params_dict().items()})
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    assert {n: tuple(t.shape) for n, t in p64.items()} == VO.param_shapes(cfg["model"], 3)
    st64 = _oracle_state(m.state_dict())
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, aux, out, new_state = VO.vqvae_loss(leaves, st64, cfg, x, True)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))

    got = m(f32d(x), is_training=True)
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    tol = 2e-5 if not bf16x3 else 2e-4
    assert got["z"].shape == (B, 16, 16, 64) and rel_err(got["z"], out["z"]) < tol
    nbad = _indices_match(got["vq_output"]["encoding_indices"], out["z"].detach().reshape(-1, 64), st64["vq/embeddings"],
                          tol=1e-5 if not bf16x3 else 1e-4)
    assert nbad == 0 or bf16x3
    if nbad:
        pytest.skip("bf16x3 forward flipped a near-tie code; remaining comparisons need identical codes")
    assert torch.equal(got["vq_output"]["encoding_indices"].cpu().long(), out["vq_output"]["encoding_indices"])
    assert got["reconstruction"].shape == (B,) + XS and rel_err(got["reconstruction"], out["reconstruction"]) < tol
    assert abs(got["loss"].item() - loss.item()) < tol * abs(loss.item())
    assert abs(got["vq_output"]["perplexity"].item() - aux["perplexity"].item()) < 1e-4 * aux["perplexity"].item()
    gd = m.grads_dict()
    l32 = {n: t.float().clone().requires_grad_(True) for n, t in p64.items()}
    st32 = {k: (v.float() if v.is_floating_point() else v) for k, v in st64.items()}
    loss32 = VO.vqvae_loss(l32, st32, cfg, x.float(), True)[0]
    g32 = dict(zip(l32, torch.autograd.grad(loss32, list(l32.values()))))
    for n in grads:
        e, e32 = rel_err(gd[n], grads[n]), rel_err(g32[n], grads[n])
        if not bf16x3:
            assert e < max(5e-5, 10 * e32) and e < max(2e-4, 3 * e32), (n, e, e32)
        else:
            assert e < 5e-3, (n, e, e32)
    sd = _oracle_state(m.state_dict())
    for k, v in new_state.items():
        if k.endswith("counter"):
            assert int(sd[k]) == int(v)
        else:
            assert rel_err(sd[k], v) < (2e-5 if not bf16x3 else 2e-4), k


def test_celeba_stage2_reference_config_small_batch():
    """configs/pm_vqvae_celeb_a.py network sizes at batch 2, f32 mode: code indices exact, log-prob / loss 2e-5,
    a sample of gradient tensors from every part of the graph 2e-4."""
    cfg, vq_cfg = pm_vqvae_celeb_a(), vqvae_celeb_a()["model"]
    B = 2
    ts, p64, vq64, st64 = _stage2(cfg, vq_cfg, XS, B, seed=6)
    assert ts.num_trainable_params == 51039616 + 17142912            # 68.2 M
    want_shapes = dict(PO.pixel_cnn_param_shapes("pixel_cnn", dict(cfg["pixel_cnn"], num_indices=512), 512))
    want_shapes.update(PO.partial_encoder_param_shapes("partial_encoder", vq_cfg, 4, (16, 16), 512))
    assert {n: tuple(t.shape) for n, t in p64.items()} == want_shapes
    rng = np.random.default_rng(2)
    x, b = _batch(rng, B)
    leaves = {n: t.clone().requires_grad_(True) for n, t in p64.items()}
    loss, idx, lp = PO.pm_vqvae_loss(leaves, vq64, st64, cfg, vq_cfg, x, b, False)
    names = ["pixel_cnn/embed/embeddings", "pixel_cnn/down_5/horizontal/conv2/w", "pixel_cnn/up_11/horizontal/linear/w",
             "pixel_cnn/up_0/vertical/cond/w", "pixel_cnn/down_11/vertical/conv1/w", "partial_encoder/linear/w",
             "partial_encoder/encoder/enc_1/w", "partial_encoder/encoder/res3x3_1/w", "pixel_cnn/out_conv/b"]
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names])))
    ts.set_batch(f32d(x), f32d(b))
    with torch.cuda.stream(ts.stream):
        ll = ts.forward(False)
        from posterior_matching_amd import ops
        ops.neg_mean_loss(ll, 1.0 / B, ts.metrics, ts.g_ll)
        ops.fill_zero(ts.store.flat_g)
        ts.penc.backward(ts.pcnn.backward(ts.g_ll))
    ts.synchronize()
    assert ts._idx.shape == (B, 16, 16) and torch.equal(ts._idx.cpu().long(), idx)
    _compared()
    assert rel_err(ll, lp) < 2e-5 and abs(ts.read_metrics()["loss"] - loss.item()) < 2e-5 * abs(loss.item())
    gd = ts.store.to_dict("g")
    for n in names:
        assert rel_err(gd[n], grads[n]) < 2e-4, (n, rel_err(gd[n], grads[n]))


def test_celeba_stage2_default_mode_step_and_properties():
    """BASELINE's per-GPU batch 16 in the default arithmetic (bf16x3), size-independent properties: the per-example
    log-probs are invariant under a batch permutation (1e-5), the gradient is linear in the upstream gradient (1e-5),
    the bf16x3 log-probs agree with the strict f32 path (1e-4), and two optimizer steps run (loss finite, step counter
    advanced, frozen VQ-VAE untouched)."""
    from posterior_matching_amd import ops

    cfg, vq_cfg = pm_vqvae_celeb_a(), vqvae_celeb_a()["model"]
    B = 16
    ts, p64, vq64, st64 = _stage2(cfg, vq_cfg, XS, B, seed=8, bf16x3=True)
    rng = np.random.default_rng(3)
    x, b = _batch(rng, B)
    xd, bd = f32d(x), f32d(b)
    ts.set_batch(xd, bd)
    with torch.cuda.stream(ts.stream):
        ll = ts.forward(False).clone()
        idx = ts._idx.clone()
    ts.synchronize()
    assert ll.shape == (B,) and torch.isfinite(ll).all() and int(idx.min()) >= 0 and int(idx.max()) < 512
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev())
    ts.set_batch(xd[perm].contiguous(), bd[perm].contiguous())
    with torch.cuda.stream(ts.stream):
        ll_p = ts.forward(False).clone()
    ts.synchronize()
    assert torch.equal(ts._idx, idx[perm]) and rel_err(ll_p, ll[perm]) < 1e-5
    # linearity of the backward pass in g_ll (same forward buffers)
    g1 = torch.full((B,), -1.0 / B, device=dev())
    with torch.cuda.stream(ts.stream):
        ops.fill_zero(ts.store.flat_g)
        ts.penc.backward(ts.pcnn.backward(g1))
        ts.ws.join_aux()
    ts.synchronize()
    ga = ts.store.flat_g.clone()
    torch.cuda.synchronize()            # the copy runs on the default stream: it must finish before ts.stream zeroes flat_g
    with torch.cuda.stream(ts.stream):
        ops.fill_zero(ts.store.flat_g)
        ts.penc.backward(ts.pcnn.backward((2.0 * g1).contiguous()))
        ts.ws.join_aux()
    ts.synchronize()
    assert torch.isfinite(ga).all() and ga.abs().max() > 0 and rel_err(ts.store.flat_g, 2.0 * ga) < 1e-5
    # strict f32 path on the same inputs
    ts.store.use_bf16 = False
    with torch.cuda.stream(ts.stream):
        ll_f32 = ts.forward(False).clone()
    ts.synchronize()
    ts.store.use_bf16 = True
    assert rel_err(ll_p, ll_f32) < 1e-4
    frozen = {n: t.clone() for n, t in ts.vqvae.params_dict().items()}
    ts.dropout_masks = [f32d(m) for m in _masks(rng, cfg, B)]
    for _ in range(2):
        ts.step()
    assert np.isfinite(ts.read_metrics()["loss"]) and ts.step_dev.item() == 2
    for n, t in ts.vqvae.params_dict().items():
        assert torch.equal(t, frozen[n]), n


def test_celeba_two_stage_scripts_end_to_end(tmp_path):
    """train_vqvae.py --config configs/vqvae_celeb_a.py (full stage-1 size) -> train_pm_vqvae.py --config
    configs/pm_vqvae_celeb_a.py (16x16 code grid, K = 512, CelebAMaskGenerator; PixelCNN width reduced for test time)
    -> eval_pm_vqvae.py: the CLI surface of BASELINE config 5."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(script, *argv):
        out = subprocess.run([sys.executable, os.path.join(root, script), *argv], cwd=tmp_path, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        return out.stdout

    run("train_vqvae.py", "--config", os.path.join(root, "configs", "vqvae_celeb_a.py"), "--config.steps=10",
        "--config.validation_freq=10", "--config.seed=1", "--config.data.train_batch_size=16",
        "--config.data.val_batch_size=16")
    runs = os.path.join(tmp_path, "runs")
    stage1 = os.path.join(runs, [d for d in os.listdir(runs) if d.startswith("vqvae-celeb_a")][0])
    assert json.load(open(os.path.join(stage1, "model_config.json")))["num_embeddings"] == 512
    run("train_pm_vqvae.py", "--config", os.path.join(root, "configs", "pm_vqvae_celeb_a.py"),
        f"--config.vqvae_dir={stage1}", "--config.steps=4", "--config.validation_freq=2", "--config.seed=2",
        "--config.pixel_cnn.num_resnet=1", "--config.pixel_cnn.num_filters=32", "--config.conditional_dim=64",
        "--config.data.train_batch_size=8", "--config.data.val_batch_size=8")
    stage2 = os.path.join(runs, [d for d in os.listdir(runs) if d.startswith("pm-vqvae-celeb_a")][0])
    lines = [json.loads(l) for l in open(os.path.join(stage2, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [2, 4] and all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) for l in lines)
    imp = np.load(os.path.join(stage2, "tb", "imputations_4.npy"))
    assert imp.shape == (3, 64, 64 * 7, 3) and imp.min() >= 0.0 and imp.max() <= 1.0
    sys.path.insert(0, root)
    s2 = pickle.load(open(os.path.join(stage2, "train_state.pkl"), "rb"))
    assert s2.step == 4 and s2.state["vqvae/embeddings"].shape == (64, 512)
    out = run("eval_pm_vqvae.py", "--run_dir", stage2, "--num_instances", "8", "--batch_size", "4", "--num_samples", "2")
    res = json.loads(out.strip().splitlines()[-1])
    assert res["num_instances"] == 8 and np.isfinite(res["mean_psnr"]) and 0.0 < res["mean_psnr"] < 60.0
