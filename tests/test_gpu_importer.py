"""SURVEY.md 8(f)-4 on the device: a reference-shaped checkpoint (haiku module paths, exported to .npz as
posterior_matching_amd/checkpoint.py describes) -> load_npz / vq_state_to_native -> the product model's outputs equal the
oracle's on the SAME parameter values, for the PM-VAE (conv and dense), the VQ-VAE (+ its haiku state) and the VDVAE
(incl. the "x_bias_{res}]" leaves of vdvae.py:797).  Also here: the four loss objects called as the reference calls its
loss_fn(step, is_training, batch) -> (loss, aux) (train_pm_vae.py:58-72, train_vqvae.py:67-75, train_pm_vqvae.py:81-99,
train_pm_vdvae.py:109-120)."""
import numpy as np
import pytest

from tests import conftest as _conftest
import torch

from oracle import pixel_cnn_oracle as PO
from oracle import pm_vae_oracle as O
from oracle import vdvae_oracle as DO
from oracle import vqvae_oracle as VO
from tests.haiku_names import to_tree, vae_names, vdvae_names, vq_names, vq_state_tree
from tests.ref_configs import pm_vae_gas, pm_vae_mnist, vqvae_mnist
from tests.test_gpu_vdvae import TINY

pytestmark = pytest.mark.gpu
F64 = torch.float64


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    from tests import conftest

    conftest.confirm_compared()          # the kernels launched so far in this test have a compared result
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def f32d(t):
    return t.float().to(dev())


def _export(tmp_path, name, native, to_haiku):
    """what the three-line jax-side export of checkpoint.py's docstring writes: flat {module/leaf: array}"""
    tree = to_tree({n: v.numpy().astype(np.float32) for n, v in native.items()}, to_haiku)
    flat = {f"{mod}/{leaf}": v for mod, d in tree.items() for leaf, v in d.items()}
    path = tmp_path / f"{name}.npz"
    np.savez(path, **flat)
    return str(path)


def _perturbed(p, seed):
    gen = torch.Generator().manual_seed(seed)       # biases / log_scale are zero at init: move every leaf
    return {n: (t + 0.05 * torch.randn(t.shape, generator=gen, dtype=F64)).float().double() for n, t in p.items()}


@pytest.mark.parametrize("name", ["mnist", "gas"])
def test_pm_vae_checkpoint_import(tmp_path, name):
    from posterior_matching_amd.checkpoint import load_npz
    from posterior_matching_amd.models import PosteriorMatchingVAE

    cfg, xs = (pm_vae_mnist(), (28, 28, 1)) if name == "mnist" else (pm_vae_gas(), (8,))
    p64 = _perturbed(O.init_params(cfg["model"], xs, seed=11), 1)
    path = _export(tmp_path, "pm_vae", p64, vae_names)
    with np.load(path) as z:
        assert any("conv2_d_1/" in k or "linear_1/" in k for k in z.files) and not any("/conv_1/" in k for k in z.files)
    m = PosteriorMatchingVAE.from_config(cfg["model"], device="cuda:0", seed=99)      # different values before the import
    m.init(xs)
    m.store.use_bf16 = False
    load_npz(m, path)
    B = 6
    rng = np.random.default_rng(2)
    x = torch.tensor(rng.uniform(size=(B,) + xs))
    b = torch.tensor((rng.uniform(size=(B,) + (xs[:-1] + (1,) if len(xs) == 3 else xs)) < 0.5) * 1.0)
    eps = torch.tensor(rng.normal(size=(B, cfg["model"]["latent_dim"])))
    want = O.pm_vae_forward(p64, cfg["model"], x, b, eps)
    got = m(f32d(x), f32d(b), False, eps=f32d(eps))
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        assert rel_err(got[key], want[key]) < 1e-5, key


@_conftest.compares
def test_vqvae_checkpoint_and_state_import(tmp_path):
    from posterior_matching_amd.checkpoint import load_npz, vq_state_to_native
    from posterior_matching_amd.models.vqvae import VQVAE

    cfg = vqvae_mnist()
    p64 = _perturbed(VO.init_params(cfg["model"], 1, seed=5), 3)
    st64 = VO.init_state(cfg["model"], seed=7)
    rng = np.random.default_rng(4)
    st64["vq/ema_cluster_size/hidden"] = torch.tensor(rng.uniform(size=st64["vq/ema_cluster_size/hidden"].shape))
    for ema in ("ema_cluster_size", "ema_dw"):
        st64[f"vq/{ema}/counter"] = torch.tensor(9)
    path = _export(tmp_path, "vqvae", p64, vq_names)
    m = VQVAE(**cfg["model"], device="cuda:0", seed=77)
    m.init((28, 28, 1))
    m.store.use_bf16 = False
    load_npz(m, path)
    m.load_state(vq_state_to_native({mod: {k: v.numpy() for k, v in d.items()} for mod, d in vq_state_tree(st64).items()}))
    assert int(m.state_dict()["counter"]) == 9
    x = torch.tensor(rng.uniform(size=(5, 28, 28, 1)) * (rng.uniform(size=(5, 28, 28, 1)) < 0.3))
    loss, aux, out, _ = VO.vqvae_loss(p64, st64, cfg, x, False)
    got = m(f32d(x), is_training=False)
    torch.cuda.synchronize()
    assert torch.equal(got["vq_output"]["encoding_indices"].cpu().long().reshape(-1), out["vq_output"]["encoding_indices"].reshape(-1))
    assert abs(got["loss"].item() - loss.item()) < 2e-5 * abs(loss.item())


def test_vdvae_checkpoint_import(tmp_path):
    from posterior_matching_amd.checkpoint import load_npz
    from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE

    p64 = _perturbed(DO.init_params(TINY["model"], seed=3), 8)
    path = _export(tmp_path, "vdvae", p64, vdvae_names)
    with np.load(path) as z:
        assert any(k.endswith("]") for k in z.files)                      # the stray bracket of vdvae.py:797
    m = PosteriorMatchingVDVAE(**TINY["model"], device="cuda:0", seed=55)
    m.init()
    m.store.use_bf16 = False
    load_npz(m, path)
    B = 3
    rng = np.random.default_rng(6)
    x = torch.tensor(np.round(rng.uniform(size=(B, 7, 7, 1)) * 255.0))
    b = torch.tensor((rng.uniform(size=(B, 7, 7, 1)) < 0.5) * 1.0)
    eps = [torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)]
    loss, aux, out = DO.vdvae_loss(p64, TINY, x, b, eps)
    got = m(f32d(x), f32d(b), [f32d(e) for e in eps])
    for k in ("reconstruction_ll", "kl", "pm_kl"):
        assert rel_err(got[k], out[k]) < 2e-5, k


# ----------------------------------------------------------------------------------------------
# the loss objects as functions: loss_fn(step, is_training, batch) -> (loss, aux)
# ----------------------------------------------------------------------------------------------
@_conftest.compares
def test_pm_vae_loss_fn_is_callable_like_the_reference():
    from posterior_matching_amd.models import PosteriorMatchingVAE
    from posterior_matching_amd.trainer import PMVAELoss

    cfg, xs, B = pm_vae_gas(), (8,), 37
    m = PosteriorMatchingVAE.from_config(cfg["model"], device="cuda:0", seed=3)
    m.init(xs)
    m.store.use_bf16 = False
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    rng = np.random.default_rng(1)
    x, b = torch.tensor(rng.normal(size=(B,) + xs)), torch.tensor((rng.uniform(size=(B,) + xs) < 0.5) * 1.0)
    eps = torch.tensor(rng.normal(size=(B, 16)))
    loss_fn = PMVAELoss(cfg, m, "features")
    batch = {"features": x.float(), "mask": b.float()}                     # host tensors, as a data pipeline yields them
    for step in (0, 13500, 60000):                                         # cyclic beta (train_pm_vae.py:28-43): 0 during the delay, then the ramp
        want, aux, _ = O.pm_vae_loss(p64, cfg, x, b, eps, step)
        loss, got = loss_fn(step, False, batch, eps=f32d(eps))
        assert loss.ndim == 0 and abs(loss.item() - want.item()) < 1e-5 * abs(want.item()), step
        for k in ("reconstruction_ll", "kl", "matching_ll"):
            assert abs(got[k].item() - float(aux[k])) < 1e-5 * abs(float(aux[k])), (step, k)
        assert abs(got["beta"].item() - O.beta_value(cfg, step)) < 1e-6
    l1, _ = loss_fn(5, True, batch)                                        # own noise: keyed by (seed, step)
    l2, _ = loss_fn(5, True, batch)
    l3, _ = loss_fn(6, True, batch)
    assert l1.item() == l2.item() != l3.item()


@_conftest.compares
def test_vqvae_and_stage2_loss_fns_are_callable():
    from posterior_matching_amd.models.vqvae import VQVAE, build_partial_posterior
    from posterior_matching_amd.trainer import PMVQVAELoss, VQVAELoss
    from tests.test_gpu_pixelcnn import TINY_CFG, TINY_VQ

    xs, B = (12, 12, 1), 4
    vq = VQVAE(**TINY_VQ, device="cuda:0", seed=4)
    vq.init(xs)
    vq.store.use_bf16 = False
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.uniform(size=(B,) + xs) * (rng.uniform(size=(B,) + xs) < 0.3))
    b = torch.tensor((rng.uniform(size=(B, 12, 12, 1)) < 0.5) * 1.0)
    vq64 = {n: t.cpu().double() for n, t in vq.params_dict().items()}
    sd = vq.state_dict()
    st64 = {"vq/embeddings": sd["embeddings"].cpu().double()}
    for name in ("ema_cluster_size", "ema_dw"):
        st64[f"vq/{name}/hidden"] = sd[f"{name}/hidden"].cpu().double()
        st64[f"vq/{name}/average"] = sd[f"{name}/average"].cpu().double()
        st64[f"vq/{name}/counter"] = torch.tensor(0)
    want, aux, _, _ = VO.vqvae_loss(vq64, st64, {"model": TINY_VQ}, x, False)
    loss, got = VQVAELoss({"model": TINY_VQ}, vq)(0, False, {"image": x.float()})
    assert abs(loss.item() - want.item()) < 2e-5 * abs(want.item())
    assert set(got) == {"perplexity", "reconstruction_loss", "vq_loss"}
    for k in got:
        assert abs(got[k].item() - float(aux[k])) < 1e-4 * abs(float(aux[k])) + 1e-7, k
    assert int(vq.state_dict()["counter"]) == 0                           # is_training=False: no EMA update

    penc, pcnn, store = build_partial_posterior(vq, TINY_CFG["conditional_dim"], TINY_CFG["pixel_cnn"], xs, seed=4)
    store.use_bf16 = False
    p64 = {n: t.cpu().double() for n, t in store.to_dict("p").items()}
    want2, _, _ = PO.pm_vqvae_loss(p64, vq64, st64, TINY_CFG, TINY_VQ, x, b, False)
    loss2, aux2 = PMVQVAELoss(TINY_CFG, vq, penc, pcnn)(0, False, {"image": x.float(), "mask": b.float()})
    assert aux2 == {} and abs(loss2.item() - want2.item()) < 2e-5 * abs(want2.item())


@_conftest.compares
def test_vdvae_loss_fn_is_callable():
    from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE
    from posterior_matching_amd.trainer import VDVAELoss

    m = PosteriorMatchingVDVAE(**TINY["model"], device="cuda:0", seed=5)
    m.init()
    m.store.use_bf16 = False
    p64 = {n: t.cpu().double() for n, t in m.params_dict().items()}
    B = 3
    rng = np.random.default_rng(7)
    x = torch.tensor(np.round(rng.uniform(size=(B, 7, 7, 1)) * 255.0))
    b = torch.tensor((rng.uniform(size=(B, 7, 7, 1)) < 0.5) * 1.0)
    eps = [torch.tensor(rng.normal(size=s)) for s in m.eps_shapes(B)]
    want, aux, _ = DO.vdvae_loss(p64, TINY, x, b, eps)
    loss, got = VDVAELoss(TINY, m)(0, True, {"image": x.float(), "mask": b.float()}, eps=[f32d(e) for e in eps])
    assert abs(loss.item() - want.item()) < 2e-5 * abs(want.item())
    for k in ("reconstruction_ll", "kl", "pm_kl", "bpd"):
        assert abs(got[k].item() - float(aux[k])) < 2e-5 * abs(float(aux[k])), k
    l1, _ = VDVAELoss(TINY, m, seed=2)(4, True, {"image": x.float(), "mask": b.float()})
    assert np.isfinite(l1.item())
