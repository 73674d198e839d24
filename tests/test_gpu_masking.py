"""Device-side mask generation (csrc/pm_mask.hip) against oracle/masking_oracle.py: bit-exact streams, plus the
distributional properties each reference generator defines (reference masking.py:24-286)."""
import numpy as np
import pytest
import torch

from oracle import masking_oracle as MO

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("name,B,H,seed", [("MNISTMaskGenerator", 64, 28, 1), ("OmniglotMaskGenerator", 33, 28, 2),
                                           ("Cifar10MaskGenerator", 40, 32, 3), ("MNISTMaskGenerator", 17, 14, 4)])
def test_image_mixture_bit_exact(name, B, H, seed):
    from posterior_matching_amd import ops
    from posterior_matching_amd.masking import get_mask_generator

    dim = H if name == "MNISTMaskGenerator" else None
    gen = get_mask_generator(name, device=dev(), seed=seed, **({"dim": dim} if dim else {}))
    comps = MO.image_mixture_components(name, dim)
    for step in range(3):
        desc = torch.zeros((B, 6), dtype=torch.int32, device=dev())
        out = torch.empty((B, H, H, 1), device=dev())
        gen.fill(out, desc_out=desc)
        gen._advance()
        want, wdesc = MO.image_mask_mixture(B, H, H, comps, seed, step=step)
        assert np.array_equal(desc.cpu().numpy(), wdesc), step
        assert np.array_equal(out.cpu().numpy(), want), step
    # the public call path advances the stream by itself
    m = gen((B, H, H, 1))
    assert np.array_equal(m.cpu().numpy(), MO.image_mask_mixture(B, H, H, comps, seed, step=3)[0])


def test_image_mixture_distribution_full_batch():
    """4096 MNIST masks: component frequencies = weights/10, rectangles inside their area bounds, squares 14x14,
    half planes exact, pixel-Bernoulli mean 0.5 (reference masking.py:235-249)."""
    from posterior_matching_amd.masking import get_mask_generator

    B, H = 4096, 28
    gen = get_mask_generator("MNISTMaskGenerator", device=dev(), seed=11)
    desc = torch.zeros((B, 6), dtype=torch.int32, device=dev())
    out = torch.empty((B, H, H, 1), device=dev())
    gen.fill(out, desc_out=desc)
    d, m = desc.cpu().numpy(), out.cpu().numpy()[..., 0]
    assert set(np.unique(m)) <= {0.0, 1.0}
    freq = np.bincount(d[:, 5], minlength=7) / B
    assert np.abs(freq - np.array([2, 1, 1, 1, 1, 2, 2]) / 10).max() < 0.03
    missing = (1 - m).sum((1, 2))
    bern = d[:, 5] == 0
    assert abs(m[bern].mean() - 0.5) < 0.01
    for ci, (y1, x1, y2, x2) in zip(range(1, 5), [(0, 0, 28, 14), (0, 0, 14, 28), (0, 14, 28, 28), (14, 0, 28, 28)]):
        sel = m[d[:, 5] == ci]
        want = np.ones((28, 28))
        want[y1:y2, x1:x2] = 0
        assert (sel == want).all()
    sq = d[:, 5] == 5
    assert (missing[sq] == 14 * 14).all()
    assert d[sq, 1].min() >= 0 and d[sq, 1].max() <= 13 and d[sq, 2].max() <= 13      # randint(28 - 14): 0..13
    assert len(np.unique(d[sq, 2])) == 14
    rc = d[:, 5] == 6
    area = (d[rc, 3] - d[rc, 1]) * (d[rc, 4] - d[rc, 2])
    assert (area == missing[rc]).all()
    assert area.min() >= 0.3 * 784 and area.max() <= 784


@pytest.mark.parametrize("shape,p", [((37, 8), 0.5), ((5, 28, 28, 1), 0.2), ((3, 1001), 0.9)])
def test_bernoulli_mask_bit_exact(shape, p):
    from posterior_matching_amd.masking import get_mask_generator

    gen = get_mask_generator("BernoulliMaskGenerator", device=dev(), seed=5, p=p)
    for step in range(2):
        got = gen(shape).cpu().numpy()
        assert np.array_equal(got, MO.bernoulli_mask(shape, p, 5, step=step))
    big = get_mask_generator("BernoulliMaskGenerator", device=dev(), seed=6, p=p)((4096, 512)).mean().item()
    assert abs(big - p) < 2e-3


@pytest.mark.parametrize("B,D,bounds", [(64, 8, None), (31, 43, None), (16, 300, (0.25, 0.5))])
def test_uniform_mask_bit_exact(B, D, bounds):
    from posterior_matching_amd.masking import get_mask_generator

    gen = get_mask_generator("UniformMaskGenerator", device=dev(), seed=9, bounds=bounds)
    lo, span = (0, D) if bounds is None else (int(D * bounds[0]), int(D * bounds[1]))
    got = gen((B, D)).cpu().numpy()
    assert np.array_equal(got, MO.uniform_mask(B, D, lo, span, 9, step=0))
    counts = got.sum(1)
    assert counts.min() >= lo and counts.max() <= min(D, lo + span - 1)


def test_uniform_mask_counts_are_uniform():
    from posterior_matching_amd.masking import get_mask_generator

    got = get_mask_generator("UniformMaskGenerator", device=dev(), seed=1)((8192, 8)).cpu().numpy()
    freq = np.bincount(got.sum(1).astype(int), minlength=8) / 8192           # q = choice(8): 0..7
    assert got.sum(1).max() <= 7 and np.abs(freq - 1 / 8).max() < 0.02
    assert np.abs(got.mean(0) - got.mean()).max() < 0.02                          # every feature equally likely


def test_dataset_device_masks_feed_a_train_step():
    """SyntheticDataset(device_masks=True): fresh masks every batch, consumed by the PM-VAE train step."""
    from posterior_matching_amd import optim
    from posterior_matching_amd.data import SyntheticDataset
    from posterior_matching_amd.engine import PMVAETrainStep
    from posterior_matching_amd.models import PosteriorMatchingVAE
    from tests.ref_configs import pm_vae_mnist

    cfg = pm_vae_mnist()
    ds = SyntheticDataset({"dataset": "mnist", "mask_generator": "MNISTMaskGenerator"}, 16, 2, 0, dev(), device_masks=True)
    it = iter(ds)
    masks = [next(it)["mask"].clone() for _ in range(3)]
    assert not torch.equal(masks[0], masks[2])          # batch 0 of the pool comes back with a new mask
    model = PosteriorMatchingVAE.from_config(cfg["model"], device="cuda:0", seed=3)
    model.init((28, 28, 1))
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(model, cfg, opt, 16, (28, 28, 1), use_graph=False)
    for _ in range(3):
        batch = next(it)
        ts.set_batch(batch["image"], batch["mask"])
        ts.step()
    assert np.isfinite(ts.read_metrics()["loss"])


def test_device_masks_match_golden_fixture():
    """the committed bit streams (tests/golden/masks_tiny.npz) straight from the kernels, without the oracle in the loop"""
    import os

    from posterior_matching_amd import ops
    from posterior_matching_amd.masking import get_mask_generator
    from tests.golden.make_golden_masks import CASES

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "masks_tiny.npz"))
    for key, (name, B, H, seed) in CASES.items():
        gen = get_mask_generator(name, device=dev(), seed=seed)
        for step in range(6):
            desc = torch.zeros((B, 6), dtype=torch.int32, device=dev())
            out = torch.empty((B, H, H, 1), device=dev())
            gen.fill(out, desc_out=desc)
            gen._advance()
            if step in (0, 5):
                assert np.array_equal(np.packbits(out.cpu().numpy().astype(np.uint8).reshape(-1)), gold[f"{key}_step{step}_mask"])
                assert np.array_equal(desc.cpu().numpy(), gold[f"{key}_step{step}_desc"])
    step2 = torch.full((1,), 2, dtype=torch.int32, device=dev())
    m = torch.empty((9, 43), device=dev())
    ops.bernoulli_mask(m, 0.3, 3, step2)
    assert np.array_equal(np.packbits(m.cpu().numpy().astype(np.uint8).reshape(-1)), gold["bernoulli_p03"])
    step1 = torch.full((1,), 1, dtype=torch.int32, device=dev())
    u = torch.empty((16, 21), device=dev())
    ops.uniform_mask(u, 0, 21, 4, step1)
    assert np.array_equal(np.packbits(u.cpu().numpy().astype(np.uint8).reshape(-1)), gold["uniform_d21"])
    u2 = torch.empty((8, 300), device=dev())
    ops.uniform_mask(u2, 75, 150, 5, None)
    assert np.array_equal(np.packbits(u2.cpu().numpy().astype(np.uint8).reshape(-1)), gold["uniform_bounds"])


def _pattern_comps(max_size, low, weight_pattern=3.0):
    """a pattern-heavy mixture for the bit-exactness test (CelebA itself draws the pattern for 5 % of the examples)"""
    from posterior_matching_amd._lib import MaskComponent

    oc = [MO.Component(MO.PATTERN, weight_pattern, p=0.25, size=max_size, low_size=low, min_prop=0.05),
          MO.Component(MO.RECT, 1.0, min_prop=0.3, max_prop=1.0)]
    cum = MO.cumulative_weights(oc)
    dc = (MaskComponent * 2)()
    dc[0].kind, dc[0].p, dc[0].size, dc[0].y1, dc[0].min_prop, dc[0].cum_weight = 4, 0.25, max_size, low, 0.05, float(cum[0])
    dc[1].kind, dc[1].min_prop, dc[1].max_prop, dc[1].cum_weight = 3, 0.3, 1.0, float(cum[1])
    return oc, dc


@pytest.mark.parametrize("max_size,low,H,W", [(2000, 120, 64, 64), (10000, 600, 64, 64), (500, 40, 32, 48)])
def test_random_pattern_masks_bit_exact(max_size, low, H, W):
    """PM_MASK_PATTERN (RandomPatternMaskGenerator, reference masking.py:177-232): window origins, accepted tries and every
    pixel equal the oracle's, whose bicubic arithmetic is pinned against Pillow (tests/test_oracle_kat.py); blob coverage
    inside density +- 0.05; the device state counts the pixels handed out and bumps the noise epoch past the threshold."""
    from posterior_matching_amd import ops

    B, seed = 48, 21
    oc, dc = _pattern_comps(max_size, low)
    state = torch.zeros(2, dtype=torch.int64, device=dev())
    step = torch.tensor([5], dtype=torch.int32, device=dev())
    out, desc = torch.empty((B, H, W, 1), device=dev()), torch.zeros((B, 6), dtype=torch.int32, device=dev())
    ops.image_mask_mixture(out, dc, seed, step, 2, desc, state, 10 ** 12)
    torch.cuda.synchronize()
    want, wdesc = MO.image_mask_mixture(B, H, W, oc, seed, step=5, stream=2, pattern_epoch=0)
    assert np.array_equal(desc.cpu().numpy(), wdesc)
    assert np.array_equal(out.cpu().numpy(), want)
    pat = wdesc[:, 0] == MO.PATTERN
    assert pat.sum() >= B // 2
    cover = 1.0 - want[pat].mean(axis=(1, 2, 3))
    assert (np.abs(cover - 0.25) < 0.05 + 1e-6).all()
    assert state.cpu().tolist() == [0, int(pat.sum()) * H * W]
    # a threshold below the pixels handed out: the next launch reads epoch 1 - a different noise field
    ops.image_mask_mixture(out, dc, seed, step, 2, desc, state, 10)
    torch.cuda.synchronize()
    assert state.cpu().tolist() == [1, 0]
    ops.image_mask_mixture(out, dc, seed, step, 2, desc, state, 10 ** 12)
    torch.cuda.synchronize()
    want1, wdesc1 = MO.image_mask_mixture(B, H, W, oc, seed, step=5, stream=2, pattern_epoch=1)
    assert np.array_equal(out.cpu().numpy(), want1) and np.array_equal(desc.cpu().numpy(), wdesc1)
    assert not np.array_equal(want1[pat], want[pat])


def test_celeba_device_mixture_bit_exact_and_distribution():
    """get_mask_generator("CelebAMaskGenerator", device=...) (reference masking.py:289-325, 14 flattened components): two
    batches bit-exact against the oracle, then 4096 masks: component frequencies = the product weights, the fixed GCF /
    SIIDGM rectangles exact, rectangles >= 30 % of the area, pixel-Bernoulli mean 0.2, pattern coverage 0.25 +- 0.05."""
    from posterior_matching_amd.masking import get_mask_generator

    comps = MO.celeba_components()
    gen = get_mask_generator("CelebAMaskGenerator", device=dev(), seed=9)
    for step in range(2):
        got = gen((40, 64, 64, 1))
        assert np.array_equal(got.cpu().numpy(), MO.image_mask_mixture(40, 64, 64, comps, 9, step=step)[0]), step
    B = 4096
    gen = get_mask_generator("CelebAMaskGenerator", device=dev(), seed=10)
    desc = torch.zeros((B, 6), dtype=torch.int32, device=dev())
    out = torch.empty((B, 64, 64, 1), device=dev())
    gen.fill(out, desc_out=desc)
    d, m = desc.cpu().numpy(), out.cpu().numpy()[..., 0]
    assert set(np.unique(m)) <= {0.0, 1.0}
    w = np.array([c.weight for c in comps])
    freq = np.bincount(d[:, 5], minlength=len(comps)) / B
    assert np.abs(freq - w / w.sum()).max() < 0.02
    pat, bern, rect = d[:, 5] == 0, d[:, 5] == 1, d[:, 5] == len(comps) - 1
    assert (np.abs((1 - m[pat]).mean((1, 2)) - 0.25) < 0.05 + 1e-6).all() and pat.sum() > 100
    assert abs(m[bern].mean() - 0.2) < 0.01
    for ci, r in list(enumerate(MO.SIIDGM_RECTS, start=2)) + list(enumerate(MO.GCF_RECTS, start=7)):
        want = np.ones((64, 64))
        want[r[0]:r[2], r[1]:r[3]] = 0
        sel = m[d[:, 5] == ci]
        assert len(sel) > 0 and (sel == want).all(), ci
    area = (d[rect, 3] - d[rect, 1]) * (d[rect, 4] - d[rect, 2])
    assert (area == (1 - m[rect]).sum((1, 2))).all() and area.min() >= 0.3 * 4096
    assert gen.pattern_state.cpu().tolist() == [0, int(pat.sum()) * 4096]
