"""CPU-side checks: configs, ConfigDict, masks, C-ABI symbol export, golden fixture vs oracle,
optimizer-chain lowering, and the data-parallel glue under gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_configs_match_reference_values():
    from posterior_matching_amd.config_dict import load_config_file
    from tests.ref_configs import (lookahead_mnist16, pm_vae_bsds, pm_vae_hepmass, pm_vae_mnist16, pm_vae_power, pm_vade_mnist, pm_vae_gas, pm_vae_miniboone, pm_vae_mnist, pm_vdvae_mnist,
                                   pm_vqvae_celeb_a, pm_vqvae_mnist, vade_mnist, vqvae_celeb_a, vqvae_mnist)

    for name, ref in (("pm_vae_mnist", pm_vae_mnist()), ("pm_vae_gas", pm_vae_gas()), ("vqvae_mnist", vqvae_mnist()),
                      ("pm_vqvae_mnist", pm_vqvae_mnist()), ("pm_vdvae_mnist", pm_vdvae_mnist()),
                      ("vqvae_celeb_a", vqvae_celeb_a()), ("pm_vqvae_celeb_a", pm_vqvae_celeb_a()),
                      ("pm_vae_miniboone", pm_vae_miniboone()), ("vade_mnist", vade_mnist()), ("pm_vade_mnist", pm_vade_mnist()),
                      ("pm_vae_mnist16", pm_vae_mnist16()), ("lookahead_mnist16", lookahead_mnist16()),
                      ("pm_vae_power", pm_vae_power()), ("pm_vae_hepmass", pm_vae_hepmass()), ("pm_vae_bsds", pm_vae_bsds())):
        cfg = load_config_file(os.path.join(ROOT, "configs", name + ".py")).to_dict()
        assert cfg == ref, name


def test_config_dict_semantics():
    from posterior_matching_amd.config_dict import ConfigDict, apply_overrides

    c = ConfigDict()
    c.a = ConfigDict()
    c.a.b = 1
    c.x = {"y": 2}
    assert c.a.b == 1 and c["x"]["y"] == 2 and "a" in c and c.get("zz", 5) == 5
    assert dict(**c.a) == {"b": 1}
    apply_overrides(c, ["a.b=7", "x.y=[1, 2]"])
    assert c.a.b == 7 and c.x.y == [1, 2]
    c.lock()
    c.a.b = 9                                   # existing keys stay writable (ml_collections semantics)
    with pytest.raises(KeyError):
        c.new_key = 1
    with pytest.raises(KeyError):
        c.a.other = 1


def test_mask_generators():
    from posterior_matching_amd.masking import get_mask_generator

    m = get_mask_generator("MNISTMaskGenerator", seed=0)((400, 28, 28, 1))
    assert m.shape == (400, 28, 28, 1) and m.dtype == np.float32 and set(np.unique(m)) <= {0.0, 1.0}
    frac = m.mean(axis=(1, 2, 3))
    assert (np.isclose(frac, 0.5, atol=1e-6).sum() >= 100)       # the four half-image masks + some others
    assert frac.min() >= 0.0 and frac.max() <= 1.0
    g = get_mask_generator("BernoulliMaskGenerator", seed=1)((1000, 8))
    assert g.shape == (1000, 8) and abs(g.mean() - 0.5) < 0.03
    with pytest.raises(KeyError):
        get_mask_generator("NoSuchMaskGenerator")
    with pytest.raises(AssertionError):
        get_mask_generator("MNISTMaskGenerator")((3, 784))


def test_celeba_mask_mixture_and_bicubic_window_matches_pillow():
    """CelebAMaskGenerator (reference masking.py:177-232,289-325): the random-pattern component interpolates only the
    requested window; that window must be bit-identical to a crop of Pillow's full BICUBIC resize (the third-party
    routine the reference calls at masking.py:196-198) - the one external pin this generator has."""
    from PIL import Image

    from posterior_matching_amd.masking import (RandomPatternMaskGenerator, SIIDGMMaskGenerator, bicubic_window,
                                                get_mask_generator)

    rng = np.random.default_rng(0)
    low = rng.uniform(size=(60, 60)).astype(np.float32)
    full = np.array(Image.fromarray(low).resize((1000, 1000), Image.BICUBIC))
    for y0, x0, h, w in [(0, 0, 64, 64), (936, 936, 64, 64), (500, 3, 64, 17), (7, 990, 30, 10), (123, 456, 1, 1)]:
        assert np.array_equal(bicubic_window(low, y0, x0, h, w, 1000), full[y0:y0 + h, x0:x0 + w]), (y0, x0)
    gen = RandomPatternMaskGenerator(max_size=2000, resolution=0.06, seed=3)
    m = gen((50, 64, 64, 1))
    cover = 1.0 - m.mean(axis=(1, 2, 3))
    assert m.shape == (50, 64, 64, 1) and set(np.unique(m)) <= {0.0, 1.0} and (np.abs(cover - 0.25) < 0.05).all()
    assert gen.points_used == 50 * 64 * 64
    gen.update_freq, gen.points_used = 1e-2, 0                # 40 000 points: the cache is redrawn by the 10th mask
    before = gen.low_pattern.copy()
    gen((10, 64, 64, 1))
    assert not np.array_equal(before, gen.low_pattern) and gen.points_used == 0
    s = SIIDGMMaskGenerator(seed=1, max_size=2000)((300, 64, 64, 1))
    assert s.shape == (300, 64, 64, 1) and 0.3 < s.mean() < 0.8
    c = get_mask_generator("CelebAMaskGenerator", seed=0)((600, 64, 64, 1))
    assert c.shape == (600, 64, 64, 1) and c.dtype == np.float32 and set(np.unique(c)) <= {0.0, 1.0}
    frac = c.mean(axis=(1, 2, 3))
    # half of the examples are area-bounded rectangles (>= 30 % missing); the six GCF boxes leave 0.80-0.90 observed
    assert (frac <= 0.7 + 1e-6).mean() > 0.45 and frac.min() >= 0.0 and ((frac > 0.8) & (frac < 0.91)).mean() > 0.1
    o = get_mask_generator("OmniglotMaskGenerator", seed=2)((64, 28, 28, 1))
    assert o.shape == (64, 28, 28, 1)


def test_reference_import_paths_resolve():
    """every import path INTEGRATION.md shows goes through the `posterior_matching` shim package (reference
    posterior_matching/models/*.py, masking.py, utils.py); building a model needs the GPU, importing must not."""
    import importlib

    want = {
        "posterior_matching.models.vae": ["PosteriorMatchingVAE"],
        "posterior_matching.models.networks": ["get_network", "ConvEncoder", "ConvDecoder", "ResidualMLP"],
        "posterior_matching.models.distributions": ["get_distribution", "TriLGaussian", "AutoregressiveGMM", "Bernoulli",
                                                    "IdentityGaussian", "DiagonalGaussian"],
        "posterior_matching.models.vqvae": ["VQVAE", "VQVAEPartialEncoder", "vqvae_impute", "ConvResidualEncoder",
                                            "ConvResidualDecoder"],
        "posterior_matching.models.pixel_cnn": ["PixelCNN"],
        "posterior_matching.models.vdvae": ["PosteriorMatchingVDVAE", "Encoder", "Block", "PosteriorMatchingDecoderBlock",
                                            "parse_layer_string"],
        "posterior_matching.models.vade": ["VADE", "PosteriorMatchingVADE"],
        "posterior_matching.models.lookahead": ["LookaheadPosterior", "LookaheadBlock"],
        "posterior_matching.acquisition": ["rmse", "make_acquisition_eval_fn", "make_collect_trajectory_fn"],
        "posterior_matching.masking": ["get_mask_generator", "MNISTMaskGenerator", "CelebAMaskGenerator",
                                       "BernoulliMaskGenerator", "UniformMaskGenerator"],
        "posterior_matching.utils": ["load_datasets", "cyclical_annealing_schedule", "make_run_dir",
                                     "configure_environment", "TensorBoardCallback"],
    }
    for mod, names in want.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for mod in set(re.findall(r"from (posterior_matching(?:_amd)?[\w.]*) import", text)):
        importlib.import_module(mod)


def test_checkpoint_importer_renames_haiku_trees():
    """posterior_matching_amd.checkpoint (the half that needs no jax): haiku-style module paths of the
    reference (auto-numbered conv2_d / conv2_d_transpose / linear, explicit enc_i / res3x3_i / dec_i, the stage-2 "vqvae/"
    prefix, the "x_bias_{res}]" typo) map onto this package's parameter names with identical shapes; wrong shapes and
    missing leaves are refused."""
    from oracle import pm_vae_oracle as O
    from oracle import vdvae_oracle as DO
    from oracle import vqvae_oracle as VO
    from posterior_matching_amd.checkpoint import haiku_to_native, vq_state_to_native
    from tests.ref_configs import pm_vae_gas, pm_vae_mnist, pm_vdvae_mnist, vqvae_mnist

    rng = np.random.default_rng(0)

    from tests.haiku_names import to_tree, vae_names, vdvae_names, vq_names

    def fake(shapes, to_haiku):
        native = {n: rng.normal(size=s).astype(np.float32) for n, s in shapes.items()}
        return native, to_tree(native, to_haiku)

    for cfg, xs in ((pm_vae_mnist(), (28, 28, 1)), (pm_vae_gas(), (8,))):
        shapes = O.param_shapes(cfg["model"], xs)
        native, tree = fake(shapes, vae_names)
        assert "encoder_net/conv2_d_1" in tree or "encoder_net/linear_1" in tree
        got = haiku_to_native(tree, shapes)
        assert list(got) == list(shapes) and all(np.array_equal(got[n], native[n]) for n in shapes)

    shapes = VO.param_shapes(vqvae_mnist()["model"], 1)
    native, tree = fake(shapes, vq_names)
    got = haiku_to_native(tree, shapes)
    assert all(np.array_equal(got[n], native[n]) for n in shapes)

    shapes = DO.param_shapes(pm_vdvae_mnist()["model"])
    native, tree = fake(shapes, vdvae_names)
    got = haiku_to_native(tree, shapes)
    assert len(got) == len(shapes) and np.array_equal(got["decoder/x_bias_28"], native["decoder/x_bias_28"])

    bad = dict(tree)
    first = next(iter(bad))
    bad[first] = {k: v[..., :-1] for k, v in bad[first].items()}
    with pytest.raises((KeyError, ValueError)):
        haiku_to_native(bad, shapes)
    st = {"vqvae/~/vector_quantizer_ema": {"embeddings": np.zeros((64, 256), np.float32)},
          "vqvae/~/vector_quantizer_ema/~/ema_cluster_size": {"hidden": np.zeros(256), "average": np.zeros(256), "counter": np.array(7)},
          "vqvae/~/vector_quantizer_ema/~/ema_dw": {"hidden": np.zeros((64, 256)), "average": np.zeros((64, 256)), "counter": np.array(7)}}
    vs = vq_state_to_native(st)
    assert vs["counter"].tolist() == [7] and vs["ema_dw/hidden"].shape == (64, 256)


def test_library_exports_every_declared_symbol():
    """the C-ABI library loads (no GPU needed) and exports every function include/pmhip.h declares"""
    from posterior_matching_amd import _lib

    header = open(os.path.join(ROOT, "include", "pmhip.h")).read()
    declared = set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", header))
    declared -= {"pm_stream_t"}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    bound = set(_lib.SIGNATURES) | set(_lib._OTHER_RESTYPE)
    assert declared <= bound, sorted(declared - bound)
    assert _lib.load().pm_version() >= 1
    assert _lib.load().pm_strerror(-1).decode().startswith("invalid")


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from posterior_matching_amd.models import PosteriorMatchingVAE
    from tests.ref_configs import pm_vae_mnist

    m = PosteriorMatchingVAE.from_config(pm_vae_mnist()["model"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.init((28, 28, 1))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "posterior_matching_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# ", ""), (dirpath, f)


def test_golden_fixture_matches_oracle():
    from oracle import pm_vae_oracle as O
    from tests.golden.make_golden import CFG

    z = np.load(os.path.join(ROOT, "tests", "golden", "pm_vae_tiny.npz"))
    p = {k[len("param/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, _, out = O.pm_vae_loss(leaves, CFG, torch.tensor(z["x"]), torch.tensor(z["b"]), torch.tensor(z["eps"]), 0)
    assert loss.item() == pytest.approx(float(z["loss"]), rel=1e-12)
    for key in ("reconstruction_ll", "kl", "matching_ll"):
        np.testing.assert_allclose(out[key].detach().numpy(), z[key], rtol=1e-11)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    for k, g in zip(leaves, grads):
        np.testing.assert_allclose(g.numpy(), z["grad/" + k], rtol=1e-9, atol=1e-13)


def test_vqvae_golden_fixture_matches_oracle():
    from oracle import vqvae_oracle as VO
    from tests.golden.make_golden_vqvae import CFG

    z = np.load(os.path.join(ROOT, "tests", "golden", "vqvae_tiny.npz"))
    p = {k[len("param/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    st = {k[len("state/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("state/")}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, aux, out, new_state = VO.vqvae_loss(leaves, st, CFG, torch.tensor(z["x"]), True)
    assert loss.item() == pytest.approx(float(z["loss"]), rel=1e-12)
    assert aux["perplexity"].item() == pytest.approx(float(z["perplexity"]), rel=1e-12)
    assert np.array_equal(out["vq_output"]["encoding_indices"].numpy(), z["encoding_indices"])
    for k, g in zip(leaves, torch.autograd.grad(loss, list(leaves.values()))):
        np.testing.assert_allclose(g.numpy(), z["grad/" + k], rtol=1e-9, atol=1e-13)
    for k, v in new_state.items():
        np.testing.assert_allclose(v.numpy(), z["new_state/" + k], rtol=1e-12)


def test_optimizer_chain_lowering():
    from posterior_matching_amd import optim

    sch = optim.exponential_decay(init_value=1e-3, transition_steps=5000, decay_rate=0.9)
    assert sch(5000) == pytest.approx(9e-4)
    ch = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(1e-5), optim.scale_by_schedule(sch),
                     optim.scale(-1.0))
    cfg = ch.adam_cfg(grad_scale=0.5)
    assert (cfg.b1, cfg.lr_init, cfg.grad_scale) == (pytest.approx(0.9), pytest.approx(1e-3), 0.5)
    assert cfg.weight_decay == pytest.approx(1e-5)
    with pytest.raises(NotImplementedError):
        optim.chain(optim.scale_by_adam(), optim.scale(-1.0))


_DP_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from posterior_matching_amd.parallel import init_distributed, shard_rows, allreduce_sum_, allreduce_mean_scalars
from oracle import pm_vae_oracle as O
from tests.golden.make_golden import CFG, XS
import numpy as np

rank, local_rank, world = init_distributed("gloo")
assert world == 2 and dist.get_backend() == "gloo"
rng = np.random.default_rng(7)                         # identical global batch on every rank
G = 8
x = torch.tensor(rng.uniform(size=(G,) + XS)); b = torch.tensor((rng.uniform(size=(G,) + XS) < 0.5) * 1.0)
eps = torch.tensor(rng.normal(size=(G, CFG["model"]["latent_dim"])))
p = O.init_params(CFG["model"], XS, seed=3)
def grads(rows):
    leaves = {{k: v.clone().requires_grad_(True) for k, v in p.items()}}
    loss, aux, _ = O.pm_vae_loss(leaves, CFG, x[rows], b[rows], eps[rows], 0)
    g = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), torch.cat([t.reshape(-1) for t in g])
rows = shard_rows(G, rank, world)
assert (rows.start, rows.stop) == (rank * 4, rank * 4 + 4)
loss_r, flat = grads(rows)
allreduce_sum_(flat)                                    # what PMVAETrainStep._allreduce does
flat /= world                                           # pm_adam_step's grad_scale = 1 / world
loss_full, flat_full = grads(slice(0, G))
assert torch.allclose(flat, flat_full, rtol=1e-10, atol=1e-14), (flat - flat_full).abs().max()
m = allreduce_mean_scalars(torch.stack([loss_r]))
assert torch.allclose(m[0], loss_full, rtol=1e-12)

# bucketed reduction (parallel.GradReducer, the N > 1 path of every train step): modules report in reverse order, weight
# ranges are reduced as buckets fill, the 1-D suffix and anything unreported at finish(); each element exactly once
from posterior_matching_amd.parallel import GradReducer
class FakeStore:
    device = torch.device("cpu")
    def __init__(self):
        specs = [("enc/l0/w", 600), ("enc/l1/w", 500), ("dec/l0/w", 700), ("dec/l1/w", 300), ("extra/w", 50),
                 ("enc/l0/b", 6), ("enc/l1/b", 5), ("dec/l0/b", 7), ("dec/l1/b", 3)]
        self.offsets, off = {{}}, 0
        for n, c in specs:
            self.offsets[n] = (off, c); off += c
        self.n_decay = 600 + 500 + 700 + 300 + 50
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_g = torch.randn(off, generator=g, dtype=torch.float64)
both = [torch.randn(2171, generator=torch.Generator().manual_seed(100 + r), dtype=torch.float64) for r in range(world)]
for overlap in (True, False):
    st = FakeStore()
    red = GradReducer(st, bucket_bytes=4 * 900, overlap=overlap)
    for step in range(2):                                 # two steps: the bookkeeping resets
        st.flat_g.copy_(both[rank])
        red.ready(["dec"])                                # 1000 elements >= bucket -> issued at once
        red.ready(["enc/l1"])                             # 500: pending
        red.ready(["enc/l0"])                             # adjacent -> merged 1100 -> issued as ONE range
        red.finish()
        assert torch.allclose(st.flat_g, both[0] + both[1], rtol=0, atol=0), (overlap, step)
        assert red.calls_last_step == (3 if overlap else 1), red.calls_last_step
    try:
        red.ready(["enc/l0", "dec"])                      # not contiguous: refused
        assert not overlap
    except ValueError:
        assert overlap
    red.finish()
# fragments (ops.WgradBatch reports one layer shape of many Blocks at a time): pieces below min_issue_bytes wait for their
# neighbours; a range reported twice is refused; finish() reduces the rest in ONE call
st = FakeStore()
red = GradReducer(st, bucket_bytes=4 * 900, overlap=True)
for step in range(2):
    st.flat_g.copy_(both[rank])
    red.ready_ranges([(0, 100), (600, 700)])
    red.ready_ranges([(1100, 1300), (1800, 1900)])
    assert red._calls == 0
    red.ready_ranges([(100, 600), (700, 1100)])           # 1400 pending: (0, 1300) goes, the 100-element piece stays
    assert red._calls == 1 and red._pending == [(1800, 1900)], (red._calls, red._pending)
    try:
        red.ready_ranges([(50, 60)])
        raise SystemExit("a range was accepted twice")
    except ValueError:
        pass
    red.finish()
    assert torch.allclose(st.flat_g, both[0] + both[1], rtol=0, atol=0), step
    assert red.calls_last_step == 2 and red.calls_before_finish_last_step == 1, (red.calls_last_step, red.calls_before_finish_last_step)
# replicas start identical (ADVICE r3): rank-local initial parameters (train_vade.py fits its GMM per rank) are replaced by
# rank 0's before the first step - Trainer._broadcast_initial_state on a model whose stores hold rank-dependent values
from posterior_matching_amd.trainer import Trainer
class _St:
    def __init__(self, n, seed):
        self.flat_p = torch.randn(n, generator=torch.Generator().manual_seed(seed + 17 * rank))
        self.splits = 0
    def split_all(self):
        self.splits += 1
class _Vq:
    def __init__(self):
        self.state = {{"embeddings": torch.full((4, 3), float(rank)), "counter": torch.tensor([rank], dtype=torch.int32)}}
class _Model:
    def __init__(self):
        self.store, self.partial_store, self.vq = _St(100, 1), _St(40, 2), _Vq()
class _Lf:
    partial_encoder = None
    pixel_cnn = None
tr = Trainer.__new__(Trainer)
tr.world, tr.rank = world, rank
mdl = _Model()
tr._broadcast_initial_state(mdl, _Lf())
for t in (mdl.store.flat_p, mdl.partial_store.flat_p, mdl.vq.state["embeddings"].reshape(-1), mdl.vq.state["counter"].float()):
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    assert all(torch.equal(gathered[0], g) for g in gathered), "replicas differ after the broadcast"
assert torch.equal(mdl.store.flat_p, torch.randn(100, generator=torch.Generator().manual_seed(1)))      # rank 0's values
assert mdl.store.splits == 1 and mdl.partial_store.splits == 1                                          # bf16 copies refreshed
dist.barrier()
if rank == 0:
    print("DP-OK")
dist.destroy_process_group()
"""


def test_data_parallel_glue_gloo_world2(tmp_path):
    """2 ranks, gloo: shard rows -> per-rank grads -> all-reduce sum / N == full-batch gradient."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DP-OK" in out.stdout


def test_ctypes_signatures_match_the_header():
    """every prototype of include/pmhip.h against the argtypes table of posterior_matching_amd/_lib.py: same number of
    parameters, pointer / int / long long / float class by class (a missing pointer entry makes ctypes pass the next
    device pointer as a 32-bit int: a GPU memory fault instead of an error)."""
    import ctypes as C
    import re

    from posterior_matching_amd import _lib

    src = open(os.path.join(ROOT, "include", "pmhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = re.findall(r"\b(?:int|void|const char\s*\*)\s+(pm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
    assert len(protos) > 100

    def cls(param: str):
        p = " ".join(param.split())
        if p in ("", "void"):
            return None
        if "*" in p or p.startswith("pm_stream_t"):
            return C.c_void_p
        if "unsigned long long" in p or "long long" in p:
            return C.c_longlong
        if p.startswith("float") or " float " in f" {p} ":
            return C.c_float
        if p.startswith(("int ", "unsigned ", "const int ")):
            return C.c_int
        raise AssertionError(f"unclassified parameter {p!r}")

    table = dict(_lib.SIGNATURES)
    table.update({k: v[0] for k, v in _lib._OTHER_RESTYPE.items()})
    checked = 0
    for name, params in protos:
        want = [c for c in (cls(p) for p in params.split(",")) if c is not None]
        assert name in table, f"{name} is declared in pmhip.h but has no ctypes signature"
        got = []
        for a in table[name]:
            if a in (C.c_void_p, C.c_char_p) or (isinstance(a, type) and issubclass(a, (C._Pointer, C.Array))) or hasattr(a, "_type_") and a not in (C.c_int, C.c_longlong, C.c_float, C.c_ulonglong, C.c_uint):
                got.append(C.c_void_p)
            elif a in (C.c_longlong, C.c_ulonglong):
                got.append(C.c_longlong)
            elif a in (C.c_int, C.c_uint):
                got.append(C.c_int)
            elif a is C.c_float:
                got.append(C.c_float)
            else:
                got.append(C.c_void_p)       # POINTER(struct) and friends
        assert got == want, f"{name}: header {[t.__name__ for t in want]} vs _lib {[t.__name__ for t in got]}"
        checked += 1
    assert checked == len(protos)


def test_planners_and_argument_checks_without_a_gpu():
    """The host side of the C ABI that needs no device: launch planning (which kernel form, how many m-splits / partial-sum
    slots) and the argument validation in front of every launch.  `make -C posterior_matching_amd/csrc asan` runs this file
    against a host-only AddressSanitizer + UBSan build of the same sources (race / memory checking of the
    native code happens on the CPU build; GPU sanitizers are not available on the pool)."""
    import ctypes as C

    from posterior_matching_amd import _lib
    from posterior_matching_amd.ops import LayerGeom

    lib = _lib.load()
    aligned = C.c_void_p(1 << 20)                                  # never dereferenced: planning only looks at alignment
    seen = set()
    for geom, B, mode in [(LayerGeom.conv(28, 28, 32, 32, 5, 2, "SAME"), 256, "wgrad"),
                          (LayerGeom.conv(14, 14, 32, 64, 5, 1, "SAME"), 256, "wgrad"),
                          (LayerGeom.conv_t(28, 28, 32, 32, 5, 1, "SAME"), 256, "dgrad"),
                          (LayerGeom.conv(14, 14, 64, 64, 5, 2, "SAME"), 256, "wgrad"),
                          (LayerGeom.dense(256, 256), 8192, "wgrad"),
                          (LayerGeom.masked_conv(7, 7, 256, 256, 3, 3, 2, 3), 256, "wgrad"),
                          (LayerGeom.conv(28, 28, 48, 48, 3, 1, "SAME"), 16, "wgrad")]:
        d = geom._desc(B, mode)
        n = C.c_int(-1)
        assert lib.pm_wgrad_part_slots(C.byref(d), aligned, aligned, 0, 1, 1, C.byref(n)) == 0, geom
        assert 1 <= n.value <= 1024
        seen.add(n.value)
        v = [C.c_int() for _ in range(5)]
        assert lib.pm_query_wgrad_plan(C.byref(d), 1, 1, *[C.byref(x) for x in v]) == 0
        assert v[4].value >= 1
        bm, bn, md = C.c_int(), C.c_int(), C.c_int()
        assert lib.pm_query_gemm_plan(C.byref(geom._desc(B, "fwd")), 1, C.byref(bm), C.byref(bn), C.byref(md)) == 0
        assert bm.value in (32, 64, 128) and bn.value in (32, 64, 128)
    assert len(seen) > 1                                            # persistent forms (one slot per workgroup) and m-splits
    n = C.c_int()
    assert lib.pm_colsum_part_slots(256 * 784, 32, C.byref(n)) == 0 and n.value >= 1
    assert lib.pm_colsum_part_slots(100, 30, C.byref(n)) == 0 and n.value == 1     # any width (column-per-thread form)
    assert lib.pm_colsum_part_slots(100, 0, C.byref(n)) != 0
    d = LayerGeom.conv(28, 28, 1, 32, 5, 1, "SAME")._desc(256, "wgrad")
    assert lib.pm_thin_wgrad_part_slots(C.byref(d), aligned, aligned, C.byref(n)) == 0 and n.value == 256
    # argument validation returns before anything is launched (no device needed): NULL operands, bad shapes, bad tables
    d = LayerGeom.dense(256, 256)._desc(64, "wgrad")
    part = _lib.WgradPart()
    assert lib.pm_gather_wgrad_part(None, C.byref(d), None, aligned, None, 1, 1, C.byref(part)) != 0
    assert lib.pm_gather_wgrad_part(None, C.byref(d), aligned, aligned, None, 1, 1, None) != 0
    assert lib.pm_reduce_partials(None, None, 4, aligned) != 0
    assert lib.pm_adam_step_jobs(None, aligned, 0, aligned, aligned, aligned, aligned, 0, aligned, None) != 0
    assert lib.pm_colsum_part(None, aligned, 1024, 32, aligned, 8, 3) != 0          # stride < N
    bad = LayerGeom.dense(256, 256)._desc(64, "wgrad")
    bad.C = 0
    assert lib.pm_wgrad_part_slots(C.byref(bad), aligned, aligned, 0, 1, 1, C.byref(n)) != 0
    assert lib.pm_gather_gemm_bf16(None, C.byref(bad), aligned, aligned, None, None, None, aligned) != 0
    # second half of round 4: split-K slab planning, the fixed-order reductions' scratch sizes, the batching entry points
    nf = C.c_longlong(-1)
    d = LayerGeom.dense(2048, 512)._desc(64, "fwd")                 # 8 tiles, 64 k-chunks: splits K
    assert lib.pm_gemm_splitk_floats(C.byref(d), 0, 1, C.byref(nf)) == 0 and nf.value == 16 * 64 * 512
    d = LayerGeom.conv(8, 8, 256, 256, 3, 1, "SAME")._desc(16, "fwd")
    assert lib.pm_gemm_splitk_floats(C.byref(d), 1, 1, C.byref(nf)) == 0 and nf.value % (16 * 64 * 256) == 0 and nf.value > 0
    d = LayerGeom.dense(256, 256)._desc(8192, "fwd")                # fills the chip: no split
    assert lib.pm_gemm_splitk_floats(C.byref(d), 0, 1, C.byref(nf)) == 0 and nf.value == 0
    assert lib.pm_gemm_splitk_floats(C.byref(d), 0, 1, None) != 0
    assert lib.pm_gather_gemm_sk(None, C.byref(d), aligned, aligned, None, None, None, aligned, None, 0) != 0      # no scratch
    assert lib.pm_gather_gemm_bf16_sk(None, C.byref(d), aligned, aligned, None, None, None, aligned, None, 0, None, 0) != 0
    assert lib.pm_vq_dw_exact_floats(12544, 64, 256, C.byref(nf)) == 0 and nf.value == 7 * 64 * 256    # ceil(N / 2048) segments
    assert lib.pm_vq_dw_exact_floats(1000, 64, 256, C.byref(nf)) == 0 and nf.value == 0               # one segment: no scratch
    assert lib.pm_vq_dw_exact_floats(1000, 2048, 256, C.byref(nf)) != 0                               # D <= 1024
    assert lib.pm_vq_dw_exact(None, aligned, aligned, aligned, 12544, 64, 256, None, 0) != 0          # scratch missing
    assert lib.pm_embed_bwd_sorted(None, aligned, aligned, aligned, 4096, 128, 512, aligned, 10) != 0   # scratch too small
    assert lib.pm_normal_ll_bwd_det(None, aligned, aligned, aligned, aligned, aligned, aligned, 8, 4, 0.0, None) != 0
    assert lib.pm_rows_sum_multi(None, None, 3, aligned, 16, 256, 256) != 0
    ptrs = (C.c_void_p * 65)(*[1 << 20] * 65)
    assert lib.pm_rows_sum_multi(None, ptrs, 65, aligned, 16, 256, 256) != 0                            # <= 64 tensors
    jobs = (_lib.ColsumJob * 2)()
    for j in jobs:
        j.x, j.part, j.part_stride, j.M, j.N, j.nslots = 1 << 20, 1 << 20, 32, 256 * 784, 32, 1         # wrong slot count
    assert lib.pm_colsum_part_multi(None, jobs, 2) != 0
    assert lib.pm_colsum_part_multi(None, jobs, 9) != 0 and lib.pm_colsum_part_multi(None, None, 1) != 0
