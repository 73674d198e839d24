"""The uint8 image pipeline on the device (reference utils.py:36-58: uint8 images -> shuffle -> batch -> cast [/ 255] -> masks):
index draws and the gather / convert kernel are exact against oracle/masking_oracle.py and numpy; the HBM-resident dataset
feeds the train scripts."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import masking_oracle as MO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("N,shape,B", [(1000, (28, 28, 1), 256), (77, (64, 64, 3), 16), (50, (5, 3, 1), 7), (3, (28, 28, 1), 9)])
def test_random_indices_and_u8_gather_exact(N, shape, B):
    from posterior_matching_amd import ops

    rng = np.random.default_rng(N)
    data = rng.integers(0, 256, size=(N,) + shape, dtype=np.uint8)
    src = torch.from_numpy(data).to(dev())
    idx = torch.zeros(B, dtype=torch.int32, device=dev())
    step = torch.tensor([4], dtype=torch.int32, device=dev())
    ops.random_indices(idx, N, 123, step, stream_id=77)
    want_idx = MO.random_indices(B, N, 123, step=4, stream=77)
    assert np.array_equal(idx.cpu().numpy(), want_idx) and want_idx.min() >= 0 and want_idx.max() < N
    out = torch.empty((B,) + shape, device=dev())
    ops.gather_u8_rows(src, idx, out, 1.0 / 255.0)
    assert np.array_equal(out.cpu().numpy(), data[want_idx].astype(np.float32) * np.float32(1.0 / 255.0))
    if B <= N:
        ops.gather_u8_rows(src[2:2 + B] if B + 2 <= N else src, None, out, 1.0)         # sequential rows, raw 0..255
        assert np.array_equal(out.cpu().numpy(), data[2:2 + B].astype(np.float32) if B + 2 <= N else data[:B].astype(np.float32))
    ops.random_indices(idx, N, 123, torch.tensor([5], dtype=torch.int32, device=dev()), stream_id=77)
    assert not np.array_equal(idx.cpu().numpy(), want_idx) or N < 4


def test_uniformity_of_index_draws():
    from posterior_matching_amd import ops

    idx = torch.zeros(1 << 16, dtype=torch.int32, device=dev())
    ops.random_indices(idx, 60, 7, None, 0)
    cnt = np.bincount(idx.cpu().numpy(), minlength=60)
    assert cnt.min() > 0.85 * (1 << 16) / 60 and cnt.max() < 1.15 * (1 << 16) / 60


def test_device_uint8_dataset_and_train_script(tmp_path):
    """DeviceUint8Dataset: batches are rows of the array scaled by 1/255 (or raw for the VDVAE), masks come from the device
    generator; `train_pm_vae.py --data mnist_u8.npy` and `train_vqvae.py --data celeb_u8.npy` run on it."""
    from posterior_matching_amd.data import DeviceUint8Dataset, make_dataset

    rng = np.random.default_rng(0)
    imgs = (rng.integers(0, 256, size=(300, 28, 28, 1)) * (rng.uniform(size=(300, 28, 28, 1)) < 0.2)).astype(np.uint8)
    cfg = {"dataset": "mnist", "mask_generator": "MNISTMaskGenerator"}
    ds = make_dataset(cfg, 32, 4, 3, dev(), training=True, arrays=imgs)
    assert isinstance(ds, DeviceUint8Dataset)
    it = iter(ds)
    b0 = {k: v.clone() for k, v in next(it).items()}
    want = imgs[MO.random_indices(32, 300, 3, step=0, stream=77)].astype(np.float32) * np.float32(1.0 / 255.0)
    assert np.array_equal(b0["image"].cpu().numpy(), want)
    assert b0["mask"].shape == (32, 28, 28, 1) and set(torch.unique(b0["mask"]).cpu().tolist()) <= {0.0, 1.0}
    b1 = next(it)
    assert not torch.equal(b1["image"], b0["image"])
    raw = make_dataset(cfg, 16, 4, 3, dev(), training=False, arrays=imgs, normalize_images=False)
    vb = raw.next_batch()
    assert np.array_equal(vb["image"].cpu().numpy(), imgs[:16].astype(np.float32))          # validation walks in order
    with pytest.raises(ValueError):
        make_dataset(cfg, 8, 4, 0, dev(), arrays=np.zeros((10, 32, 32, 1), np.uint8))

    np.save(tmp_path / "mnist_u8.npy", imgs)

    def run(script, *argv):
        out = subprocess.run([sys.executable, os.path.join(ROOT, script), *argv], cwd=tmp_path, capture_output=True, text=True,
                             timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]

    run("train_pm_vae.py", "--config", os.path.join(ROOT, "configs", "pm_vae_mnist.py"), "--data", str(tmp_path / "mnist_u8.npy"),
        "--config.steps=12", "--config.validation_freq=6", "--config.seed=1", "--config.data.train_batch_size=64",
        "--config.data.val_batch_size=64")
    runs = os.path.join(tmp_path, "runs")
    rd = os.path.join(runs, [d for d in os.listdir(runs) if d.startswith("pm-vae")][0])
    lines = [json.loads(l) for l in open(os.path.join(rd, "tb", "scalars.jsonl"))]
    assert [l["step"] for l in lines] == [6, 12] and all(np.isfinite(l["train_loss"]) and np.isfinite(l["val_loss"]) for l in lines)
    assert lines[1]["train_loss"] < lines[0]["train_loss"]
    celeb = rng.integers(0, 256, size=(64, 64, 64, 3), dtype=np.uint8)
    np.save(tmp_path / "celeb_u8.npy", celeb)
    run("train_vqvae.py", "--config", os.path.join(ROOT, "configs", "vqvae_celeb_a.py"), "--data", str(tmp_path / "celeb_u8.npy"),
        "--config.steps=6", "--config.validation_freq=6", "--config.seed=1", "--config.data.train_batch_size=16",
        "--config.data.val_batch_size=16")
