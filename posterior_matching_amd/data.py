"""Synthetic input batches for the hot path: there is no network, so tfds
(reference utils.py:36-121) is replaced by seeded synthetic data of the same shapes, dtypes and
value ranges, or by user-supplied .npy arrays."""
from __future__ import annotations

from typing import Dict, Iterator, List, Mapping, Optional

import numpy as np
import torch

from .masking import get_mask_generator


def data_shape(dataset: str):
    if dataset == "mnist16":
        return (16, 16, 1)                  # datasets/mnist16.py: MNIST resized to 16 x 16 (configs/{pm_vae,lookahead}_mnist16.py)
    if "mnist" in dataset:
        return (28, 28, 1)
    if dataset == "celeb_a":
        return (64, 64, 3)                  # utils.py:76-85: centre crop [45:-45, 25:-25] resized to 64 x 64
    return {"gas": (8,), "power": (6,), "hepmass": (21,), "miniboone": (43,), "bsds": (63,)}[dataset]


class SyntheticDataset:
    """A fixed pool of pre-generated batches, cycled.  Batches are dicts like the reference's
    ({"image"|"features": f32[B,...], "mask": f32[B,...]}), already resident on `device`."""

    def __init__(self, config: Mapping, batch_size: int, num_batches: int = 64, seed: int = 0, device="cpu",
                 training: bool = True, arrays: Optional[np.ndarray] = None, normalize_images: bool = True,
                 device_masks: bool = False, labels=None):
        """device_masks: draw a FRESH mask for every yielded batch on the GPU (masking.DeviceMaskGenerator, SURVEY.md
        8(f)-1) instead of cycling the masks generated on the host with the pool."""
        rng = np.random.default_rng(seed)
        name = config["dataset"]
        shape = data_shape(name)
        self.key = "image" if len(shape) == 3 else "features"
        gen = None                                          # stage-1 VQ-VAE batches carry no mask (utils.py:338-350)
        self._device_gen = None
        if config.get("mask_generator") is not None and device_masks:
            self._device_gen = get_mask_generator(config["mask_generator"], device=device, seed=seed + 1,
                                                  **config.get("mask_generator_kwargs", {}))
        elif config.get("mask_generator") is not None:
            gen = get_mask_generator(config["mask_generator"], seed=seed + 1, **config.get("mask_generator_kwargs", {}))
        self.batches: List[Dict[str, torch.Tensor]] = []
        for i in range(num_batches):
            if arrays is not None:
                idx = rng.integers(0, arrays.shape[0], size=batch_size)
                x = arrays[idx].astype(np.float32)
            elif self.key == "image" and name == "celeb_a":
                x = rng.uniform(size=(batch_size,) + shape).astype(np.float32)     # U[0, 1] RGB
            elif self.key == "image":
                # MNIST-like: ~19 % of the pixels carry ink, values in [0, 1] (utils.py:50-54: x / 255)
                x = (rng.uniform(size=(batch_size,) + shape) * (rng.uniform(size=(batch_size,) + shape) < 0.19)).astype(np.float32)
                if not normalize_images:                 # load_datasets(normalize_images=False): raw 0..255 pixel values
                    x = np.round(x * 255.0).astype(np.float32)
            else:
                x = rng.normal(size=(batch_size,) + shape).astype(np.float32)
                if training and "training_noise" in config:          # utils.py:108-116
                    x = x + rng.normal(scale=config["training_noise"], size=x.shape).astype(np.float32)
            batch = {self.key: torch.from_numpy(x).to(device)}
            if labels is not None:          # "label" of the tfds examples (train_vade.py:105-108): an int array aligned with
                # `arrays`, or True for synthetic class labels 0..9
                lab = labels[idx] if (arrays is not None and not isinstance(labels, bool)) else rng.integers(0, 10, size=batch_size)
                batch["label"] = torch.from_numpy(np.asarray(lab, dtype=np.int64))
            if gen is not None:
                mshape = (batch_size,) + shape
                batch["mask"] = torch.from_numpy(np.ascontiguousarray(gen(mshape))).to(device)
            elif self._device_gen is not None:
                mshape = (batch_size,) + (shape[:-1] + (1,) if self.key == "image" else shape)
                batch["mask"] = torch.empty(mshape, dtype=torch.float32, device=device)
            self.batches.append(batch)
        self.batch_size, self.shape = batch_size, shape

    def __len__(self):
        return len(self.batches)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        i = 0
        while True:
            batch = self.batches[i % len(self.batches)]
            if self._device_gen is not None:
                self._device_gen(batch["mask"].shape, out=batch["mask"])
            yield batch
            i += 1


_RESIDENT: Dict[tuple, torch.Tensor] = {}     # host array identity -> its uint8 copy in HBM


class DeviceUint8Dataset:
    """A uint8 image array [N, H, W, C] resident in HBM (MNIST 47 MB, CelebA 64x64 2.4 GB of the 288 GB): every yielded
    batch draws its B row indices from a device Philox stream and gathers + converts the rows in one kernel
    (pm_random_indices, pm_gather_u8_rows) - the reference's tfds uint8 -> shuffle -> batch -> cast [/ 255] pipeline
    (utils.py:36-58) with no host work per step and a quarter of the bytes of float32 storage.  Sampling is with
    replacement (the reference reshuffles a 40 000-element buffer; same marginal distribution).  Batches are dicts like
    the reference's; masks come from a device generator (one launch) or, without one, are absent (stage-1 VQ-VAE).
    CelebA inputs must already be cropped / resized to 64 x 64 (utils.py:76-85 does that with tf.image.resize)."""

    def __init__(self, config: Mapping, images, batch_size: int, seed: int = 0, device="cuda:0", normalize_images: bool = True,
                 training: bool = True):
        from . import ops  # noqa: F401  (fails loudly without the HIP library)

        self.device = torch.device(device)
        if images.dtype not in (np.uint8, torch.uint8):
            raise TypeError(f"DeviceUint8Dataset holds uint8 pixels, got {images.dtype}")
        shape = data_shape(config["dataset"])
        if tuple(images.shape[1:]) != tuple(shape):
            raise ValueError(f"{config['dataset']} examples are {shape}, the array holds {tuple(images.shape[1:])}")
        if images.shape[0] < batch_size:        # pm_gather_u8_rows reads batch_size rows: a shorter array would be read past its end
            raise ValueError(f"the array holds {images.shape[0]} examples, fewer than one batch of {batch_size}")
        if isinstance(images, torch.Tensor):     # already resident (the train and validation datasets of one run share it)
            self.images = images.to(self.device).contiguous()
        else:
            key = (images.__array_interface__["data"][0], tuple(images.shape), str(self.device))
            if key not in _RESIDENT:
                _RESIDENT.clear()                # one resident array per process (CelebA 64x64: 2.4 GB)
                _RESIDENT[key] = torch.from_numpy(np.ascontiguousarray(images)).to(self.device)
            self.images = _RESIDENT[key]
        self.key, self.batch_size, self.shape, self.seed = "image", batch_size, shape, seed
        self.scale = 1.0 / 255.0 if normalize_images else 1.0
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._idx = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
        self._x = torch.empty((batch_size,) + shape, dtype=torch.float32, device=self.device)
        self._mask, self._gen = None, None
        if config.get("mask_generator") is not None:
            self._gen = get_mask_generator(config["mask_generator"], device=self.device, seed=seed + 1,
                                           **config.get("mask_generator_kwargs", {}))
            self._mask = torch.empty((batch_size,) + shape[:-1] + (1,), dtype=torch.float32, device=self.device)
        self.sequential = not training          # validation: walk the array in order instead of sampling
        self._cursor = 0
        self._batches = None

    def next_batch(self) -> Dict[str, torch.Tensor]:
        from . import ops

        if self.sequential:
            n = self.images.shape[0]
            start = self._cursor if self._cursor + self.batch_size <= n else 0
            self._cursor = start + self.batch_size
            ops.gather_u8_rows(self.images[start:start + self.batch_size], None, self._x, self.scale)
        else:
            ops.random_indices(self._idx, self.images.shape[0], self.seed, self._step, stream_id=77)
            ops.gather_u8_rows(self.images, self._idx, self._x, self.scale)
            ops.counter_increment(self._step)
        batch = {self.key: self._x}
        if self._gen is not None:
            self._gen(self._mask.shape, out=self._mask)
            batch["mask"] = self._mask
        return batch

    @property
    def batches(self):
        """a few materialised batches (callbacks that index `.batches`, Trainer._validate)"""
        if self._batches is None:               # materialised once: indexing `.batches` must not advance the cursor again
            n = max(1, min(8, self.images.shape[0] // self.batch_size))
            self._batches = []
            for _ in range(n):
                b = self.next_batch()
                self._batches.append({k: v.clone() for k, v in b.items()})
        return self._batches

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        while True:
            yield self.next_batch()


def make_dataset(config: Mapping, batch_size: int, num_batches: int, seed: int, device, training: bool = True, arrays=None,
                 normalize_images: bool = True, device_masks: bool = False, labels=None):
    """What the train scripts call: a uint8 image .npy stays in HBM as uint8 (DeviceUint8Dataset); anything else (float
    arrays, feature tables, no array at all) goes through SyntheticDataset's pre-generated pool."""
    if arrays is not None and getattr(arrays, "dtype", None) == np.uint8 and arrays.ndim == 4 and labels is None:
        return DeviceUint8Dataset(config, arrays, batch_size, seed, device, normalize_images, training)
    if arrays is not None and getattr(arrays, "dtype", None) == np.uint8:
        arrays = arrays.astype(np.float32) / (255.0 if normalize_images else 1.0)
    return SyntheticDataset(config, batch_size, num_batches, seed, device, training=training, arrays=arrays,
                            normalize_images=normalize_images, device_masks=device_masks, labels=labels)


def load_datasets(config: Mapping, device="cpu", seed: int = 0, num_batches: int = 64):
    """Counterpart of reference utils.py:36-121 for synthetic / .npy data: (train, val)."""
    train = SyntheticDataset(config, config["train_batch_size"], num_batches, seed, device, training=True)
    val = SyntheticDataset(config, config["val_batch_size"], max(1, num_batches // 8), seed + 1000, device, training=False)
    return train, val
