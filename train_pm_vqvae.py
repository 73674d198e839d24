#!/usr/bin/env python3
"""Stage 2 of Posterior Matching for VQ-VAE on the MI355X-native path: trains the partial encoder and
the conditional PixelCNN over the codes of a frozen stage-1 VQ-VAE.

Same entry point as the reference's train_pm_vqvae.py:

    python train_pm_vqvae.py --config configs/pm_vqvae_mnist.py --config.vqvae_dir=runs/vqvae-mnist-<id>

`vqvae_dir` is a run directory written by train_vqvae.py (model_config.json + train_state.pkl).
Data are synthetic batches of the dataset's shape (or a .npy given with --data).
"""
import argparse
import json
import os
import pickle
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.config_dict import apply_overrides, load_config_file  # noqa: E402
from posterior_matching_amd.data import data_shape, make_dataset  # noqa: E402
from posterior_matching_amd.models.vqvae import VQVAE, build_partial_posterior, vqvae_impute  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import CheckpointCallback, PMVQVAELoss, Trainer  # noqa: E402
from posterior_matching_amd.utils import Callback, TensorBoardCallback, configure_environment, make_run_dir  # noqa: E402

configure_environment()


class ImputationCallback(Callback):
    """reference train_pm_vqvae.py:33-60: 5 imputations of 3 validation images after every validation pass,
    logged next to the image and its observed part (unobserved pixels shown as 0.5)."""

    def __init__(self, vqvae, partial_encoder, pixel_cnn, dataset):
        self._mods = (vqvae, partial_encoder, pixel_cnn)
        self._batches, self._i = dataset.batches, 0

    def on_validation_end(self, train_state, step, logs):
        import torch

        batch = self._batches[self._i % len(self._batches)]
        self._i += 1
        x, b = batch["image"][:3].contiguous(), batch["mask"][:3].contiguous()
        imp = vqvae_impute(*self._mods, x, b, num_samples=5, seed=random.randint(0, int(2e9)))
        torch.cuda.synchronize()
        x_o = torch.where(b == 1, x, torch.full_like(x, 0.5))
        tiles = imp.permute(0, 2, 1, 3, 4).reshape(3, x.shape[1], 5 * x.shape[2], x.shape[3])   # b h (s w) c
        logs["imputations"] = torch.cat([x, x_o, tiles], dim=2).cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", default=None, help="optional .npy with the training examples; a uint8 [N,H,W,C] image array stays "
                                                 "resident in HBM as uint8 and is sampled / converted on the device")
    ap.add_argument("--device_masks", action="store_true",
                    help="draw a fresh mask for every training batch on the GPU (pm_image_mask_mixture & co.) "
                         "instead of cycling host-generated masks")
    args, rest = ap.parse_known_args()
    config = load_config_file(args.config)
    apply_overrides(config, [r[len("--config."):] for r in rest if r.startswith("--config.")])
    if "seed" not in config:
        config.seed = random.randint(0, int(2e9))

    rank, local_rank, world = env_world()
    import numpy as np
    import torch

    device = torch.device("cuda", local_rank)
    arrays = np.load(args.data) if args.data else None
    train_dataset = make_dataset(config.data, config.data.train_batch_size, 64, config.seed + rank, device,
                                     training=True, arrays=arrays, device_masks=args.device_masks)
    val_dataset = make_dataset(config.data, config.data.val_batch_size, 8, config.seed + 10007 + rank, device,
                                   training=False, arrays=arrays)

    with open(os.path.join(config.vqvae_dir, "model_config.json"), "r") as fp:
        vqvae_config = json.load(fp)
    with open(os.path.join(config.vqvae_dir, "train_state.pkl"), "rb") as fp:
        vqvae_state = pickle.load(fp)
    config.pixel_cnn.num_indices = vqvae_config["num_embeddings"]
    config.lock()

    x_shape = data_shape(config.data.dataset)
    vqvae = VQVAE(**vqvae_config, device=device, seed=config.seed)
    vqvae.init(x_shape)
    pc_cfg = {k: v for k, v in config.pixel_cnn.to_dict().items() if k != "num_indices"}
    partial_encoder, pixel_cnn, _ = build_partial_posterior(vqvae, config.conditional_dim, pc_cfg, x_shape,
                                                            seed=config.seed)
    loss_fn = PMVQVAELoss(config, vqvae, partial_encoder, pixel_cnn)   # loss_fn of train_pm_vqvae.py:81-99

    schedule = optim.exponential_decay(**config.lr_schedule)
    optimizer = optim.chain(
        optim.scale_by_adam(**config.get("adam", {})),
        optim.scale_by_schedule(schedule),
        optim.scale(-1.0),
    )

    def trainable_predicate(module_name, name, value):
        return not module_name.startswith("vqvae/")

    trainer = Trainer(loss_fn, optimizer, trainable_predicate=trainable_predicate, num_devices=world, seed=config.seed)

    run_dir = make_run_dir(prefix=f"pm-vqvae-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)
        with open(os.path.join(run_dir, "config.json"), "w") as fp:
            json.dump(config.to_dict(), fp)
        with open(os.path.join(run_dir, "vqvae_config.json"), "w") as fp:
            json.dump(vqvae_config, fp)

    callbacks = [
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        ImputationCallback(vqvae, partial_encoder, pixel_cnn, val_dataset),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    trainer.fit(train_dataset, config.steps, val_dataset=val_dataset, validation_freq=config.validation_freq,
                callbacks=callbacks, initial_params=vqvae_state.params, initial_state=vqvae_state.state)


if __name__ == "__main__":
    main()
