#!/usr/bin/env python3
"""Train a VQ-VAE (stage 1 of Posterior Matching for VQ-VAE) on the MI355X-native path.

Same entry point as the reference's train_vqvae.py:

    python train_vqvae.py --config configs/vqvae_mnist.py [--config.steps=2000 ...]

Differences forced by the environment (no network, no tfds / TensorBoard): data are synthetic
batches of the dataset's shape (or a .npy given with --data), scalars and reconstructions go to
runs/<id>/tb/.
"""
import argparse
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.config_dict import apply_overrides, load_config_file  # noqa: E402
from posterior_matching_amd.data import make_dataset  # noqa: E402
from posterior_matching_amd.models.vqvae import VQVAE  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import CheckpointCallback, Trainer, VQVAELoss  # noqa: E402
from posterior_matching_amd.utils import Callback, TensorBoardCallback, configure_environment, make_run_dir  # noqa: E402

configure_environment()


class ReconstructionCallback(Callback):
    """reference train_vqvae.py:32-56: reconstructs three validation images after every validation
    pass (is_training=False, clipped to [0, 1]) and logs them side by side with the inputs."""

    def __init__(self, model: VQVAE, dataset):
        self._model = model
        self._batches = dataset.batches
        self._i = 0

    def on_validation_end(self, train_state, step, logs):
        import torch

        batch = self._batches[self._i % len(self._batches)]["image"][:3].contiguous()
        self._i += 1
        rec = self._model(batch, is_training=False)["reconstruction"].clamp(0.0, 1.0)
        torch.cuda.synchronize()
        logs["reconstructions"] = torch.cat([batch, rec], dim=2).cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", default=None, help="optional .npy with the training examples; a uint8 [N,H,W,C] image array stays "
                                                 "resident in HBM as uint8 and is sampled / converted on the device")
    args, rest = ap.parse_known_args()
    config = load_config_file(args.config)
    apply_overrides(config, [r[len("--config."):] for r in rest if r.startswith("--config.")])
    if "seed" not in config:
        config.seed = random.randint(0, int(2e9))
    config.lock()

    rank, local_rank, world = env_world()
    import numpy as np
    import torch

    device = torch.device("cuda", local_rank)
    arrays = np.load(args.data) if args.data else None
    train_dataset = make_dataset(config.data, config.data.train_batch_size, 64, config.seed + rank, device,
                                     training=True, arrays=arrays)
    val_dataset = make_dataset(config.data, config.data.val_batch_size, 8, config.seed + 10007 + rank, device,
                                   training=False, arrays=arrays)

    model = VQVAE(**config.model, device=device, seed=config.seed)
    loss_fn = VQVAELoss(config, model)                     # loss_fn of the reference's train_vqvae.py:67-75
    optimizer = optim.adam(config.learning_rate)

    trainer = Trainer(loss_fn, optimizer, num_devices=world, seed=config.seed)

    run_dir = make_run_dir(prefix=f"vqvae-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)
        with open(os.path.join(run_dir, "model_config.json"), "w") as fp:
            json.dump(config.model.to_dict(), fp)

    callbacks = [
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        ReconstructionCallback(model, val_dataset),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    trainer.fit(train_dataset, config.steps, val_dataset=val_dataset, validation_freq=config.validation_freq,
                callbacks=callbacks)


if __name__ == "__main__":
    main()
