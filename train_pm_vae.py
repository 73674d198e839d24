#!/usr/bin/env python3
"""Train a Posterior-Matching VAE on the MI355X-native path.

Same entry point as the reference's train_pm_vae.py:

    python train_pm_vae.py --config configs/pm_vae_mnist.py [--config.steps=2000 ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_pm_vae.py --config ...

Differences forced by the environment (no network, no tfds / TensorBoard): data are synthetic
batches of the dataset's shape (or a .npy given with --data), scalars go to runs/<id>/tb/scalars.jsonl.
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
from posterior_matching_amd.data import make_dataset  # noqa: E402
from posterior_matching_amd.models.vae import PosteriorMatchingVAE  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import (CheckpointCallback, LearningRateLoggerCallback, PMVAELoss,  # noqa: E402
                                            Trainer)
from posterior_matching_amd.utils import TensorBoardCallback, configure_environment, make_run_dir  # noqa: E402

configure_environment()


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
    config.lock()

    rank, local_rank, world = env_world()
    import numpy as np
    import torch

    device = torch.device("cuda", local_rank)
    arrays = np.load(args.data) if args.data else None
    train_dataset = make_dataset(config.data, config.data.train_batch_size, 64, config.seed + rank, device,
                                     training=True, arrays=arrays, device_masks=args.device_masks)
    val_dataset = make_dataset(config.data, config.data.val_batch_size, 8, config.seed + 10007 + rank, device,
                                   training=False, arrays=arrays)
    data_key = train_dataset.key

    model = PosteriorMatchingVAE.from_config(config.model, device=device, seed=config.seed)
    loss_fn = PMVAELoss(config, model, data_key)          # loss_fn of the reference's train_pm_vae.py:58-72

    schedule = optim.exponential_decay(**config.lr_schedule)
    optimizer = optim.chain(
        optim.scale_by_adam(**config.get("adam", {})),
        optim.add_decayed_weights(config.get("weight_decay", 0.0), mask="ndim != 1"),
        optim.scale_by_schedule(schedule),
        optim.scale(-1.0),
    )

    trainer = Trainer(loss_fn, optimizer, num_devices=world, seed=config.seed)

    run_dir = make_run_dir(prefix=f"pm-vae-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)
    callbacks = [
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        LearningRateLoggerCallback(schedule),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    train_state = trainer.fit(train_dataset, config.steps, val_dataset=val_dataset,
                              validation_freq=config.validation_freq, callbacks=callbacks)
    if rank == 0:
        if config.get("save_final_state", False):
            with open(os.path.join(run_dir, "train_state.pkl"), "wb") as fp:
                pickle.dump(train_state, fp)
        with open(os.path.join(run_dir, "model_config.json"), "w") as fp:
            json.dump(config.model.to_dict(), fp)


if __name__ == "__main__":
    main()
