#!/usr/bin/env python3
"""Lookahead posteriors for a trained PM-VAE on the MI355X-native path.

Same entry point as the reference's train_lookahead_posterior.py:

    python train_lookahead_posterior.py --config configs/lookahead_mnist16.py --config.pm_vae_dir=runs/pm-vae-mnist16-<id>

`pm_vae_dir` is a run directory written by train_pm_vae.py (model_config.json + train_state.pkl).  The PM-VAE's parameters are
frozen; the lookahead encoder is trained on loss = -mean lookahead_lls (train_lookahead_posterior.py:46-52, 61-62).
"""
import argparse
import json
import math
import os
import pickle
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.config_dict import apply_overrides, load_config_file  # noqa: E402
from posterior_matching_amd.data import make_dataset  # noqa: E402
from posterior_matching_amd.models.lookahead import LookaheadPosterior  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import CheckpointCallback, LearningRateLoggerCallback, LookaheadLoss, Trainer  # noqa: E402
from posterior_matching_amd.utils import TensorBoardCallback, configure_environment, make_run_dir  # noqa: E402

configure_environment()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", default=None, help="optional .npy with the training examples")
    ap.add_argument("--device_masks", action="store_true", help="draw a fresh mask per training batch on the GPU")
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
    train_dataset = make_dataset(config.data, config.data.train_batch_size, 64, config.seed + rank, device, training=True,
                                 arrays=arrays, device_masks=args.device_masks)
    val_dataset = make_dataset(config.data, config.data.val_batch_size, 8, config.seed + 10007 + rank, device,
                               training=False, arrays=arrays)
    data_key = train_dataset.key

    with open(os.path.join(config.pm_vae_dir, "model_config.json"), "r") as fp:
        pm_vae_config = json.load(fp)
    with open(os.path.join(config.pm_vae_dir, "train_state.pkl"), "rb") as fp:
        pm_vae_state = pickle.load(fp)

    config.model.num_features = math.prod(train_dataset.batches[0]["mask"].shape[1:])     # :42
    config.lock()

    model = LookaheadPosterior.from_config(config.model.to_dict(), pm_vae_config, device=device, seed=config.seed)
    loss_fn = LookaheadLoss(config, model, data_key, seed=config.seed)      # loss_fn of the reference's script (:46-52)

    schedule = optim.exponential_decay(**config.lr_schedule)
    optimizer = optim.chain(optim.scale_by_adam(**config.get("adam", {})), optim.scale_by_schedule(schedule), optim.scale(-1.0))

    def trainable_predicate(module_name, name, value):
        return "lookahead" in module_name

    trainer = Trainer(loss_fn, optimizer, trainable_predicate=trainable_predicate, num_devices=world, seed=config.seed)

    run_dir = make_run_dir(prefix=f"lookahead-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)
        with open(os.path.join(run_dir, "lookahead_config.json"), "w") as fp:
            json.dump(config.model.to_dict(), fp)
        with open(os.path.join(run_dir, "pm_vae_config.json"), "w") as fp:
            json.dump(pm_vae_config, fp)

    callbacks = [
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        LearningRateLoggerCallback(schedule),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    trainer.fit(train_dataset, config.steps, val_dataset=val_dataset, validation_freq=config.validation_freq,
                callbacks=callbacks, initial_params=pm_vae_state.params, initial_state=pm_vae_state.state)


if __name__ == "__main__":
    main()
