#!/usr/bin/env python3
"""Posterior Matching for a trained VaDE on the MI355X-native path.

Same entry point as the reference's train_pm_vade.py:

    python train_pm_vade.py --config configs/pm_vade_mnist.py --config.vade_dir=runs/vade-mnist-<id>

`vade_dir` is a run directory written by train_vade.py (model_config.json + train_state.pkl).  The VaDE's parameters are
frozen; the partial encoder and its AutoregressiveGMM are trained on loss = -mean log q(z | x_o), z ~ q(z | x)
(train_pm_vade.py:40-43, 59-60), masks from UniformMaskGenerator (:34).
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
from posterior_matching_amd.models.vade import PosteriorMatchingVADE  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import CheckpointCallback, LearningRateLoggerCallback, PMVADELoss, Trainer  # noqa: E402
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
    config.data.mask_generator = "UniformMaskGenerator"          # train_pm_vade.py:34
    config.lock()

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

    model = PosteriorMatchingVADE.from_config(config.model.to_dict(), device=device, seed=config.seed)
    loss_fn = PMVADELoss(config, model, data_key, seed=config.seed)       # loss_fn of the reference's train_pm_vade.py:40-43

    run_dir = make_run_dir(prefix=f"pm-vade-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)
    with open(os.path.join(config.vade_dir, "train_state.pkl"), "rb") as fp:
        vade_state = pickle.load(fp)

    schedule = optim.exponential_decay(**config.lr_schedule)
    optimizer = optim.chain(optim.scale_by_adam(**config.get("adam", {})), optim.scale_by_schedule(schedule), optim.scale(-1.0))

    def trainable_predicate(module_name, name, value):
        return "partial_" in module_name

    trainer = Trainer(loss_fn, optimizer, num_devices=world, trainable_predicate=trainable_predicate, seed=config.seed)
    callbacks = [
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        LearningRateLoggerCallback(schedule),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    if rank == 0:
        with open(os.path.join(run_dir, "model_config.json"), "w") as fp:
            json.dump(config.model.to_dict(), fp)
    print("Starting main training...")
    trainer.fit(train_dataset, config.steps, val_dataset=val_dataset, validation_freq=config.validation_freq,
                callbacks=callbacks, initial_params=vade_state.params, initial_state=vade_state.state)


if __name__ == "__main__":
    main()
