#!/usr/bin/env python3
"""Train a VaDE (Variational Deep Embedding) on the MI355X-native path.

Same entry point and the same three phases as the reference's train_vade.py:

    python train_vade.py --config configs/vade_mnist.py [--config.steps=2000 --config.pretrain_steps=500 ...]

  1. pre-training: autoencoder loss -mean decoder(encoder(x).mean()).log_prob(x) under optax.adam(pretrain_lr)  (:45-49, 70-80)
  2. a diagonal-covariance GaussianMixture (scikit-learn, host side, as in the reference) fitted on the encoder means of the
     training batches initialises vade/{logits, mu, log_scale}  (:82-123)
  3. main training: loss = -mean VADE.elbo(x) under chain(scale_by_adam(**adam), scale_by_schedule(exponential_decay),
     scale(-1)); validation logs the clustering accuracy of argmax_c q(c | x)  (:125-159)

Differences forced by the environment (no network, no tfds / TensorBoard): data are synthetic batches of the dataset's shape
with synthetic labels, or .npy arrays given with --data / --labels; scalars go to runs/<id>/tb/scalars.jsonl.
"""
import argparse
import json
import os
import pickle
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.clustering import ClusteringAccuracyCallback, clustering_accuracy  # noqa: E402
from posterior_matching_amd.config_dict import apply_overrides, load_config_file  # noqa: E402
from posterior_matching_amd.data import make_dataset  # noqa: E402
from posterior_matching_amd.models.vade import VADE  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import (CheckpointCallback, LearningRateLoggerCallback, Trainer, VADELoss,  # noqa: E402
                                            VADEPretrainLoss)
from posterior_matching_amd.utils import TensorBoardCallback, configure_environment, make_run_dir  # noqa: E402

configure_environment()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", default=None, help="optional .npy with the training examples")
    ap.add_argument("--labels", default=None, help="optional .npy with their integer class labels (clustering accuracy)")
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
    labels = np.load(args.labels) if args.labels else True
    train_dataset = make_dataset(config.data, config.data.train_batch_size, 64, config.seed + rank, device, training=True,
                                 arrays=arrays, labels=labels)
    val_dataset = make_dataset(config.data, config.data.val_batch_size, 8, config.seed + 10007 + rank, device,
                               training=False, arrays=arrays, labels=labels)
    data_key = train_dataset.key

    model = VADE.from_config(config.model, device=device, seed=config.seed)

    def pred_fn(batch):                                   # train_vade.py:57-61
        probs = model.predict_cluster(batch[data_key].to(device), config.cluster_pred_num_samples, seed=config.seed)
        return probs.argmax(-1)

    run_dir = make_run_dir(prefix=f"vade-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)

    # PRETRAINING
    pretrain_trainer = Trainer(VADEPretrainLoss(config, model, data_key, seed=config.seed), optim.adam(config.pretrain_lr),
                               num_devices=world, seed=config.seed)
    print("Pretraining...")
    pretrain_state = pretrain_trainer.fit(train_dataset, config.pretrain_steps)
    if rank == 0:
        with open(os.path.join(run_dir, "pretrain_state.pkl"), "wb") as fp:
            pickle.dump(pretrain_state, fp)

    # GMM on the encoder means (host side: scikit-learn, as in the reference)
    print("Fitting GMM...")
    from sklearn.mixture import GaussianMixture

    def encode(dataset):
        zs, ys = [], []
        for batch in dataset.batches:
            zs.append(model.encode_mean(batch[data_key].to(device)).cpu().numpy())
            ys.append(batch["label"].numpy())
        return np.concatenate(zs, 0), np.concatenate(ys, 0)

    latents, _ = encode(train_dataset)
    val_latents, targets = encode(val_dataset)
    gmm = GaussianMixture(n_components=config.model.num_components, covariance_type="diag", max_iter=300, n_init=10,
                          random_state=config.seed % (2 ** 31))
    gmm.fit(latents)
    gmm_acc = clustering_accuracy(targets, gmm.predict(val_latents))
    print("GMM Accuracy:", round(gmm_acc, 4))
    initial_params = dict(pretrain_state.params)
    initial_params.update({"vade/logits": np.log(gmm.weights_).astype(np.float32), "vade/mu": gmm.means_.astype(np.float32),
                           "vade/log_scale": np.log(gmm.covariances_).astype(np.float32)})     # train_vade.py:115-121, as written

    # MAIN TRAINING
    if rank == 0:
        with open(os.path.join(run_dir, "model_config.json"), "w") as fp:
            json.dump(config.model.to_dict(), fp)
    schedule = optim.exponential_decay(**config.lr_schedule)
    optimizer = optim.chain(optim.scale_by_adam(**config.get("adam", {})), optim.scale_by_schedule(schedule), optim.scale(-1.0))
    trainer = Trainer(VADELoss(config, model, data_key, seed=config.seed), optimizer, num_devices=world, seed=config.seed)
    callbacks = [
        ClusteringAccuracyCallback(pred_fn),
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        LearningRateLoggerCallback(schedule),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    print("Starting main training...")
    trainer.fit(train_dataset, config.steps, val_dataset=val_dataset, validation_freq=config.validation_freq,
                callbacks=callbacks, initial_params=initial_params)


if __name__ == "__main__":
    main()
