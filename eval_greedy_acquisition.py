#!/usr/bin/env python3
"""Greedy active feature acquisition with a trained lookahead model on the MI355X-native path.

Same entry point as the reference's eval_greedy_acquisition.py:

    python eval_greedy_acquisition.py --run_dir runs/lookahead-mnist16-<id> --dataset mnist16 [--num_instances 1000]
                                      [--num_samples 50] [--episode_length 31] [--data test_images.npy]

`run_dir` is a directory written by train_lookahead_posterior.py (lookahead_config.json, pm_vae_config.json, train_state.pkl).
Writes run_dir/trajectories/{sampling,lookahead}_trajectories.pkl: one dict of [episode_length, ...] arrays per instance (the
eval_fn outputs + "rmse" + "mask" + "truth").  There is no network here, so the instances come from --data (an .npy of
[N, ...] examples in [0, 1]) or from the seeded synthetic pool of posterior_matching_amd.data.
"""
import argparse
import json
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd.acquisition import make_acquisition_eval_fn, make_collect_trajectory_fn  # noqa: E402
from posterior_matching_amd.utils import configure_environment  # noqa: E402

configure_environment()


def load_data(dataset, num_instances, path=None, seed=91):
    import numpy as np

    from posterior_matching_amd.data import SyntheticDataset

    if path:
        data = np.load(path).astype(np.float32)
        return data[:num_instances] if num_instances is not None else data
    n = num_instances or 1000
    ds = SyntheticDataset({"dataset": dataset}, batch_size=32, num_batches=-(-n // 32), seed=seed, training=False)
    return np.concatenate([b[ds.key].numpy() for b in ds.batches])[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run_dir", required=True, help="run directory written by train_lookahead_posterior.py")
    ap.add_argument("--dataset", required=True, help="dataset name (mnist16 for the reference configuration)")
    ap.add_argument("--num_instances", type=int, default=1000, help="how many test instances to run episodes for")
    ap.add_argument("--num_samples", type=int, default=50, help="samples per expectation (sampling-based gains, imputations)")
    ap.add_argument("--episode_length", type=int, default=31, help="acquisitions per episode")
    ap.add_argument("--data", default=None, help="optional .npy with the evaluation examples")
    args = ap.parse_args()

    import torch

    data = load_data(args.dataset, args.num_instances, args.data)
    with open(os.path.join(args.run_dir, "lookahead_config.json"), "r") as fp:
        lookahead_config = json.load(fp)
    with open(os.path.join(args.run_dir, "pm_vae_config.json"), "r") as fp:
        pm_vae_config = json.load(fp)
    with open(os.path.join(args.run_dir, "train_state.pkl"), "rb") as fp:
        model_state = pickle.load(fp)

    device = torch.device("cuda", 0)
    eval_fn = make_acquisition_eval_fn(lookahead_config, pm_vae_config, args.num_samples, device=device, seed=91)
    eval_fn.model.init(tuple(data.shape[1:]), device)
    eval_fn.model.load_params(model_state.params, require_trainable=True)     # evaluation: the lookahead networks too
    collect_trajectory = make_collect_trajectory_fn(eval_fn, args.episode_length)

    sampling_trajectories, lookahead_trajectories = [], []
    for i, x in enumerate(data):
        sampling_traj, look_traj = collect_trajectory(torch.from_numpy(x).to(device))
        sampling_traj["truth"], look_traj["truth"] = x, x
        sampling_trajectories.append(sampling_traj)
        lookahead_trajectories.append(look_traj)
        if (i + 1) % 50 == 0:
            print(f"{i + 1} / {len(data)} episodes", flush=True)

    results_dir = os.path.join(args.run_dir, "trajectories")
    os.makedirs(results_dir, exist_ok=True)
    with open(os.path.join(results_dir, "sampling_trajectories.pkl"), "wb") as fp:
        pickle.dump(sampling_trajectories, fp)
    with open(os.path.join(results_dir, "lookahead_trajectories.pkl"), "wb") as fp:
        pickle.dump(lookahead_trajectories, fp)


if __name__ == "__main__":
    main()
