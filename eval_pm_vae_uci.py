#!/usr/bin/env python3
"""Imputation NRMSE and arbitrary-conditional log-likelihood of a PM-VAE run on tabular (UCI-style) data.

Counterpart of the reference's eval_pm_vae_uci.py:46-138: Bernoulli(0.5) masks, `num_trials` passes over the
evaluation rows; per batch the mean of `num_samples` imputations (PosteriorMatchingVAE.impute) and the
importance-sampled log p(x_u | x_o) (PosteriorMatchingVAE.is_log_prob); NRMSE as defined there (:60-66).

    python eval_pm_vae_uci.py --run_dir runs/pm-vae-gas-<id> --dataset gas [--data test.npy]

Data are synthetic standard-normal rows of the dataset's width unless --data gives a [N, D] .npy.  Masks are drawn on
the device (pm_bernoulli_mask).  Writes uci_results/{nrmse,ac_lls}.npy like the reference.
"""
import argparse
import json
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd.data import data_shape  # noqa: E402
from posterior_matching_amd.masking import get_mask_generator  # noqa: E402
from posterior_matching_amd.models import PosteriorMatchingVAE  # noqa: E402
from posterior_matching_amd.utils import configure_environment  # noqa: E402

configure_environment()


def nrmse_score(imputations, true_data, observed_mask):
    """reference eval_pm_vae_uci.py:60-66"""
    import numpy as np

    error = (imputations - true_data) ** 2
    mse = np.sum(error, axis=-2) / np.count_nonzero(1.0 - observed_mask, axis=-2)
    nrmse = np.sqrt(mse) / np.std(true_data, axis=-2)
    return np.mean(nrmse, axis=-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run_dir", required=True, help="The run directory of the model to evaluate.")
    ap.add_argument("--dataset", required=True, help="The dataset to evaluate on.")
    ap.add_argument("--num_instances", type=int, default=None, help="The number of instances to evaluate.")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--num_samples", type=int, default=512, help="The number of samples to use for expectations.")
    ap.add_argument("--num_trials", type=int, default=5, help="The number of trials to compute means and std. over.")
    ap.add_argument("--data", default=None, help="optional .npy [N, D] with the evaluation rows")
    args = ap.parse_args()
    import numpy as np
    import torch

    with open(os.path.join(args.run_dir, "model_config.json")) as fp:
        model_config = json.load(fp)
    with open(os.path.join(args.run_dir, "train_state.pkl"), "rb") as fp:
        state = pickle.load(fp)

    device = torch.device("cuda", 0)
    shape = data_shape(args.dataset)
    if args.data:
        data_np = np.load(args.data).astype(np.float32)
    else:
        data_np = np.random.default_rng(0).normal(size=(args.num_instances or 1024,) + shape).astype(np.float32)
    if args.num_instances is not None:
        data_np = data_np[:args.num_instances]
    n = data_np.shape[0] // args.batch_size * args.batch_size                  # drop_remainder=True (:51)
    data_np = data_np[:n]

    model = PosteriorMatchingVAE.from_config(model_config, device=device)
    model.init(shape)
    model.load_params(state.params)
    masks_gen = get_mask_generator("BernoulliMaskGenerator", device=device, seed=91)     # hk.PRNGSequence(91) (:100)

    imputations, masks, lls = [], [], []
    for trial in range(args.num_trials):
        imps, ms, ls = [], [], []
        for i in range(0, n, args.batch_size):
            x = torch.from_numpy(data_np[i:i + args.batch_size]).to(device)
            b = masks_gen(tuple(x.shape))
            seed = 91 + 7919 * (trial * (n // args.batch_size) + i // args.batch_size)
            imp = model.impute(x, b, num_samples=args.num_samples, seed=seed).mean(0)
            _, ll = model.is_log_prob(x, b, num_samples=args.num_samples, seed=seed + 1)
            imps.append(imp.cpu().numpy())
            ms.append(b.cpu().numpy())
            ls.append(ll.cpu().numpy())
        imputations.append(np.vstack(imps))
        masks.append(np.vstack(ms))
        lls.append(np.hstack(ls))
    imputations, masks, lls = np.array(imputations), np.array(masks), np.array(lls)
    x = np.broadcast_to(data_np[None], (args.num_trials,) + data_np.shape)
    nrmse = nrmse_score(imputations, x, masks)
    lls = np.mean(lls, axis=1)

    results_dir = os.path.join(args.run_dir, "uci_results")
    os.makedirs(results_dir, exist_ok=True)
    np.save(os.path.join(results_dir, "nrmse.npy"), nrmse)
    np.save(os.path.join(results_dir, "ac_lls.npy"), lls)
    print("\n****RESULTS****")
    print(f"NRMSE: {np.mean(nrmse).item()} ± {np.std(nrmse).item()}")
    print(f"AC LL: {np.mean(lls).item()} ± {np.std(lls).item()}")


if __name__ == "__main__":
    main()
