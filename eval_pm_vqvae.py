#!/usr/bin/env python3
"""Held-out imputation PSNR of a PM-VQVAE run (the parity metric BASELINE.json names for stage 2).

Counterpart of the reference's eval_pm_vqvae.py:103-138 for the PSNR part (the precision / recall
scores of that script need a TF-hub Inception network, which is not available offline):

    python eval_pm_vqvae.py --run_dir runs/pm-vqvae-mnist-<id> [--num_samples 5 --num_instances 256]
"""
import argparse
import json
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd.data import SyntheticDataset, data_shape  # noqa: E402
from posterior_matching_amd.models.vqvae import VQVAE, build_partial_posterior, imputation_psnr, vqvae_impute  # noqa: E402
from posterior_matching_amd.utils import configure_environment  # noqa: E402

configure_environment()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run_dir", required=True)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--mask_generator", default=None)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--num_instances", type=int, default=256)
    ap.add_argument("--num_samples", type=int, default=5)
    ap.add_argument("--data", default=None, help="optional .npy with the evaluation examples")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch

    with open(os.path.join(args.run_dir, "vqvae_config.json")) as fp:
        vqvae_config = json.load(fp)
    with open(os.path.join(args.run_dir, "config.json")) as fp:
        config = json.load(fp)
    with open(os.path.join(args.run_dir, "train_state.pkl"), "rb") as fp:
        state = pickle.load(fp)

    device = torch.device("cuda", 0)
    data_cfg = dict(config["data"])
    if args.dataset:
        data_cfg["dataset"] = args.dataset
    if args.mask_generator:
        data_cfg["mask_generator"] = args.mask_generator
    x_shape = data_shape(data_cfg["dataset"])
    nb = max(1, args.num_instances // args.batch_size)
    arrays = np.load(args.data) if args.data else None
    ds = SyntheticDataset(data_cfg, args.batch_size, nb, args.seed, device, training=False, arrays=arrays)

    vqvae = VQVAE(**vqvae_config, device=device)
    vqvae.init(x_shape)
    pc_cfg = {k: v for k, v in config["pixel_cnn"].items() if k != "num_indices"}
    penc, pcnn, store = build_partial_posterior(vqvae, config["conditional_dim"], pc_cfg, x_shape)
    vqvae.load_params({k[len("vqvae/"):]: v for k, v in state.params.items() if k.startswith("vqvae/")})
    vqvae.load_state({k[len("vqvae/"):]: v for k, v in state.state.items() if k.startswith("vqvae/")})
    store.load_dict({k: v for k, v in state.params.items() if not k.startswith("vqvae/")})

    psnrs = []
    for i, batch in enumerate(ds.batches):
        imp = vqvae_impute(vqvae, penc, pcnn, batch["image"], batch["mask"], num_samples=args.num_samples,
                           seed=args.seed + i)
        psnrs.append(imputation_psnr(imp, batch["image"]).cpu().numpy())
    psnrs = np.concatenate(psnrs)
    finite = np.ma.masked_invalid(psnrs)
    out_dir = os.path.join(args.run_dir, "imputation_results")
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "psnrs.npy"), psnrs)
    print(json.dumps({"mean_psnr": float(finite.mean()), "num_instances": int(psnrs.size),
                      "num_samples": args.num_samples}))


if __name__ == "__main__":
    main()
