#!/usr/bin/env python3
"""Held-out imputation PSNR of a PM-VDVAE run (the parity metric BASELINE.json names for the VDVAE).

Counterpart of the reference's eval_pm_vdvae_imputation.py:100-130 for the PSNR part:

    python eval_pm_vdvae_imputation.py --run_dir runs/pm-vdvae-mnist-<id> [--num_samples 5 --num_instances 256]

Evaluates with the EMA parameters of the checkpoint (what the reference's Trainer(use_ema_for_eval) validates with).
"""
import argparse
import json
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd.data import SyntheticDataset  # noqa: E402
from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE, vdvae_imputation_psnr  # noqa: E402
from posterior_matching_amd.utils import configure_environment  # noqa: E402

configure_environment()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run_dir", required=True)
    ap.add_argument("--dataset", default="mnist")
    ap.add_argument("--mask_generator", default="MNISTMaskGenerator")
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--num_instances", type=int, default=256)
    ap.add_argument("--num_samples", type=int, default=5)
    ap.add_argument("--data", default=None)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch

    with open(os.path.join(args.run_dir, "model_config.json")) as fp:
        model_config = json.load(fp)
    with open(os.path.join(args.run_dir, "train_state.pkl"), "rb") as fp:
        state = pickle.load(fp)
    device = torch.device("cuda", 0)
    data_cfg = {"dataset": args.dataset, "mask_generator": args.mask_generator}
    nb = max(1, args.num_instances // args.batch_size)
    arrays = np.load(args.data) if args.data else None
    ds = SyntheticDataset(data_cfg, args.batch_size, nb, args.seed, device, training=False, arrays=arrays,
                          normalize_images=False)
    model = PosteriorMatchingVDVAE(**model_config, device=device)
    model.init()
    model.load_params(state.ema_params if state.ema_params is not None else state.params)
    psnrs = []
    for i, batch in enumerate(ds.batches):
        imp = model.impute(batch["image"], batch["mask"], num_samples=args.num_samples, seed=args.seed + i)
        psnrs.append(vdvae_imputation_psnr(imp, batch["image"]).cpu().numpy())
    psnrs = np.concatenate(psnrs)
    out_dir = os.path.join(args.run_dir, "imputation_results")
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "psnrs.npy"), psnrs)
    print(json.dumps({"mean_psnr": float(np.ma.masked_invalid(psnrs).mean()), "num_instances": int(psnrs.size),
                      "num_samples": args.num_samples}))


if __name__ == "__main__":
    main()
