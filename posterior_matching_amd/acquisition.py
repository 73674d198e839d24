"""Greedy active feature acquisition (reference posterior_matching/acquisition.py:13-127): same function names.  The reference
builds the model inside a haiku transform and scans the episode on the device; here the model object is built once by the
caller (its parameters loaded) and the episode is a host loop over launches - every step is a handful of small kernels on one
instance (expected_info_gains of both kinds, impute, pm_acquisition_policy, pm_reconstruction_rmse)."""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import ops
from .models.lookahead import LookaheadPosterior


def rmse(true: torch.Tensor, pred: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """acquisition.py:13-15: sqrt(mean((true - pred)^2 (1 - b))) as a device scalar [1]"""
    out, rec = torch.empty(1, device=true.device), torch.empty_like(true)
    ops.reconstruction_rmse(pred.reshape((1,) + tuple(true.shape)).contiguous(), true.contiguous(), b.contiguous(), rec, out)
    return out


def make_acquisition_eval_fn(lookahead_config: Dict[str, Any], pm_vae_config: Dict[str, Any], num_samples: int,
                             model: Optional[LookaheadPosterior] = None, device=None, seed: int = 0
                             ) -> Callable[..., Dict[str, torch.Tensor]]:
    """acquisition.py:18-64.  `model`: the LookaheadPosterior to evaluate (built from the two configs when None - load its
    parameters through `eval_fn.model.load_params`).  eval_fn(x_o, b, noise=None) -> {"sampling_action", "lookahead_action",
    "sampling_probs", "lookahead_probs", "reconstruction"} for ONE instance; noise: {"eps" [1,S,k]} (both the sampling
    estimate and the imputations consume it in parity mode), device Philox draws otherwise."""
    if model is None:
        model = LookaheadPosterior.from_config(lookahead_config, pm_vae_config, device=device, seed=seed)
    S = int(num_samples)
    calls = [0]

    def eval_fn(x_o: torch.Tensor, b: torch.Tensor, noise=None) -> Dict[str, torch.Tensor]:
        if model.store is None:
            model.init(tuple(x_o.shape), x_o.device)
        dev, F = x_o.device, b.numel()
        calls[0] += 1
        sd = seed + 7919 * calls[0]
        sampling = model.pm_vae.expected_info_gains(x_o, b, S, noise=noise, seed=sd).clone()
        look = model.expected_info_gains(x_o, b).clone()
        out: Dict[str, torch.Tensor] = {}
        for name, gains in (("sampling", sampling), ("lookahead", look)):
            probs, action = torch.empty(F, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
            ops.acquisition_policy(gains, probs, action)
            out[f"{name}_action"], out[f"{name}_probs"] = action, probs
        imp = model.pm_vae.impute(x_o.unsqueeze(0).contiguous(), b.unsqueeze(0).contiguous(), S, noise=noise, seed=sd + 1)
        recon, tmp = torch.empty_like(x_o), torch.empty(1, device=dev)
        ops.reconstruction_rmse(imp.contiguous().view((S,) + tuple(x_o.shape)), x_o.contiguous(), b.contiguous(), recon, tmp)
        out["reconstruction"] = recon
        return out

    eval_fn.model = model
    return eval_fn


def make_collect_trajectory_fn(eval_fn: Callable[..., Dict[str, torch.Tensor]], episode_length: int
                               ) -> Callable[[torch.Tensor], Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]]:
    """acquisition.py:67-127: two episodes from an empty mask, one following the sampling-based information gains, one the
    lookahead posteriors'; every entry of the returned dicts is stacked over the steps ([episode_length, ...] numpy arrays:
    the five eval_fn outputs + "rmse" + "mask")."""

    def episode(x: torch.Tensor, which: str) -> Dict[str, np.ndarray]:
        mshape = tuple(x.shape[:-1]) + (1,) if x.dim() == 3 else tuple(x.shape)
        cur_b = torch.zeros(mshape, device=x.device)
        F = math.prod(mshape)
        steps = []
        for _ in range(int(episode_length)):
            x_o = (x * cur_b).contiguous()
            data = eval_fn(x_o, cur_b)
            err, rec = torch.empty(1, device=x.device), torch.empty_like(x)
            ops.reconstruction_rmse(data["reconstruction"].reshape((1,) + tuple(x.shape)).contiguous(), x.contiguous(), cur_b, rec, err)
            row = {k: v.detach().cpu().numpy() for k, v in data.items()}
            row["sampling_action"], row["lookahead_action"] = int(row["sampling_action"][0]), int(row["lookahead_action"][0])
            row["rmse"], row["mask"] = float(err.item()), cur_b.cpu().numpy()
            steps.append(row)
            one = torch.zeros(F, device=x.device)
            one[row[f"{which}_action"]] = 1.0                                         # new_b = cur_b + one_hot(action)
            cur_b = cur_b + one.view(mshape)
        return {k: np.stack([np.asarray(s[k]) for s in steps]) for k in steps[0]}

    def collect_trajectory(x: torch.Tensor):
        return episode(x, "sampling"), episode(x, "lookahead")

    return collect_trajectory
