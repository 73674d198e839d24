"""TEST INFRASTRUCTURE - float64 CPU restatement of the reference's lookahead posteriors (posterior_matching/models/lookahead.py,
train_lookahead_posterior.py).  Only tests/ may import this module.

PARITY UNPINNED: the reference cannot be imported here (jax / haiku / tfp absent) and ships no fixtures for this model; every
function cites the lines it follows, and tests/test_oracle_kat.py pins the algebra with known answers (the masked average of
Gaussian log-densities against scipy.stats, the closed-form entropy difference of expected_info_gains).

Layout: the reference stacks the model samples in front ([z, b, s, ...]); here rows are (b, z, s) - the layout of the HIP
kernels - and the explicit noise tensors are indexed the same way, so both sides consume identical draws:
  noise["eps"]      [B, Z, k]      z ~ q(z | x_o)              (lookahead.py:136-138)
  noise["eps_look"] [B, Z, S, k]   z ~ q(z | x_o, x_i = sample) (lookahead.py:178-185)
  (+ "gumbel" [B, Z, k, nc], "gumbel_look" [B, Z, S, k, nc] when the partial posterior is an AutoregressiveGMM)
  inds [S]: the subsampled features (jax.random.choice without replacement, lookahead.py:151-156)
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import pm_vae_oracle as O

Tensor = torch.Tensor
Params = Dict[str, Tensor]
LOG_2PI = math.log(2.0 * math.pi)


def _net_cfg(look_cfg: dict, pm_cfg: dict):
    """LookaheadPosterior.from_config (lookahead.py:101-108): the lookahead encoder defaults to the PM-VAE's encoder network"""
    kind = look_cfg.get("lookahead_encoder_net", pm_cfg["encoder_net"])
    cfg = look_cfg.get("lookahead_encoder_net_config", pm_cfg.get("encoder_net_config")) or {}
    return kind, cfg


def param_shapes(look_cfg: dict, pm_cfg: dict, x_shape: Sequence[int]) -> Dict[str, Tuple[int, ...]]:
    """lookahead_encoder_net on [x_o | b], then LookaheadBlock's hk.Linear(2 k num_features) (lookahead.py:26-31, 76-79)"""
    kind, cfg = _net_cfg(look_cfg, pm_cfg)
    tmp = {"latent_dim": pm_cfg["latent_dim"], "encoder_net": "ResidualMLP", "encoder_net_config": {"residual_blocks": 0, "hidden_units": 1},
           "posterior_dist": "TriLGaussian", "decoder_net": "ResidualMLP", "decoder_net_config": {"residual_blocks": 0, "hidden_units": 1},
           "decoder_dist": "Bernoulli", "partial_encoder_net": kind, "partial_encoder_net_config": cfg,
           "partial_posterior_dist": "DiagonalGaussian"}
    full = O.param_shapes(tmp, tuple(x_shape))
    out: Dict[str, Tuple[int, ...]] = {}
    fin = None
    for n, s in full.items():
        if n.startswith("partial_encoder_net/"):
            out["lookahead_encoder_net/" + n[len("partial_encoder_net/"):]] = tuple(s)
        if n == "partial_posterior_dist/linear/w":
            fin = s[0]
    k, F = pm_cfg["latent_dim"], look_cfg["num_features"]
    out["lookahead_block/linear/w"] = (fin, 2 * k * F)
    out["lookahead_block/linear/b"] = (2 * k * F,)
    return out


def init_params(look_cfg: dict, pm_cfg: dict, x_shape: Sequence[int], seed: int = 3, dtype=torch.float64) -> Params:
    """haiku defaults: TruncatedNormal(1 / sqrt(fan_in)) weights, zero biases"""
    from scipy.special import ndtr, ndtri

    rng = np.random.default_rng(seed)
    lo, hi = ndtr(-2.0), ndtr(2.0)
    out: Params = {}
    for name, shp in param_shapes(look_cfg, pm_cfg, x_shape).items():
        if name.endswith("/w"):
            fan_in = shp[0] * shp[1] * shp[2] if len(shp) == 4 else shp[0]
            arr = ndtri(lo + rng.uniform(size=shp) * (hi - lo)) / math.sqrt(fan_in)
        else:
            arr = np.zeros(shp)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out


def lookahead_params(p_look: Params, look_cfg: dict, pm_cfg: dict, x: Tensor, b: Tensor) -> Tensor:
    """lookahead_encoder(x_o_b) -> [B, F, 2k] (lookahead.py:21-36, 187)"""
    kind, cfg = _net_cfg(look_cfg, pm_cfg)
    feats = O._net(p_look, kind, cfg, "lookahead_encoder_net", torch.cat([x * b, b], -1))
    out = O.linear(feats.reshape(feats.shape[0], -1), p_look["lookahead_block/linear/w"], p_look["lookahead_block/linear/b"])
    return out.reshape(x.shape[0], look_cfg["num_features"], 2 * pm_cfg["latent_dim"])


def model_one_step_z(p_vae: Params, pm_cfg: dict, x: Tensor, b: Tensor, noise: Dict[str, Tensor], inds) -> Tuple[Tensor, Tensor]:
    """lookahead.py:131-185 -> (model_one_step_z [B, Z, S, k], valid [B, S]); no gradient flows from here (stop_gradient)"""
    B, Z = noise["eps"].shape[0], noise["eps"].shape[1]
    S = len(inds)
    x_o = x * b
    with torch.no_grad():
        kind, cfg, pfeats, k = O._partial_posterior(p_vae, pm_cfg, x_o, b)
        z, _ = O._sample_partial_posterior(p_vae, kind, cfg, pfeats, k, noise, Z)                 # rows b*Z + z
        dkind, out = O._decoder(p_vae, pm_cfg, z)
        mean = torch.sigmoid(out) if dkind == "Bernoulli" else out                                # decoder(z).mean()
        mean = mean.reshape((B, Z) + tuple(x.shape[1:]))
        samples = torch.where(b[:, None] == 1, x_o[:, None], mean)                                # :143-145
        F = int(np.prod(b.shape[1:]))
        one_hots = torch.eye(F, dtype=x.dtype).reshape((F,) + tuple(b.shape[1:]))[list(inds)]     # [S, ...]
        b_look = torch.maximum(b[:, None], one_hots[None])                                        # [B, S, ...]
        x_look = samples[:, :, None] * b_look[:, None]                                            # [B, Z, S, ...]
        valid = (b[:, None] + one_hots[None]).reshape(B, S, -1).amax(-1) < 2                      # :166-173
        b_rep = b_look[:, None].expand((B, Z, S) + tuple(b.shape[1:]))
        xf = x_look.reshape((B * Z * S,) + tuple(x.shape[1:]))
        bf = b_rep.reshape((B * Z * S,) + tuple(b.shape[1:]))
        kind2, cfg2, pf2, _ = O._partial_posterior(p_vae, pm_cfg, xf, bf)
        n2 = {"eps": noise["eps_look"].reshape(B * Z * S, 1, k)}
        if "gumbel_look" in noise:
            n2["gumbel"] = noise["gumbel_look"].reshape((B * Z * S, 1) + tuple(noise["gumbel_look"].shape[3:]))
        z2, _ = O._sample_partial_posterior(p_vae, kind2, cfg2, pf2, k, n2, 1)
    return z2.reshape(B, Z, S, k), valid


def masked_mean_ll(params: Tensor, zs: Tensor, valid: Tensor, inds) -> Tensor:
    """lookahead.py:188-203: params [B, F, 2k] -> per example the mean over the valid subsampled features of the mean over the
    model samples of log N(z; loc_f, softplus(raw_f) + 1e-5); 0 where nothing is valid"""
    k = zs.shape[-1]
    sub = params[:, list(inds)]                                        # [B, S, 2k]
    loc, scale = sub[..., :k], O.softplus(sub[..., k:]) + 1e-5
    d = (zs - loc[:, None]) / scale[:, None]
    lp = (-0.5 * d * d - torch.log(scale[:, None]) - 0.5 * LOG_2PI).sum(-1)        # [B, Z, S]
    lls = lp.mean(1) * valid.to(lp.dtype)
    denom = valid.sum(-1)
    out = lls.sum(-1) / denom.clamp(min=1).to(lp.dtype)
    return torch.where(denom == 0, torch.zeros_like(out), out)


def lookahead_lls(p_look: Params, p_vae: Params, look_cfg: dict, pm_cfg: dict, x: Tensor, b: Tensor, noise: Dict[str, Tensor],
                  inds) -> Tensor:
    """LookaheadPosterior.__call__ (lookahead.py:128-203) -> [B]"""
    zs, valid = model_one_step_z(p_vae, pm_cfg, x, b, noise, inds)
    return masked_mean_ll(lookahead_params(p_look, look_cfg, pm_cfg, x, b), zs, valid, inds)


def loss(p_look: Params, p_vae: Params, look_cfg: dict, pm_cfg: dict, x: Tensor, b: Tensor, noise, inds) -> Tensor:
    """train_lookahead_posterior.py:46-52"""
    return -lookahead_lls(p_look, p_vae, look_cfg, pm_cfg, x, b, noise, inds).mean()


def expected_info_gains(p_look: Params, p_vae: Params, look_cfg: dict, pm_cfg: dict, x: Tensor, b: Tensor) -> Tensor:
    """lookahead.py:205-227 for ONE instance (x, b without a batch axis) -> [num_features]"""
    k = pm_cfg["latent_dim"]
    feats = O._net(p_vae, pm_cfg["encoder_net"], pm_cfg.get("encoder_net_config") or {}, "encoder_net", x[None])
    if pm_cfg["posterior_dist"] == "TriLGaussian":
        _, tril = O.tril_gaussian_params(p_vae, "posterior_dist", feats, k)
        cur = 0.5 * k * (1.0 + LOG_2PI) + torch.log(torch.diagonal(tril[0])).sum()
    else:
        _, scale = O.diagonal_gaussian_params(p_vae, "posterior_dist", feats, k)
        cur = 0.5 * k * (1.0 + LOG_2PI) + torch.log(scale[0]).sum()
    prm = lookahead_params(p_look, look_cfg, pm_cfg, x[None], b[None])[0]          # [F, 2k]
    ents = 0.5 * k * (1.0 + LOG_2PI) + torch.log(O.softplus(prm[:, k:]) + 1e-5).sum(-1)
    gains = cur - ents
    return torch.where(b.reshape(-1) == 0, gains, torch.full_like(gains, -math.inf))
