"""TEST INFRASTRUCTURE - float64 CPU restatement of the reference's VaDE / PM-VaDE (posterior_matching/models/vade.py,
train_vade.py, train_pm_vade.py, posterior_matching/clustering.py).  Only tests/ may import this module.

PARITY UNPINNED: the reference cannot be imported here (jax / haiku / tfp / distrax absent) and ships no fixtures; every
function cites the lines it follows, and tests/test_oracle_kat.py pins the algebra with known answers (the ELBO written with
the responsibilities, as the reference writes it, equals rec_ll + log p(z) - log q(z | x) with the mixture marginal; cluster
probabilities sum to one; clustering accuracy of a permuted labelling is 1).

Third-party semantics taken from knowledge of the pinned versions (each a parity risk):
  * distrax.Categorical(logits).logits returns NORMALISED log-probabilities (logits - logsumexp(logits)): vade.py:116,136,141
    therefore use log pi_c, not the raw parameter.
  * tfd.MultivariateNormalDiag(loc, scale_diag).log_prob sums the per-dimension normal log-densities.
  * haiku initialisers: Constant(0) for `logits`, RandomNormal() (stddev 1, mean 0) for `mu` and `log_scale` (vade.py:40-54).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import pm_vae_oracle as O

Tensor = torch.Tensor
Params = Dict[str, Tensor]
LOG_2PI = math.log(2.0 * math.pi)


# ---- parameters -----------------------------------------------------------------------------------------------------------
def param_shapes(model_cfg: dict, x_shape: Sequence[int], partial: bool = False) -> Dict[str, Tuple[int, ...]]:
    """haiku creation order of VADE.__init__ / PosteriorMatchingVADE.__init__ (vade.py:28-66, 157-176) in this repo's names:
    vade/{logits,mu,log_scale}; encoder_net + posterior_dist (DiagonalGaussian(latent_dim), hk.Sequential "encoder");
    decoder_net + decoder_dist ("decoder"); [partial_encoder_net + partial_posterior_dist ("partial_encoder")]."""
    k, C = model_cfg["latent_dim"], model_cfg["num_components"]
    cfg = {"latent_dim": k, "encoder_net": model_cfg["encoder_net"], "decoder_net": model_cfg["decoder_net"],
           "encoder_net_config": model_cfg.get("encoder_net_config"), "decoder_net_config": model_cfg.get("decoder_net_config"),
           "posterior_dist": "DiagonalGaussian", "decoder_dist": model_cfg["decoder_dist"],
           "decoder_dist_config": model_cfg.get("decoder_dist_config"),
           "partial_posterior_dist": model_cfg.get("partial_posterior_dist", "TriLGaussian"),
           "partial_posterior_dist_config": model_cfg.get("partial_posterior_dist_config")}
    if model_cfg.get("partial_encoder_net"):
        cfg["partial_encoder_net"] = model_cfg["partial_encoder_net"]
        cfg["partial_encoder_net_config"] = model_cfg.get("partial_encoder_net_config")
    full = O.param_shapes(cfg, tuple(x_shape))
    out: Dict[str, Tuple[int, ...]] = {"vade/logits": (C,), "vade/mu": (C, k), "vade/log_scale": (C, k)}
    for n, s in full.items():
        if n.startswith("partial_") and not partial:
            continue
        out[n] = tuple(s)
    return out


def init_params(model_cfg: dict, x_shape: Sequence[int], seed: int = 1, partial: bool = False, dtype=torch.float64) -> Params:
    from scipy.special import ndtr, ndtri

    rng = np.random.default_rng(seed)
    lo, hi = ndtr(-2.0), ndtr(2.0)
    out: Params = {}
    for name, shp in param_shapes(model_cfg, x_shape, partial).items():
        if name == "vade/logits":
            arr = np.zeros(shp)
        elif name in ("vade/mu", "vade/log_scale"):
            arr = rng.normal(size=shp)
        elif name.endswith("/w"):
            fan_in = int(np.prod(shp[:-1])) if len(shp) == 2 else shp[0] * shp[1] * (shp[3] if "/conv_t_" in name else shp[2])
            arr = ndtri(lo + rng.uniform(size=shp) * (hi - lo)) / math.sqrt(fan_in)
        else:
            arr = np.zeros(shp)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out


# ---- pieces ----------------------------------------------------------------------------------------------------------------
def _cfg(model_cfg: dict) -> dict:
    return dict(model_cfg, posterior_dist="DiagonalGaussian")


def encoder_params(p: Params, model_cfg: dict, x: Tensor) -> Tuple[Tensor, Tensor]:
    """self.encoder(x): encoder_net -> DiagonalGaussian(latent_dim) (vade.py:60-62; distributions.py:58-84) -> (loc, scale)"""
    feat = O._net(p, model_cfg["encoder_net"], model_cfg.get("encoder_net_config") or {}, "encoder_net", x)
    return O.diagonal_gaussian_params(p, "posterior_dist", feat, model_cfg["latent_dim"])


def decoder_log_prob(p: Params, model_cfg: dict, z: Tensor, x: Tensor) -> Tensor:
    """self.decoder(z).log_prob(x): decoder_net -> decoder_dist -> Independent (vade.py:63-65)"""
    feat = O._net(p, model_cfg["decoder_net"], model_cfg.get("decoder_net_config") or {}, "decoder_net", z)
    B = x.shape[0]
    if model_cfg["decoder_dist"] == "Bernoulli":
        return O.bernoulli_log_prob(feat.reshape(B, -1), x.reshape(B, -1)).sum(-1)
    if model_cfg["decoder_dist"] == "IdentityGaussian":
        loc = O.linear(feat.reshape(B, -1), p["decoder_dist/linear/w"], p["decoder_dist/linear/b"])
        return O.normal_log_prob(x.reshape(B, -1), loc, torch.exp(p["decoder_dist/log_scale"])).sum(-1)
    raise KeyError(model_cfg["decoder_dist"])


def component_log_probs(p: Params, z: Tensor) -> Tensor:
    """jax.vmap(self.components.log_prob)(z): [..., k] -> [..., C] with MultivariateNormalDiag(mu, exp(log_scale)) (vade.py:55-57)"""
    mu, ls = p["vade/mu"], p["vade/log_scale"]
    d = (z.unsqueeze(-2) - mu) * torch.exp(-ls)
    return (-0.5 * d * d - ls - 0.5 * LOG_2PI).sum(-1)


def log_pi(p: Params) -> Tensor:
    return torch.log_softmax(p["vade/logits"], -1)          # distrax.Categorical(logits).logits is normalised (see header)


def diag_log_prob(z: Tensor, loc: Tensor, scale: Tensor) -> Tensor:
    d = (z - loc) / scale
    return (-0.5 * d * d - torch.log(scale) - 0.5 * LOG_2PI).sum(-1)


# ---- vade.py:96-150 --------------------------------------------------------------------------------------------------------
def elbo(p: Params, model_cfg: dict, x: Tensor, eps: Tensor) -> Tensor:
    """VADE.elbo (vade.py:117-150) as written: responsibilities gamma and the five terms; eps [B, k] is the draw behind
    posterior.sample(seed=hk.next_rng_key())"""
    loc, scale = encoder_params(p, model_cfg, x)
    z = loc + scale * eps
    log_p_x_given_z = decoder_log_prob(p, model_cfg, z, x)
    log_p_z_given_c = component_log_probs(p, z)
    lp = log_pi(p)
    unnorm = log_p_z_given_c + lp.unsqueeze(0)
    log_q_c = torch.log_softmax(unnorm, -1)
    log_q_z = diag_log_prob(z, loc, scale)
    gamma = torch.exp(log_q_c)
    return (log_p_x_given_z + (log_p_z_given_c * gamma).sum(-1) + (lp.unsqueeze(0) * gamma).sum(-1) - log_q_z
            - (log_q_c * gamma).sum(-1))


def elbo_marginal_form(p: Params, model_cfg: dict, x: Tensor, eps: Tensor) -> Tensor:
    """the same number: log p(x | z) + log sum_c pi_c p(z | c) - log q(z | x)  (sum_c gamma_c (s_c - log gamma_c) = logsumexp s)"""
    loc, scale = encoder_params(p, model_cfg, x)
    z = loc + scale * eps
    return (decoder_log_prob(p, model_cfg, z, x) + torch.logsumexp(component_log_probs(p, z) + log_pi(p), -1)
            - diag_log_prob(z, loc, scale))


def predict_cluster(p: Params, model_cfg: dict, x: Tensor, eps: Tensor) -> Tensor:
    """VADE.predict_cluster (vade.py:96-115): eps [S, B, k] -> q(c | x) [B, C] = mean over samples of softmax(log p(z|c) + log pi)"""
    loc, scale = encoder_params(p, model_cfg, x)
    z = loc.unsqueeze(0) + scale.unsqueeze(0) * eps
    h = component_log_probs(p, z) + log_pi(p)
    return torch.softmax(h, -1).mean(0)


def pretrain_loss(p: Params, model_cfg: dict, x: Tensor) -> Tensor:
    """pretrain_loss_fn of train_vade.py:45-49: z = encoder(x).mean(); loss = -mean decoder(z).log_prob(x)"""
    loc, _ = encoder_params(p, model_cfg, x)
    return -decoder_log_prob(p, model_cfg, loc, x).mean()


def vade_loss(p: Params, model_cfg: dict, x: Tensor, eps: Tensor) -> Tensor:
    """loss_fn of train_vade.py:51-55"""
    return -elbo(p, model_cfg, x, eps).mean()


# ---- PosteriorMatchingVADE (vade.py:153-265) --------------------------------------------------------------------------------
def partial_posterior_log_prob(p: Params, model_cfg: dict, x: Tensor, b: Tensor, z: Tensor) -> Tensor:
    x_o_b = torch.cat([x * b, b], dim=-1)
    kind = model_cfg.get("partial_encoder_net", model_cfg["encoder_net"])
    ncfg = model_cfg.get("partial_encoder_net_config", model_cfg.get("encoder_net_config")) or {}
    feat = O._net(p, kind, ncfg, "partial_encoder_net", x_o_b)
    k = model_cfg["latent_dim"]
    dist = model_cfg.get("partial_posterior_dist", "TriLGaussian")
    dcfg = dict(model_cfg.get("partial_posterior_dist_config") or {})
    B = x.shape[0]
    if dist == "AutoregressiveGMM":
        return O.autoregressive_gmm_log_prob_batched(p, "partial_posterior_dist", feat.reshape(B, -1), z, k,
                                                     dcfg.get("num_components", 10), dcfg.get("residual_blocks", 2))
    if dist == "TriLGaussian":
        loc, tril = O.tril_gaussian_params(p, "partial_posterior_dist", feat, k)
        return O.mvn_tril_log_prob(z, loc, tril)
    if dist == "DiagonalGaussian":
        loc, scale = O.diagonal_gaussian_params(p, "partial_posterior_dist", feat, k)
        return diag_log_prob(z, loc, scale)
    raise KeyError(dist)


def posterior_matching_ll(p: Params, model_cfg: dict, x: Tensor, b: Tensor, eps: Tensor) -> Tensor:
    """PosteriorMatchingVADE.posterior_matching_ll (vade.py:247-265): z ~ q(z | x) (no gradient), log q(z | x_o)"""
    loc, scale = encoder_params(p, model_cfg, x)
    z = (loc + scale * eps).detach()
    return partial_posterior_log_prob(p, model_cfg, x, b, z)


def partial_predict_cluster(p: Params, model_cfg: dict, x: Tensor, b: Tensor, noise: Dict[str, Tensor]) -> Tensor:
    """PosteriorMatchingVADE.partial_predict_cluster (vade.py:225-245): noise {"eps" [B,S,k], "gumbel" [B,S,k,nc]}, the rows
    b*S + s of the flattened draws are the S samples of example b -> q(c | x_o) [B, C]"""
    B, S, k = noise["eps"].shape[0], noise["eps"].shape[1], model_cfg["latent_dim"]
    cfg = dict(model_cfg, posterior_dist="DiagonalGaussian")
    kind, dcfg, pfeats, _ = O._partial_posterior(p, dict(cfg, partial_posterior_dist=model_cfg.get("partial_posterior_dist",
                                                                                                  "TriLGaussian")), x, b)
    z, _ = O._sample_partial_posterior(p, kind, dcfg, pfeats, k, noise, S)
    h = component_log_probs(p, z) + log_pi(p)
    return torch.softmax(h, -1).reshape(B, S, -1).mean(1)


def pm_vade_loss(p: Params, model_cfg: dict, x: Tensor, b: Tensor, eps: Tensor) -> Tensor:
    """loss_fn of train_pm_vade.py:40-43"""
    return -posterior_matching_ll(p, model_cfg, x, b, eps).mean()


# ---- optimizer: optax.chain(scale_by_adam(**adam), scale_by_schedule(exponential_decay), scale(-1)) / optax.adam(lr) ----------
def adam_update(p: Params, g: Params, m: Params, v: Params, count: int, lr: float, b1=0.9, b2=0.999, eps=1e-8,
                trainable=None) -> None:
    """train_vade.py:72-74 (optax.adam(pretrain_lr)), :128-133 and train_pm_vade.py:52-57 (no weight decay); `trainable`
    restricts the update to the names it accepts (train_pm_vade.py:59-60: "partial_" in module_name)"""
    t = count + 1
    for n in p:
        if trainable is not None and not trainable(n):
            continue
        m[n] = b1 * m[n] + (1 - b1) * g[n]
        v[n] = b2 * v[n] + (1 - b2) * g[n] * g[n]
        mh, vh = m[n] / (1 - b1 ** t), v[n] / (1 - b2 ** t)
        p[n] = p[n] - lr * mh / (torch.sqrt(vh) + eps)


def lr_value(schedule: dict, count: int) -> float:
    return schedule["init_value"] * schedule["decay_rate"] ** (count / schedule["transition_steps"])


# ---- clustering.py:14-37 ------------------------------------------------------------------------------------------------------
def clustering_accuracy(y_true, y_pred) -> float:
    """max over assignments of clusters to labels of the accuracy (linear sum assignment on the confusion matrix)"""
    from scipy.optimize import linear_sum_assignment

    y_true, y_pred = np.asarray(y_true).astype(int), np.asarray(y_pred).astype(int)
    labels = np.unique(np.concatenate([y_true, y_pred]))
    idx = {v: i for i, v in enumerate(labels)}
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    for t, q in zip(y_true, y_pred):
        cm[idx[t], idx[q]] += 1
    r, c = linear_sum_assignment(-cm + cm.max())
    return float(cm[r, c].sum()) / float(cm.sum())
