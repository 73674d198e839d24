"""CPU oracle for the Posterior-Matching VDVAE training step.  TEST INFRASTRUCTURE ONLY.

CPU restatement (torch on the CPU, float64 by default) of ``PosteriorMatchingVDVAE.__call__``
(posterior_matching/models/vdvae.py:76-94), its ``Encoder`` (:302-348), ``Block`` (:263-299),
``PosteriorMatchingDecoderBlock.{get_inputs,forward_posterior,sample_posterior}`` (:662-687, :532-571),
``PosteriorMatchingDecoder.forward_posterior`` (:757-824), the discretised logistic mixture
(:351-476) and the loss / optimizer of ``train_pm_vdvae.py:109-154``.  Imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg - never by the product package.

PARITY UNPINNED (see oracle/pm_vae_oracle.py's header).  Pinned by the self-derived known-answer
tests in tests/test_oracle_kat.py (SURVEY.md 8c viii, ix, xi): the discretised logistic sums to one
over 0..255, a freshly initialised prior block gives loc 0 / scale softplus(0)+1e-5 and h = 0, the
NEAREST 3->7 index map is [0,0,1,1,1,2,2], KL closed forms against torch.distributions.

Third-party semantics restated: jax.nn.gelu(approximate=True) (tanh form),
hk.AvgPool VALID, jax.image.resize NEAREST (floor((i + 0.5) * in / out)), TFP MultivariateNormalDiag /
TriL KL, tfb.FillScaleTriL, QuantizedDistribution(Logistic shifted by -0.5) evaluated through
log-cdf / log-survival differences.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .pm_vae_oracle import conv2d, fill_scale_tril, softplus

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def gelu(x: Tensor) -> Tensor:
    """jax.nn.gelu default (approximate=True): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))"""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def parse_layer_string(s: str) -> List[Tuple[int, Optional[int]]]:
    """vdvae.py:213-229"""
    layers: List[Tuple[int, Optional[int]]] = []
    for ss in s.split(","):
        if "x" in ss:
            res, num = ss.split("x")
            layers += [(int(res), None) for _ in range(int(num))]
        elif "m" in ss:
            res, mixin = [int(a) for a in ss.split("m")]
            layers.append((res, mixin))
        elif "d" in ss:
            res, down = [int(a) for a in ss.split("d")]
            layers.append((res, down))
        else:
            layers.append((int(ss), None))
    return layers


def avg_pool(x: Tensor, k: int) -> Tensor:
    """hk.AvgPool(k, k, 'VALID') on NHWC: trailing rows / columns that do not fill a window are dropped."""
    return F.avg_pool2d(x.permute(0, 3, 1, 2), k, k).permute(0, 2, 3, 1)


def nearest_index(out_size: int, in_size: int) -> List[int]:
    """jax.image.resize(..., NEAREST): source index floor((i + 0.5) * in / out)"""
    return [min(int(math.floor((i + 0.5) * in_size / out_size)), in_size - 1) for i in range(out_size)]


def resize_nearest(x: Tensor, out_hw: Tuple[int, int]) -> Tensor:
    iy = torch.tensor(nearest_index(out_hw[0], x.shape[1]))
    ix = torch.tensor(nearest_index(out_hw[1], x.shape[2]))
    return x[:, iy][:, :, ix]


def conv3x3(p: Params, name: str, x: Tensor) -> Tensor:
    """get_3x3 (vdvae.py:189-190): padding ((1,1),(1,1)) = SAME for 3x3 stride 1"""
    return conv2d(x, p[f"{name}/w"], p[f"{name}/b"], 1, "SAME")


def conv1x1(p: Params, name: str, x: Tensor) -> Tensor:
    return conv2d(x, p[f"{name}/w"], p[f"{name}/b"], 1, "VALID")


def block(p: Params, name: str, x: Tensor, use_3x3: bool, residual: bool, down_rate: Optional[int] = None) -> Tensor:
    """Block.__call__ (vdvae.py:282-299)"""
    mid = conv3x3 if use_3x3 else conv1x1
    h = conv1x1(p, f"{name}/c1", gelu(x))
    h = mid(p, f"{name}/c2", gelu(h))
    h = mid(p, f"{name}/c3", gelu(h))
    h = conv1x1(p, f"{name}/c4", gelu(h))
    out = x + h if residual else h
    if down_rate is not None:
        out = avg_pool(out, down_rate)
    return out


def encoder(p: Params, prefix: str, x: Tensor, cfg: dict) -> Dict[int, Tensor]:
    """Encoder.__call__ (vdvae.py:317-348); custom widths / channel padding are not restated (the BASELINE
    configs leave custom_width_string = None)."""
    h = conv3x3(p, f"{prefix}/stem", x)
    acts = {h.shape[1]: h}
    for i, (res, down) in enumerate(parse_layer_string(cfg["encoder_blocks"])):
        h = block(p, f"{prefix}/block_{i}", h, res > 2, True, down)
        acts[h.shape[1]] = h
    return acts


def mvn_diag_kl(loc_a, scale_a, loc_b, scale_b) -> Tensor:
    """KL(N(a) || N(b)) of diagonal Gaussians, summed over the last axis (TFP MVN-LinearOperator KL, A6)"""
    return (torch.log(scale_b) - torch.log(scale_a) + (scale_a ** 2 + (loc_a - loc_b) ** 2) / (2 * scale_b ** 2) - 0.5).sum(-1)


def mvn_diag_tril_kl(loc_a, scale_a, loc_b, tril_b) -> Tensor:
    """KL(N(loc_a, diag(scale_a)^2) || N(loc_b, L L^T)): 0.5 [ |L^-1 A|_F^2 + |L^-1 (mu_b - mu_a)|^2 - k ]
    + sum log L_ii - sum log a_i  (A6; triangular solve)"""
    k = loc_a.shape[-1]
    A = torch.diag_embed(scale_a)
    LinvA = torch.linalg.solve_triangular(tril_b, A, upper=False)
    d = torch.linalg.solve_triangular(tril_b, (loc_b - loc_a).unsqueeze(-1), upper=False).squeeze(-1)
    return 0.5 * ((LinvA ** 2).sum((-2, -1)) + (d ** 2).sum(-1) - k) + \
        torch.log(torch.diagonal(tril_b, dim1=-2, dim2=-1)).sum(-1) - torch.log(scale_a).sum(-1)


def decoder_block(p: Params, name: str, xs: Dict[int, Tensor], acts: Tensor, masked_acts: Tensor, res: int,
                  mixin: Optional[int], Z: int, eps: Tensor):
    """PosteriorMatchingDecoderBlock.forward_posterior (vdvae.py:673-687) -> (z, kl [B], pm_kl [B]); updates xs."""
    W = acts.shape[-1]
    x = xs[res] if res in xs else torch.zeros_like(acts)                       # get_inputs (:662-671)
    if x.shape[0] != acts.shape[0]:
        x = x.expand(acts.shape[0], -1, -1, -1)
    if mixin is not None:
        x = x + resize_nearest(xs[mixin][..., :W], (res, res))
    use_3x3 = res > 2
    pp = block(p, f"{name}/posterior", torch.cat([x, acts], -1), use_3x3, False)
    post_loc, post_raw = pp[..., :Z], pp[..., Z:]
    mp = block(p, f"{name}/masked_posterior", torch.cat([x.detach(), masked_acts], -1), use_3x3, False)
    m_loc, m_tril = mp[..., :Z], fill_scale_tril(mp[..., Z:])
    pr = block(p, f"{name}/prior", x, use_3x3, False)
    h = pr[..., 2 * Z:]
    prior_loc, prior_raw = pr[..., :Z], pr[..., Z:2 * Z]
    post_scale = softplus(post_raw) + 1e-5
    prior_scale = softplus(prior_raw) + 1e-5
    x = x + h
    z = post_loc + post_scale * eps                                              # posterior.sample
    B = x.shape[0]
    kl = mvn_diag_kl(post_loc, post_scale, prior_loc, prior_scale).reshape(B, -1).sum(1)       # tfd.Independent: sum over H, W
    pm_kl = mvn_diag_tril_kl(post_loc.detach(), post_scale.detach(), m_loc, m_tril).reshape(B, -1).sum(1)
    x = x + conv1x1(p, f"{name}/z_proj", z)
    x = block(p, f"{name}/resnet", x, use_3x3, True)
    xs[res] = x
    return z, kl, pm_kl


def _log_sigmoid(x: Tensor) -> Tensor:
    return -softplus(-x)


def logistic_mixture_log_prob(params: Tensor, value: Tensor, num_mixtures: int, low: float = 0.0, high: float = 255.0,
                              independent: bool = True) -> Tensor:
    """LogisticMixture.__call__ + _LogisticMixtureDist.log_prob for num_channels = 1 (vdvae.py:351-394,449-476).
    params [B,H,W,3*num_mixtures] -> reshaped [.., num_mixtures, 3] = (logit, loc, raw scale) per component."""
    B, H, W, _ = params.shape
    pr = params.reshape(B, H, W, num_mixtures, 3)
    logits, locs, scales = pr[..., 0], pr[..., 1], softplus(pr[..., 2]) + math.exp(-7.0)
    locs = low + 0.5 * (high - low) * (locs + 1.0)
    scales = scales * 0.5 * (high - low)
    y = value.reshape(B, H, W, 1).clamp(low, high)                      # one channel: broadcast against the components
    # QuantizedDistribution(Logistic shifted by -0.5): P(Y = y) = F(y + .5) - F(y - .5), F(low - .5) := 0, F(high + .5) := 1
    up = (y + 0.5 - locs) / scales
    dn = (y - 0.5 - locs) / scales
    logcdf_y, logsf_y = _log_sigmoid(up), _log_sigmoid(-up)
    logcdf_ym1, logsf_ym1 = _log_sigmoid(dn), _log_sigmoid(-dn)
    ninf = torch.full_like(up, -float("inf"))
    zero = torch.zeros_like(up)
    logcdf_y = torch.where(y >= high, zero, logcdf_y)
    logsf_y = torch.where(y >= high, ninf, logsf_y)
    logcdf_ym1 = torch.where(y <= low, ninf, logcdf_ym1)
    logsf_ym1 = torch.where(y <= low, zero, logsf_ym1)
    use_sf = logsf_y < logcdf_y                                          # TFP: difference of whichever pair is smaller
    big = torch.where(use_sf, logsf_ym1, logcdf_y)
    small = torch.where(use_sf, logsf_y, logcdf_ym1)
    comp = big + torch.log1p(-torch.exp(torch.clamp(small - big, max=0.0)))   # logsubexp
    lp = torch.logsumexp(torch.log_softmax(logits, -1) + comp, dim=-1)   # MixtureSameFamily
    return lp.reshape(B, -1).sum(1) if independent else lp


def logistic_mixture_mean(params: Tensor, num_mixtures: int, low: float = 0.0, high: float = 255.0) -> Tensor:
    """_LogisticMixtureDist.mean for one channel (vdvae.py:396-435)"""
    B, H, W, _ = params.shape
    pr = params.reshape(B, H, W, num_mixtures, 3)
    w = torch.softmax(pr[..., 0], -1)
    loc = (pr[..., 1] * w).sum(-1, keepdim=True).clamp(-1.0, 1.0)
    return torch.round(low + 0.5 * (high - low) * (loc + 1.0))


def vdvae_forward(p: Params, cfg: dict, x: Tensor, b: Tensor, eps: Sequence[Tensor]) -> Dict[str, Tensor]:
    """PosteriorMatchingVDVAE.__call__ (vdvae.py:76-94).  eps: one N(0,1) draw [B,res,res,Z] per decoder block."""
    Z, W = cfg.get("latent_dim", 16), cfg.get("width", 128)
    size = cfg["image_shape"][0]
    xn = x / 127.5 - 1.0
    acts = encoder(p, "encoder", xn, cfg)
    macts = encoder(p, "masked_encoder", torch.cat([xn * b, b], -1), cfg)
    blocks = parse_layer_string(cfg["decoder_blocks"])
    resolutions = sorted({r for r, _ in blocks})
    xs = {r: p[f"decoder/x_bias_{r}"] for r in resolutions if r <= cfg.get("no_bias_above", 64)}
    kl = pm_kl = 0.0
    zs = []
    for i, (res, mixin) in enumerate(blocks):
        z, k1, k2 = decoder_block(p, f"decoder/block_{i}", xs, acts[res], macts[res], res, mixin, Z, eps[i])
        kl, pm_kl = kl + k1, pm_kl + k2
        zs.append(z)
    px_z = xs[size] * p["decoder/gain"] + p["decoder/bias"]                  # final_fn (:812)
    params = conv1x1(p, "decoder/out_net", px_z)
    nm = cfg.get("num_mixtures", 10)
    return {"reconstruction_ll": logistic_mixture_log_prob(params, x, nm), "kl": kl, "pm_kl": pm_kl,
            "reconstruction": logistic_mixture_mean(params, nm), "z": zs}


def vdvae_loss(p: Params, cfg: dict, x: Tensor, b: Tensor, eps: Sequence[Tensor]):
    """loss_fn of train_pm_vdvae.py:109-120 -> (loss, aux)"""
    out = vdvae_forward(p, cfg["model"], x, b, eps)
    elbo = (out["reconstruction_ll"] - out["kl"]).mean()
    loss = -elbo + out["pm_kl"].mean()
    aux = {"reconstruction_ll": out["reconstruction_ll"].mean(), "kl": out["kl"].mean(), "pm_kl": out["pm_kl"].mean(),
           "bpd": -elbo / (math.prod(cfg["model"]["image_shape"]) * math.log(2.0))}
    return loss, aux, out


def optimizer_update(p: Params, g: Params, m: Params, v: Params, ema: Optional[Params], count: int, cfg: dict) -> bool:
    """train_pm_vdvae.py:128-154: clip_by_global_norm -> scale_by_adam -> add_decayed_weights -> lr schedule (constant
    config.lr, or optax.linear_schedule(0, config.lr, warm_up) when config.warm_up > 0: lr * clip(count / warm_up, 0, 1)) ->
    scale(-1); Trainer(skip_nonfinite_updates=True, ema_rate).  Returns False when the step was skipped."""
    gn = math.sqrt(sum(float((t.double() ** 2).sum()) for t in g.values()))
    if not math.isfinite(gn):
        return False                                                    # parameters, moments and EMA stay untouched
    clip = cfg.get("gradient_clip", 200.0)
    scale = 1.0 if gn < clip else clip / gn                             # optax: where(g_norm < max_norm, g, g / g_norm * max_norm)
    adam = cfg.get("adam") or {}
    b1, b2, eps = adam.get("b1", 0.9), adam.get("b2", 0.999), adam.get("eps", 1e-8)
    wd, lr, t = cfg.get("weight_decay", 0.0), cfg["lr"], count + 1
    if cfg.get("warm_up", 0) > 0:
        lr = lr * min(max(count / cfg["warm_up"], 0.0), 1.0)
    for name in p:
        gg = g[name] * scale
        m[name].mul_(b1).add_(gg, alpha=1 - b1)
        v[name].mul_(b2).addcmul_(gg, gg, value=1 - b2)
        u = (m[name] / (1 - b1 ** t)) / (torch.sqrt(v[name] / (1 - b2 ** t)) + eps)
        if wd != 0.0 and p[name].ndim != 1:
            u = u + wd * p[name]
        p[name].add_(u, alpha=-lr)
        if ema is not None:
            r = cfg.get("ema_rate", 0.999)
            ema[name].mul_(r).add_(p[name], alpha=1 - r)
    return True


def train_step(p: Params, m: Params, v: Params, ema: Optional[Params], cfg: dict, x: Tensor, b: Tensor,
               eps: Sequence[Tensor], step: int):
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in p.items()}
    loss, aux, _ = vdvae_loss(leaves, cfg, x, b, eps)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(leaves, grads)}
    optimizer_update(p, g, m, v, ema, step, cfg)
    return loss.detach(), {k: a.detach() for k, a in aux.items()}, g


# ----------------------------------------------------------------------------------------------
# parameter specification / haiku-style init
# ----------------------------------------------------------------------------------------------
def _block_shapes(s, name, cin, mid, cout, use_3x3):
    k = 3 if use_3x3 else 1
    for cn, ks, ci, co in (("c1", 1, cin, mid), ("c2", k, mid, mid), ("c3", k, mid, mid), ("c4", 1, mid, cout)):
        s[f"{name}/{cn}/w"], s[f"{name}/{cn}/b"] = (ks, ks, ci, co), (co,)


def param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    W, Z = cfg.get("width", 128), cfg.get("latent_dim", 16)
    mid = int(W * cfg.get("bottleneck_multiple", 0.25))
    C = cfg["image_shape"][-1]
    s: Dict[str, Tuple[int, ...]] = {}
    for prefix, cin in (("encoder", C), ("masked_encoder", 2 * C)):
        s[f"{prefix}/stem/w"], s[f"{prefix}/stem/b"] = (3, 3, cin, W), (W,)
        for i, (res, _) in enumerate(parse_layer_string(cfg["encoder_blocks"])):
            _block_shapes(s, f"{prefix}/block_{i}", W, mid, W, res > 2)
    blocks = parse_layer_string(cfg["decoder_blocks"])
    for i, (res, _) in enumerate(blocks):
        n = f"decoder/block_{i}"
        _block_shapes(s, f"{n}/posterior", 2 * W, mid, 2 * Z, res > 2)
        _block_shapes(s, f"{n}/masked_posterior", 2 * W, mid, Z + Z * (Z + 1) // 2, res > 2)
        _block_shapes(s, f"{n}/prior", W, mid, 2 * Z + W, res > 2)
        s[f"{n}/z_proj/w"], s[f"{n}/z_proj/b"] = (1, 1, Z, W), (W,)
        _block_shapes(s, f"{n}/resnet", W, mid, W, res > 2)
    for r in sorted({r for r, _ in blocks}):
        if r <= cfg.get("no_bias_above", 64):
            s[f"decoder/x_bias_{r}"] = (1, r, r, W)
    nm = cfg.get("num_mixtures", 10)
    s["decoder/out_net/w"], s["decoder/out_net/b"] = (1, 1, W, nm * (2 * C + C * (C - 1) // 2 + 1)), (nm * (2 * C + C * (C - 1) // 2 + 1),)
    s["decoder/gain"], s["decoder/bias"] = (1, 1, 1, W), (1, 1, 1, W)
    return s


def init_params(cfg: dict, seed: int = 1, dtype=torch.float64) -> Params:
    """haiku defaults + the special initialisers of vdvae.py:193-205: c4 of every residual block and z_proj
    TruncatedNormal(stddev/sqrt(N)) with N = number of blocks of that network; c4 of the prior block zeros;
    x_bias / bias zeros; gain ones."""
    from scipy.special import ndtr, ndtri

    def _tn(size, random_state):      # scipy.stats.truncnorm.rvs(-2, 2, ...): same uniform draws, inverse CDF by ndtri (1000x faster)
        lo, hi = ndtr(-2.0), ndtr(2.0)
        return ndtri(lo + random_state.uniform(size=size) * (hi - lo))

    rng = np.random.default_rng(seed)
    n_enc = len(parse_layer_string(cfg["encoder_blocks"]))
    n_dec = len(parse_layer_string(cfg["decoder_blocks"]))
    out: Params = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("/w"):
            std = 1.0 / math.sqrt(int(np.prod(shp[:-1])))
            if name.endswith("/prior/c4/w"):
                arr = np.zeros(shp)
            else:
                if "encoder/block_" in name and name.endswith("/c4/w"):
                    std *= math.sqrt(1.0 / n_enc)
                elif name.endswith("/resnet/c4/w") or name.endswith("/z_proj/w"):
                    std *= math.sqrt(1.0 / n_dec)
                arr = _tn(shp, rng) * std
        elif name == "decoder/gain":
            arr = np.ones(shp)
        else:
            arr = np.zeros(shp)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out


# ----------------------------------------------------------------------------------------------
# imputation from the partially observed posterior and its PSNR (vdvae.py:161-186, 573-590, 689-703;
# eval_pm_vdvae_imputation.py:116-130)
# ----------------------------------------------------------------------------------------------
def vdvae_impute(p: Params, cfg: dict, x: Tensor, b: Tensor, eps: Sequence[Sequence[Tensor]]) -> Tensor:
    """PosteriorMatchingVDVAE.impute -> [B, S, H, W, C]; eps[s][i]: noise of sample s, decoder block i."""
    Z, size = cfg.get("latent_dim", 16), cfg["image_shape"][0]
    xn = x / 127.5 - 1.0
    macts = encoder(p, "masked_encoder", torch.cat([xn * b, b], -1), cfg)
    blocks = parse_layer_string(cfg["decoder_blocks"])
    resolutions = sorted({r for r, _ in blocks})
    outs = []
    for es in eps:
        xs = {r: p[f"decoder/x_bias_{r}"] for r in resolutions if r <= cfg.get("no_bias_above", 64)}
        for i, (res, mixin) in enumerate(blocks):
            ma = macts[res]
            W = ma.shape[-1]
            xx = xs[res] if res in xs else torch.zeros_like(ma)
            if xx.shape[0] != ma.shape[0]:
                xx = xx.expand(ma.shape[0], -1, -1, -1)
            if mixin is not None:
                xx = xx + resize_nearest(xs[mixin][..., :W], (res, res))
            n = f"decoder/block_{i}"
            mp = block(p, f"{n}/masked_posterior", torch.cat([xx, ma], -1), res > 2, False)
            pr = block(p, f"{n}/prior", xx, res > 2, False)
            xx = xx + pr[..., 2 * Z:]
            z = mp[..., :Z] + (fill_scale_tril(mp[..., Z:]) @ es[i].unsqueeze(-1)).squeeze(-1)
            xx = xx + conv1x1(p, f"{n}/z_proj", z)
            xs[res] = block(p, f"{n}/resnet", xx, res > 2, True)
        params = conv1x1(p, "decoder/out_net", xs[size] * p["decoder/gain"] + p["decoder/bias"])
        mean = logistic_mixture_mean(params, cfg.get("num_mixtures", 10))
        outs.append(torch.where(b == 1, x, mean))
    return torch.stack(outs, dim=1)


def _diag_log_prob(z: Tensor, loc: Tensor, raw: Tensor) -> Tensor:
    """tfd.Independent(MultivariateNormalDiag(loc, softplus(raw) + 1e-5)).log_prob(z): sum over latent dims and positions"""
    scale = softplus(raw) + 1e-5
    lp = -0.5 * ((z - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2 * math.pi)
    return lp.reshape(z.shape[0], -1).sum(1)


def _tril_log_prob(z: Tensor, loc: Tensor, tril: Tensor) -> Tensor:
    """tfd.Independent(MultivariateNormalTriL(loc, tril)).log_prob(z) summed over positions"""
    k = z.shape[-1]
    d = torch.linalg.solve_triangular(tril, (z - loc).unsqueeze(-1), upper=False).squeeze(-1)
    lp = -0.5 * (d ** 2).sum(-1) - torch.log(torch.diagonal(tril, dim1=-2, dim2=-1)).sum(-1) - 0.5 * k * math.log(2 * math.pi)
    return lp.reshape(z.shape[0], -1).sum(1)


def vdvae_is_log_probs(p: Params, cfg: dict, x: Tensor, b: Tensor, eps: Sequence[Sequence[Tensor]],
                       eps_masked: Sequence[Sequence[Tensor]]):
    """PosteriorMatchingVDVAE.is_log_probs (vdvae.py:96-146) with PosteriorMatchingDecoder.forward_lls (:844-855),
    PosteriorMatchingDecoderBlock.forward_lls (:725-754) and .sample_lls (:609-660).  Two decoder states are carried:
    x (driven by the full posterior q(z | x)) and masked_x (driven by the masked posterior q(z | x_o)); the prior
    block is evaluated on each.  eps[s][i] / eps_masked[s][i]: the N(0,1) draws of sample s, decoder block i, behind
    posterior.sample / masked_posterior.sample.  -> (log p(x) [B], log p(x_u | x_o) [B])."""
    Z, size = cfg.get("latent_dim", 16), cfg["image_shape"][0]
    nm = cfg.get("num_mixtures", 10)
    xn = x / 127.5 - 1.0
    acts = encoder(p, "encoder", xn, cfg)
    macts = encoder(p, "masked_encoder", torch.cat([xn * b, b], -1), cfg)
    blocks = parse_layer_string(cfg["decoder_blocks"])
    resolutions = sorted({r for r, _ in blocks})
    pxs, pxos = [], []
    for es, ems in zip(eps, eps_masked):
        xs = {r: p[f"decoder/x_bias_{r}"] for r in resolutions if r <= cfg.get("no_bias_above", 64)}
        mxs = dict(xs)
        pz = qzx = mpz = mqzx = 0.0
        for i, (res, mixin) in enumerate(blocks):
            a, ma = acts[res], macts[res]
            W = a.shape[-1]
            n = f"decoder/block_{i}"
            use3 = res > 2

            def start(state):
                v = state[res] if res in state else torch.zeros_like(a)
                if v.shape[0] != a.shape[0]:
                    v = v.expand(a.shape[0], -1, -1, -1)
                if mixin is not None:
                    v = v + resize_nearest(state[mixin][..., :W], (res, res))
                return v

            xx, mx = start(xs), start(mxs)
            pp = block(p, f"{n}/posterior", torch.cat([xx, a], -1), use3, False)
            mp = block(p, f"{n}/masked_posterior", torch.cat([mx, ma], -1), use3, False)
            pr = block(p, f"{n}/prior", xx, use3, False)
            mpr = block(p, f"{n}/prior", mx, use3, False)
            xx = xx + pr[..., 2 * Z:]
            mx = mx + mpr[..., 2 * Z:]
            z = pp[..., :Z] + (softplus(pp[..., Z:]) + 1e-5) * es[i]
            m_tril = fill_scale_tril(mp[..., Z:])
            mz = mp[..., :Z] + (m_tril @ ems[i].unsqueeze(-1)).squeeze(-1)
            pz = pz + _diag_log_prob(z, pr[..., :Z], pr[..., Z:2 * Z])
            qzx = qzx + _diag_log_prob(z, pp[..., :Z], pp[..., Z:])
            mpz = mpz + _diag_log_prob(mz, mpr[..., :Z], mpr[..., Z:2 * Z])
            mqzx = mqzx + _tril_log_prob(mz, mp[..., :Z], m_tril)
            xx = xx + conv1x1(p, f"{n}/z_proj", z)
            mx = mx + conv1x1(p, f"{n}/z_proj", mz)
            xs[res] = block(p, f"{n}/resnet", xx, use3, True)
            mxs[res] = block(p, f"{n}/resnet", mx, use3, True)
        prm = conv1x1(p, "decoder/out_net", xs[size] * p["decoder/gain"] + p["decoder/bias"])
        mprm = conv1x1(p, "decoder/out_net", mxs[size] * p["decoder/gain"] + p["decoder/bias"])
        pxz = logistic_mixture_log_prob(prm, x, nm)
        pxoz = (logistic_mixture_log_prob(mprm, x, nm, independent=False).unsqueeze(-1) * b).reshape(x.shape[0], -1).sum(1)
        pxs.append(pxz + pz - qzx)
        pxos.append(pxoz + mpz - mqzx)
    S = len(pxs)
    px = torch.logsumexp(torch.stack(pxs, 0), 0) - math.log(S)
    pxo = torch.logsumexp(torch.stack(pxos, 0), 0) - math.log(S)
    return px, px - pxo


def vdvae_sample(p: Params, cfg: dict, eps: Sequence[Tensor]) -> Tensor:
    """PosteriorMatchingVDVAE.sample (vdvae.py:148-159): forward_prior (:835-842, :705-723, :593-607) then the decoder
    distribution's mean.  eps[i] [N,res,res,Z]: the draw behind prior.sample of decoder block i."""
    Z, size = cfg.get("latent_dim", 16), cfg["image_shape"][0]
    blocks = parse_layer_string(cfg["decoder_blocks"])
    resolutions = sorted({r for r, _ in blocks})
    N = eps[0].shape[0]
    W = cfg.get("width", 128)
    xs = {r: p[f"decoder/x_bias_{r}"].expand(N, -1, -1, -1) for r in resolutions if r <= cfg.get("no_bias_above", 64)}
    for i, (res, mixin) in enumerate(blocks):
        n = f"decoder/block_{i}"
        xx = xs[res] if res in xs else torch.zeros((N, res, res, W), dtype=eps[0].dtype)
        if mixin is not None:
            xx = xx + resize_nearest(xs[mixin][..., :W], (res, res))
        pr = block(p, f"{n}/prior", xx, res > 2, False)
        xx = xx + pr[..., 2 * Z:]
        z = pr[..., :Z] + (softplus(pr[..., Z:2 * Z]) + 1e-5) * eps[i]
        xx = xx + conv1x1(p, f"{n}/z_proj", z)
        xs[res] = block(p, f"{n}/resnet", xx, res > 2, True)
    prm = conv1x1(p, "decoder/out_net", xs[size] * p["decoder/gain"] + p["decoder/bias"])
    return logistic_mixture_mean(prm, cfg.get("num_mixtures", 10))


def imputation_psnr(imputations: Tensor, x: Tensor) -> Tensor:
    err = ((imputations.mean(1) / 255.0 - x / 255.0) ** 2).reshape(x.shape[0], -1).mean(1)
    return -10.0 * torch.log10(err)
