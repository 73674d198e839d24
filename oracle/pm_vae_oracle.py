"""CPU oracle for the Posterior-Matching VAE training step.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch tensors on the CPU, float64 by default) of the
arithmetic of the reference's PM-VAE hot path.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg - never by the product
package ``posterior_matching_amd`` (which fails loudly when its HIP library is missing).

PARITY UNPINNED: the reference (JAX + dm-haiku + TFP + optax + bax) cannot be imported in the
build container (ordinary ModuleNotFoundError for jax/haiku/tfp/optax/bax; no network) and the
reference ships no tests, golden vectors or fixtures.  This restatement is
written from the reference's source text plus the published semantics of the pinned third-party
versions (requirements.txt: jax 0.2.26, dm-haiku 0.0.5, tensorflow-probability 0.15.0,
optax 0.1.0).  It is pinned by the self-derived known-answer tests in ``tests/test_oracle_kat.py``
and by an independent naive-loop numpy implementation (``oracle/naive_numpy.py``).

Every function cites the reference file:line it follows (paths relative to /root/reference).
Nothing here uses torch.distributions or torch's conv_transpose: convolutions are restated as
XLA states them (explicit SAME padding, lhs-dilation, un-flipped kernels) so that the oracle does
not inherit torch's conventions.

Conventions: activations NHWC, conv weights HWIO, transposed-conv weights HW(O)(I), dense
weights [in, out], masks 1 = observed.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# Third-party primitives restated (XLA / haiku / TFP / optax semantics)
# ----------------------------------------------------------------------------------------------
def same_padding(in_size: int, k: int, s: int) -> Tuple[int, int]:
    """XLA padding="SAME": out = ceil(in/s); the extra pixel goes AFTER (Appendix A1)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    lo = total // 2
    return lo, total - lo


def conv_transpose_padding(k: int, s: int, padding: str) -> Tuple[int, int]:
    """jax.lax._conv_transpose_padding (jax 0.2.26), used by lax.conv_transpose (Appendix A2)."""
    if padding == "SAME":
        pad_len = k + s - 2
        pad_a = k - 1 if s > k - 1 else int(math.ceil(pad_len / 2))
    elif padding == "VALID":
        pad_len = k + s - 2 + max(k - s, 0)
        pad_a = k - 1
    else:
        raise ValueError(padding)
    return pad_a, pad_len - pad_a


def conv2d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, padding: str) -> Tensor:
    """hk.Conv2D -> lax.conv_general_dilated, NHWC / HWIO, cross-correlation.

    Mirrors posterior_matching/models/pixel_cnn.py:188-198 (the in-repo copy of haiku's ConvND)
    as called from networks.py:30-35.
    """
    kh, kw, cin, cout = w.shape
    if padding == "SAME":
        (pt, pb), (pl, pr) = same_padding(x.shape[1], kh, stride), same_padding(x.shape[2], kw, stride)
    elif padding == "VALID":
        pt = pb = pl = pr = 0
    else:
        raise ValueError(padding)
    xn = x.permute(0, 3, 1, 2)                       # NHWC -> NCHW for torch's conv
    xn = F.pad(xn, (pl, pr, pt, pb))
    wn = w.permute(3, 2, 0, 1).contiguous()          # HWIO -> OIHW; torch conv2d = cross-correlation
    y = F.conv2d(xn, wn, None, stride=stride)
    y = y.permute(0, 2, 3, 1)
    if b is not None:
        y = y + b
    return y


def conv2d_transpose(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, padding: str) -> Tensor:
    """hk.Conv2DTranspose -> lax.conv_transpose (transpose_kernel=False).

    Mirrors pixel_cnn.py:270-306 as called from networks.py:62-67: weight [kh, kw, Cout, Cin],
    computed as conv_general_dilated(x, w, strides=1, padding=(pad_a, pad_b), lhs_dilation=stride)
    with rhs spec HWOI and NO spatial flip of the kernel.
    """
    kh, kw, cout, cin = w.shape
    pa_h, pb_h = conv_transpose_padding(kh, stride, padding)
    pa_w, pb_w = conv_transpose_padding(kw, stride, padding)
    n, h, wd, c = x.shape
    xn = x.permute(0, 3, 1, 2)
    if stride > 1:                                    # lhs dilation: insert stride-1 zeros
        xd = x.new_zeros((n, c, (h - 1) * stride + 1, (wd - 1) * stride + 1))
        xd[:, :, ::stride, ::stride] = xn
        xn = xd
    xn = F.pad(xn, (pa_w, pb_w, pa_h, pb_h))
    wn = w.permute(2, 3, 0, 1).contiguous()           # HWOI -> OIHW, un-flipped
    y = F.conv2d(xn, wn, None, stride=1).permute(0, 2, 3, 1)
    if b is not None:
        y = y + b
    return y


def leaky_relu(x: Tensor, slope: float = 0.01) -> Tensor:
    """jax.nn.leaky_relu: where(x >= 0, x, slope * x) (networks.py:36,68)."""
    return torch.where(x >= 0, x, slope * x)


def relu(x: Tensor) -> Tensor:
    """jax.nn.relu = max(x, 0), gradient 0 at x == 0 (networks.py:95 default activation)."""
    return torch.where(x > 0, x, torch.zeros_like(x))


def softplus(x: Tensor) -> Tensor:
    """jax.nn.softplus = logaddexp(x, 0)."""
    return torch.logaddexp(x, torch.zeros_like(x))


def linear(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """hk.Linear: y = x @ W[in, out] + b."""
    return x @ w + b


def fill_triangular(v: Tensor) -> Tensor:
    """TFP fill_triangular(v, upper=False): reshape(concat[v[n:], reverse(v)], [n, n]), lower part.

    Example (TFP docstring): [1,2,3,4,5,6] -> [[4,0,0],[6,5,0],[3,2,1]].  Used by
    tfb.FillScaleTriL at distributions.py:111.
    """
    m = v.shape[-1]
    n = int((math.isqrt(8 * m + 1) - 1) // 2)
    assert n * (n + 1) // 2 == m
    cat = torch.cat([v[..., n:], torch.flip(v, dims=[-1])], dim=-1)
    return torch.tril(cat.reshape(*v.shape[:-1], n, n))


def fill_scale_tril(v: Tensor, diag_shift: float = 1e-5) -> Tensor:
    """tfb.FillScaleTriL(): FillTriangular, then diag <- softplus(diag) + 1e-5."""
    l = fill_triangular(v)
    d = softplus(torch.diagonal(l, dim1=-2, dim2=-1)) + diag_shift
    return torch.tril(l, diagonal=-1) + torch.diag_embed(d)


def mvn_tril_kl_to_std_normal(loc: Tensor, scale_tril: Tensor) -> Tensor:
    """KL(N(loc, L L^T) || N(0, I)) (vae.py:130 with the prior of vae.py:55-57)."""
    k = loc.shape[-1]
    logdet = torch.log(torch.diagonal(scale_tril, dim1=-2, dim2=-1)).sum(-1)
    return 0.5 * ((scale_tril ** 2).sum((-2, -1)) + (loc ** 2).sum(-1) - k) - logdet


def mvn_tril_log_prob(z: Tensor, loc: Tensor, scale_tril: Tensor) -> Tensor:
    """tfd.MultivariateNormalTriL.log_prob: -0.5 |L^-1 (z - loc)|^2 - sum log L_ii - k/2 log 2pi."""
    k = loc.shape[-1]
    diff = (z - loc).unsqueeze(-1)
    sol = torch.linalg.solve_triangular(scale_tril, diff, upper=False).squeeze(-1)
    logdet = torch.log(torch.diagonal(scale_tril, dim1=-2, dim2=-1)).sum(-1)
    return -0.5 * (sol ** 2).sum(-1) - logdet - 0.5 * k * LOG_2PI


def bernoulli_log_prob(logits: Tensor, x: Tensor) -> Tensor:
    """tfd.Bernoulli(logits).log_prob(x) with real-valued x (distributions.py:24-25)."""
    return x * (-softplus(-logits)) + (1.0 - x) * (-softplus(logits))


def normal_log_prob(x: Tensor, loc: Tensor, scale: Tensor) -> Tensor:
    return -0.5 * ((x - loc) / scale) ** 2 - torch.log(scale) - 0.5 * LOG_2PI


# ----------------------------------------------------------------------------------------------
# Networks (posterior_matching/models/networks.py)
# ----------------------------------------------------------------------------------------------
def conv_encoder(p: Params, prefix: str, x: Tensor, conv_layers: Sequence[Tuple[int, int, int]]) -> Tensor:
    """ConvEncoder.__call__ (networks.py:24-38): SAME convs, last VALID, leaky_relu after each."""
    h = x
    for i, (_, _, stride) in enumerate(conv_layers):
        pad = "VALID" if i == len(conv_layers) - 1 else "SAME"
        h = conv2d(h, p[f"{prefix}/conv_{i}/w"], p[f"{prefix}/conv_{i}/b"], stride, pad)
        h = leaky_relu(h)
    return h


def conv_decoder(p: Params, prefix: str, z: Tensor, conv_layers: Sequence[Tuple[int, int, int]]) -> Tensor:
    """ConvDecoder.__call__ (networks.py:56-72): z -> [B,1,1,Z]; first VALID, leaky_relu after ALL."""
    h = z[:, None, None, :]
    for i, (_, _, stride) in enumerate(conv_layers):
        pad = "VALID" if i == 0 else "SAME"
        h = conv2d_transpose(h, p[f"{prefix}/conv_t_{i}/w"], p[f"{prefix}/conv_t_{i}/b"], stride, pad)
        h = leaky_relu(h)
    return h


def layer_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """hk.LayerNorm(-1, create_scale=False, create_offset=False) (networks.py:118)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def residual_mlp(p: Params, prefix: str, x: Tensor, residual_blocks: int = 2, layer_norm_: bool = False,
                 activate_final: bool = True, dropout_masks: Optional[List[Tensor]] = None) -> Tensor:
    """ResidualMLP.__call__ (networks.py:111-135), activation relu.

    ``dropout_masks`` (one per block, already scaled by 1/(1-rate)) replaces hk.dropout so that
    parity needs no RNG-stream agreement.
    """
    h = linear(x, p[f"{prefix}/linear_0/w"], p[f"{prefix}/linear_0/b"])
    if layer_norm_:
        h = layer_norm(h)
    for k in range(residual_blocks):
        r = relu(h)
        r = linear(r, p[f"{prefix}/block_{k}/linear_0/w"], p[f"{prefix}/block_{k}/linear_0/b"])
        if layer_norm_:
            r = layer_norm(r)
        r = relu(r)
        if dropout_masks is not None:
            r = r * dropout_masks[k]
        r = linear(r, p[f"{prefix}/block_{k}/linear_1/w"], p[f"{prefix}/block_{k}/linear_1/b"])
        if layer_norm_:
            r = layer_norm(r)
        h = h + r
    if activate_final:
        h = relu(h)
    return h


# ----------------------------------------------------------------------------------------------
# Distribution heads (posterior_matching/models/distributions.py)
# ----------------------------------------------------------------------------------------------
def tril_gaussian_params(p: Params, prefix: str, feats: Tensor, event_size: int) -> Tuple[Tensor, Tensor]:
    """TriLGaussian.__call__ (distributions.py:101-113): Flatten, Linear(k + k(k+1)/2), split."""
    flat = feats.reshape(feats.shape[0], -1)
    prm = linear(flat, p[f"{prefix}/linear/w"], p[f"{prefix}/linear/b"])
    return prm[:, :event_size], fill_scale_tril(prm[:, event_size:])


def diagonal_gaussian_params(p: Params, prefix: str, feats: Tensor, event_size: int) -> Tuple[Tensor, Tensor]:
    """DiagonalGaussian.__call__ (distributions.py:73-84): Flatten -> Linear(2k); loc, scale = softplus(raw) + 1e-5."""
    prm = linear(feats.reshape(feats.shape[0], -1), p[f"{prefix}/linear/w"], p[f"{prefix}/linear/b"])
    return prm[:, :event_size], softplus(prm[:, event_size:]) + 1e-5


def gmm_log_prob_columns(head: Tensor, value: Tensor, event_size: int, num_components: int) -> Tensor:
    """OneDimensionalGMM (distributions.py:124-134) + MixtureSameFamily.log_prob -> [B, event]."""
    prm = head.reshape(head.shape[0], event_size, 3 * num_components)
    logits = prm[..., :num_components]
    means = prm[..., num_components:-num_components]
    scales = softplus(prm[..., -num_components:]) + 1e-5
    comp = normal_log_prob(value[..., None], means, scales)
    return torch.logsumexp(torch.log_softmax(logits, -1) + comp, -1)


def autoregressive_gmm_log_prob(p: Params, prefix: str, context: Tensor, value: Tensor, event_size: int,
                                num_components: int = 10, residual_blocks: int = 2) -> Tensor:
    """_AutoregressiveDistribution.log_prob (distributions.py:152-166), sequential as written.

    For i in 0..k-1: mask = arange(k) < i; net([value*mask, mask, context]) -> GMM; take column i.
    """
    ctx = context.reshape(context.shape[0], -1)
    ar = torch.arange(event_size, dtype=value.dtype)
    total = torch.zeros(value.shape[0], dtype=value.dtype)
    for i in range(event_size):
        mask = (ar < i).to(value.dtype).expand_as(value)
        inp = torch.cat([value * mask, mask, ctx], -1)
        h = residual_mlp(p, f"{prefix}/mlp", inp, residual_blocks)
        head = linear(h, p[f"{prefix}/gmm/linear/w"], p[f"{prefix}/gmm/linear/b"])
        total = total + gmm_log_prob_columns(head, value, event_size, num_components)[:, i]
    return total


def autoregressive_gmm_log_prob_batched(p: Params, prefix: str, context: Tensor, value: Tensor,
                                        event_size: int, num_components: int = 10,
                                        residual_blocks: int = 2) -> Tensor:
    """Same quantity with the k scan steps stacked on the batch axis (KAT vi: must equal the scan)."""
    bsz = value.shape[0]
    ctx = context.reshape(bsz, -1)
    ar = torch.arange(event_size, dtype=value.dtype)
    mask = (ar[None, :] < ar[:, None]).to(value.dtype)                  # [step, k]
    mask = mask[:, None, :].expand(event_size, bsz, event_size)
    inp = torch.cat([value[None] * mask, mask, ctx[None].expand(event_size, bsz, -1)], -1)
    h = residual_mlp(p, f"{prefix}/mlp", inp.reshape(event_size * bsz, -1), residual_blocks)
    head = linear(h, p[f"{prefix}/gmm/linear/w"], p[f"{prefix}/gmm/linear/b"])
    cols = gmm_log_prob_columns(head, value.repeat(event_size, 1), event_size, num_components)
    cols = cols.reshape(event_size, bsz, event_size)
    return torch.diagonal(cols, dim1=0, dim2=2).sum(-1)


# ----------------------------------------------------------------------------------------------
# Model (posterior_matching/models/vae.py) and loss (train_pm_vae.py)
# ----------------------------------------------------------------------------------------------
def _net(p: Params, kind: str, cfg: dict, prefix: str, x: Tensor, dropout_masks=None) -> Tensor:
    if kind == "ConvEncoder":
        return conv_encoder(p, prefix, x, cfg["conv_layers"])
    if kind == "ConvDecoder":
        return conv_decoder(p, prefix, x, cfg["conv_layers"])
    if kind == "ResidualMLP":
        return residual_mlp(p, prefix, x, cfg.get("residual_blocks", 2), cfg.get("layer_norm", False),
                            cfg.get("activate_final", True), dropout_masks)
    raise KeyError(kind)


def pm_vae_forward(p: Params, model_cfg: dict, x: Tensor, b: Tensor, eps: Tensor,
                   dropout_masks: Optional[Dict[str, List[Tensor]]] = None) -> Dict[str, Tensor]:
    """PosteriorMatchingVAE.__call__ (vae.py:120-144) with the config plumbing of from_config
    (vae.py:61-118).  ``eps`` is the explicit N(0,1) draw behind posterior.sample (vae.py:124).
    ``dropout_masks`` (is_training with a ResidualMLP dropout rate, networks.py:114,125): explicit keep masks per
    network name ("encoder_net" / "decoder_net" / "partial_encoder_net"), one [B, hidden] tensor per residual block,
    already scaled by 1 / (1 - rate); None = no dropout (is_training False or rate 0).

    Quirk kept: from_config reads ``partial_posterior_dist[_config]`` only, so a
    config that sets ``masked_posterior_dist`` (configs/pm_vae_gas.py:24-27) silently gets the
    partial posterior = posterior_dist (TriLGaussian).
    """
    k = model_cfg["latent_dim"]
    enc_kind = model_cfg["encoder_net"]
    enc_cfg = model_cfg.get("encoder_net_config") or {}
    penc_kind = model_cfg.get("partial_encoder_net", enc_kind)
    penc_cfg = model_cfg.get("partial_encoder_net_config", enc_cfg) or {}
    dec_kind = model_cfg["decoder_net"]
    dec_cfg = model_cfg.get("decoder_net_config") or {}
    post_kind = model_cfg["posterior_dist"]
    ppost_kind = model_cfg.get("partial_posterior_dist", post_kind)
    ppost_cfg = dict(model_cfg.get("partial_posterior_dist_config", model_cfg.get("posterior_dist_config", {})) or {})
    dm = dropout_masks or {}
    feats = _net(p, enc_kind, enc_cfg, "encoder_net", x, dm.get("encoder_net"))
    if post_kind == "TriLGaussian":
        loc, tril = tril_gaussian_params(p, "posterior_dist", feats, k)
    elif post_kind == "DiagonalGaussian":                                 # distributions.py:58-84: MultivariateNormalDiag
        loc, scale = diagonal_gaussian_params(p, "posterior_dist", feats, k)
        tril = torch.diag_embed(scale)
    else:
        raise KeyError(post_kind)
    z = loc + torch.einsum("bij,bj->bi", tril, eps)                       # vae.py:124

    dec = _net(p, dec_kind, dec_cfg, "decoder_net", z, dm.get("decoder_net"))
    dd = model_cfg["decoder_dist"]
    if dd == "Bernoulli":                                                 # distributions.py:20-25
        rec = bernoulli_log_prob(dec, x)
    elif dd == "IdentityGaussian":                                        # distributions.py:41-55
        dloc = linear(dec.reshape(dec.shape[0], -1), p["decoder_dist/linear/w"], p["decoder_dist/linear/b"])
        rec = normal_log_prob(x, dloc, torch.exp(p["decoder_dist/log_scale"]))
    else:
        raise KeyError(dd)
    rec = rec.reshape(rec.shape[0], -1).sum(-1)                           # vae.py:127-128

    kl = mvn_tril_kl_to_std_normal(loc, tril)                             # vae.py:130

    x_o_b = torch.cat([x * b, b], -1)                                     # vae.py:132-133
    pfeats = _net(p, penc_kind, penc_cfg, "partial_encoder_net", x_o_b, dm.get("partial_encoder_net"))

    zm = z.detach() if model_cfg.get("matching_ll_stop_gradients", False) else z   # vae.py:136-137
    if ppost_kind == "AutoregressiveGMM":
        mll = autoregressive_gmm_log_prob(p, "partial_posterior_dist", pfeats, zm, k,
                                          ppost_cfg.get("num_components", 10),
                                          ppost_cfg.get("residual_blocks", 2))
    elif ppost_kind == "TriLGaussian":
        ploc, ptril = tril_gaussian_params(p, "partial_posterior_dist", pfeats, k)
        mll = mvn_tril_log_prob(zm, ploc, ptril)
    elif ppost_kind == "DiagonalGaussian":
        ploc, pscale = diagonal_gaussian_params(p, "partial_posterior_dist", pfeats, k)
        mll = mvn_tril_log_prob(zm, ploc, torch.diag_embed(pscale))
    else:
        raise KeyError(ppost_kind)
    return {"reconstruction_ll": rec, "kl": kl, "matching_ll": mll, "z": z}


# ----------------------------------------------------------------------------------------------
# Evaluation paths: impute (vae.py:146-169), is_log_prob (vae.py:171-226), NRMSE (eval_pm_vae_uci.py:60-66)
# ----------------------------------------------------------------------------------------------
def autoregressive_gmm_sample(p: Params, prefix: str, context: Tensor, gumbel: Tensor, eps: Tensor, event_size: int,
                              num_components: int = 10, residual_blocks: int = 2) -> Tensor:
    """_AutoregressiveDistribution._sample_n (distributions.py:168-190) for rows that already carry their own context
    (the vmap over contexts and the broadcast over n flattened into rows).  Step i: net([x*mask_i, mask_i, ctx]) ->
    MixtureSameFamily; only column i of its sample is kept (`updates = out.sample(...) * (arange == i)`).
    The draw is explicit: component = argmax(logits + gumbel[:, i]) (jax.random.categorical is Gumbel-max),
    value = mean_c + scale_c * eps[:, i].  gumbel [R, k, nc], eps [R, k]."""
    R = context.shape[0]
    ctx = context.reshape(R, -1)
    ar = torch.arange(event_size, dtype=eps.dtype)
    x = torch.zeros((R, event_size), dtype=eps.dtype)
    nc = num_components
    for i in range(event_size):
        mask = (ar < i).to(eps.dtype).expand_as(x)
        h = residual_mlp(p, f"{prefix}/mlp", torch.cat([x * mask, mask, ctx], -1), residual_blocks)
        head = linear(h, p[f"{prefix}/gmm/linear/w"], p[f"{prefix}/gmm/linear/b"])
        prm = head.reshape(R, event_size, 3 * nc)[:, i]
        comp = torch.argmax(prm[:, :nc] + gumbel[:, i], -1, keepdim=True)
        mean = torch.gather(prm[:, nc:2 * nc], 1, comp)[:, 0]
        scale = torch.gather(softplus(prm[:, 2 * nc:]) + 1e-5, 1, comp)[:, 0]
        x = x.clone()
        x[:, i] = mean + scale * eps[:, i]
    return x


def _partial_posterior(p: Params, model_cfg: dict, x: Tensor, b: Tensor):
    k = model_cfg["latent_dim"]
    enc_kind = model_cfg["encoder_net"]
    enc_cfg = model_cfg.get("encoder_net_config") or {}
    penc_kind = model_cfg.get("partial_encoder_net", enc_kind)
    penc_cfg = model_cfg.get("partial_encoder_net_config", enc_cfg) or {}
    post_kind = model_cfg["posterior_dist"]
    kind = model_cfg.get("partial_posterior_dist", post_kind)
    cfg = dict(model_cfg.get("partial_posterior_dist_config", model_cfg.get("posterior_dist_config", {})) or {})
    pfeats = _net(p, penc_kind, penc_cfg, "partial_encoder_net", torch.cat([x * b, b], -1))
    return kind, cfg, pfeats, k


def _rep(t: Tensor, S: int) -> Tensor:
    """rows b*S + s (sample-minor), the layout of the HIP evaluation kernels"""
    return t.repeat_interleave(S, dim=0)


def _sample_partial_posterior(p, kind, cfg, pfeats, k, noise, S):
    """-> z [B*S, k] and a closure giving log q(z | x_o) of such rows"""
    if kind == "AutoregressiveGMM":
        nc, rb = cfg.get("num_components", 10), cfg.get("residual_blocks", 2)
        ctx = _rep(pfeats.reshape(pfeats.shape[0], -1), S)
        z = autoregressive_gmm_sample(p, "partial_posterior_dist", ctx, noise["gumbel"].reshape(-1, k, nc),
                                      noise["eps"].reshape(-1, k), k, nc, rb)
        return z, lambda zz: autoregressive_gmm_log_prob(p, "partial_posterior_dist", ctx, zz, k, nc, rb)
    if kind == "TriLGaussian":
        loc, tril = tril_gaussian_params(p, "partial_posterior_dist", pfeats, k)
    else:
        loc, scale = diagonal_gaussian_params(p, "partial_posterior_dist", pfeats, k)
        tril = torch.diag_embed(scale)
    loc, tril = _rep(loc, S), _rep(tril, S)
    z = loc + torch.einsum("bij,bj->bi", tril, noise["eps"].reshape(-1, k))
    return z, lambda zz: mvn_tril_log_prob(zz, loc, tril)


def _decoder(p: Params, model_cfg: dict, z: Tensor):
    dec = _net(p, model_cfg["decoder_net"], model_cfg.get("decoder_net_config") or {}, "decoder_net", z)
    if model_cfg["decoder_dist"] == "Bernoulli":
        return "Bernoulli", dec
    return "IdentityGaussian", linear(dec.reshape(dec.shape[0], -1), p["decoder_dist/linear/w"], p["decoder_dist/linear/b"])


def pm_vae_impute(p: Params, model_cfg: dict, x_o: Tensor, b: Tensor, noise: Dict[str, Tensor]) -> Tensor:
    """PosteriorMatchingVAE.impute (vae.py:146-169) -> [S, B, ...].  noise["eps"] [B,S,k] (+ "gumbel" [B,S,k,nc])."""
    S = noise["eps"].shape[1]
    x_o = x_o * b
    kind, cfg, pfeats, k = _partial_posterior(p, model_cfg, x_o, b)
    z, _ = _sample_partial_posterior(p, kind, cfg, pfeats, k, noise, S)
    dkind, out = _decoder(p, model_cfg, z)
    mean = torch.sigmoid(out) if dkind == "Bernoulli" else out                     # decoder(u).mean()
    mean = mean.reshape((x_o.shape[0], S) + tuple(x_o.shape[1:])).transpose(0, 1)
    return torch.where(b[None] > 0, x_o[None], mean)


def pm_vae_is_log_prob(p: Params, model_cfg: dict, x: Tensor, b: Tensor, noise: Dict[str, Tensor]):
    """PosteriorMatchingVAE.is_log_prob (vae.py:171-226) -> (log p(x) [B], log p(x_u | x_o) [B]).
    noise: "eps_posterior" [B,S,k] for q(z|x); "eps" (+"gumbel") for q(z|x_o)."""
    k = model_cfg["latent_dim"]
    B, S = x.shape[0], noise["eps"].shape[1]
    feats = _net(p, model_cfg["encoder_net"], model_cfg.get("encoder_net_config") or {}, "encoder_net", x)
    if model_cfg["posterior_dist"] == "TriLGaussian":
        loc, tril = tril_gaussian_params(p, "posterior_dist", feats, k)
    else:
        loc, scale = diagonal_gaussian_params(p, "posterior_dist", feats, k)
        tril = torch.diag_embed(scale)
    loc, tril = _rep(loc, S), _rep(tril, S)
    z = loc + torch.einsum("bij,bj->bi", tril, noise["eps_posterior"].reshape(-1, k))
    kind, cfg, pfeats, _ = _partial_posterior(p, model_cfg, x, b)
    z_xo, log_q_xo = _sample_partial_posterior(p, kind, cfg, pfeats, k, noise, S)

    def dec_ll(zz, w):
        dkind, out = _decoder(p, model_cfg, zz)
        xr = _rep(x, S)
        if dkind == "Bernoulli":
            lls = bernoulli_log_prob(out.reshape(xr.shape), xr)
        else:
            lls = normal_log_prob(xr, out.reshape(xr.shape), torch.exp(p["decoder_dist/log_scale"]))
        if w is not None:
            lls = lls * _rep(w, S)
        return lls.reshape(lls.shape[0], -1).sum(-1)

    def prior_lp(zz):
        return -0.5 * (zz * zz).sum(-1) - 0.5 * k * math.log(2 * math.pi)

    def lme(v):
        v = v.reshape(B, S)
        return torch.logsumexp(v, 1) - math.log(S)

    log_p_x = lme(dec_ll(z, None) + prior_lp(z) - mvn_tril_log_prob(z, loc, tril))
    log_p_xo = lme(dec_ll(z_xo, b) + prior_lp(z_xo) - log_q_xo(z_xo))
    return log_p_x, log_p_x - log_p_xo


def pm_vae_expected_info_gains(p: Params, model_cfg: dict, x: Tensor, b: Tensor, noise: Dict[str, Tensor]) -> Tensor:
    """PosteriorMatchingVAE.expected_info_gains (vae.py:228-290) for ONE instance (no batch axis): the expected drop of
    the partial posterior's entropy when feature i becomes observed, -inf where b == 1.  noise["eps"] [1,S,k].
    Gaussian partial posteriors only: TFP's Autoregressive (AutoregressiveGMM) has no entropy(), the reference raises."""
    S = noise["eps"].shape[1]
    x_o = x * b                                                                      # :249
    kind, cfg, pfeats, k = _partial_posterior(p, model_cfg, x_o[None], b[None])     # :250-252
    if kind == "AutoregressiveGMM":
        raise NotImplementedError("tfd.Autoregressive has no analytic entropy")
    z, _ = _sample_partial_posterior(p, kind, cfg, pfeats, k, noise, S)              # :253-254
    dkind, out = _decoder(p, model_cfg, z)
    x_u = (torch.sigmoid(out) if dkind == "Bernoulli" else out).reshape((S,) + tuple(x.shape))   # :255 decoder(z).mean()
    F = b.numel()
    one_hots = torch.eye(F, dtype=b.dtype).reshape((F,) + tuple(b.shape))           # :257-259
    masks = torch.cat([b[None], torch.maximum(b[None], one_hots)], 0)               # :261-262  [F+1, ...]
    x_o_u = torch.where(b[None] == 1, x_o[None], x_u)                               # :264-268  [S, ...]
    ents = []
    for s in range(S):                                                               # hk.scan over the samples, :270-277
        xs = x_o_u[s][None].expand(masks.shape[0], *x.shape)
        _, _, pf, _ = _partial_posterior(p, model_cfg, xs * masks, masks)            # concat([x * masks, masks]) (masks are 0/1)
        if kind == "TriLGaussian":
            _, tril = tril_gaussian_params(p, "partial_posterior_dist", pf, k)
            logdet = torch.log(torch.diagonal(tril, dim1=-2, dim2=-1)).sum(-1)
        else:
            _, scale = diagonal_gaussian_params(p, "partial_posterior_dist", pf, k)
            logdet = torch.log(scale).sum(-1)
        ents.append(0.5 * k * (1.0 + math.log(2 * math.pi)) + logdet)               # MVN entropy
    ents = torch.stack(ents).mean(0)                                                 # :278
    gains = (ents[0] - ents[1:]).reshape(b.shape)                                    # :280-283
    gains = torch.where(b == 0, gains, torch.full_like(gains, -math.inf))            # :284
    return gains.reshape(-1)


def nrmse_score(imputations, true_data, observed_mask):
    """eval_pm_vae_uci.py:60-66 (numpy): per-feature RMSE over the missing entries / feature std, mean over features."""
    import numpy as np

    error = (imputations - true_data) ** 2
    mse = np.sum(error, axis=-2) / np.count_nonzero(1.0 - observed_mask, axis=-2)
    return np.mean(np.sqrt(mse) / np.std(true_data, axis=-2), axis=-1)


def cyclical_annealing_beta(step: int, low: float, high: float, period: int, delay: int = 0) -> float:
    """cyclical_annealing_schedule (utils.py:124-136)."""
    count = step - delay
    count = min(max(count % period, 0), period // 2)
    frac = 1 - count / (period // 2)
    x = (low - high) * frac + high
    return x * float(step >= delay)


def linear_schedule_value(step: int, init_value: float, end_value: float, transition_steps: int,
                          transition_begin: int = 0) -> float:
    """optax.linear_schedule (polynomial_schedule power=1), used for beta "monotonic" (train_pm_vae.py:32-38)."""
    if transition_steps <= 0:
        return end_value
    count = min(max(step - transition_begin, 0), transition_steps)
    frac = 1 - count / transition_steps
    return (init_value - end_value) * frac + end_value


def beta_value(cfg: dict, step: int) -> float:
    """get_beta_schedule (train_pm_vae.py:28-43)."""
    bcfg = cfg.get("beta") or {}
    if "schedule" not in bcfg:
        return 1.0
    if bcfg["schedule"] == "monotonic":
        return linear_schedule_value(step, bcfg["low_value"], bcfg["high_value"], bcfg["transition_steps"],
                                     bcfg.get("transition_begin", 0))
    if bcfg["schedule"] == "cyclic":
        return cyclical_annealing_beta(step, bcfg["low_value"], bcfg["high_value"], bcfg["period"],
                                       bcfg.get("delay", 0))
    raise KeyError(bcfg["schedule"])


def pm_vae_loss(p: Params, cfg: dict, x: Tensor, b: Tensor, eps: Tensor, step: int, dropout_masks=None):
    """loss_fn (train_pm_vae.py:58-72): -mean(rec - beta*kl) + matching_coef * (-mean(matching_ll))."""
    out = pm_vae_forward(p, cfg["model"], x, b, eps, dropout_masks)
    beta = beta_value(cfg, step)
    elbo = (out["reconstruction_ll"] - beta * out["kl"]).mean()
    matching_loss = -out["matching_ll"].mean()
    loss = -elbo + cfg.get("matching_coef", 1.0) * matching_loss
    aux = {"reconstruction_ll": out["reconstruction_ll"].mean(), "kl": out["kl"].mean(),
           "matching_ll": out["matching_ll"].mean(), "beta": beta}
    return loss, aux, out


def lr_value(cfg: dict, count: int) -> float:
    """optax.exponential_decay(init_value, transition_steps, decay_rate) non-staircase
    (train_pm_vae.py:74, configs/pm_vae_mnist.py:45-48)."""
    s = cfg["lr_schedule"]
    return s["init_value"] * s["decay_rate"] ** (count / s["transition_steps"])


def adam_update(p: Params, g: Params, m: Params, v: Params, count: int, cfg: dict) -> None:
    """optax.chain(scale_by_adam, add_decayed_weights(mask ndim != 1), scale_by_schedule, scale(-1))
    (train_pm_vae.py:74-83), applied in place; ``count`` = number of updates already done."""
    adam = cfg.get("adam") or {}
    b1, b2, eps = adam.get("b1", 0.9), adam.get("b2", 0.999), adam.get("eps", 1e-8)
    wd = cfg.get("weight_decay", 0.0)
    lr = lr_value(cfg, count)
    t = count + 1
    for name in p:
        m[name].mul_(b1).add_(g[name], alpha=1 - b1)
        v[name].mul_(b2).addcmul_(g[name], g[name], value=1 - b2)
        m_hat = m[name] / (1 - b1 ** t)
        v_hat = v[name] / (1 - b2 ** t)
        u = m_hat / (torch.sqrt(v_hat) + eps)
        if wd != 0.0 and p[name].ndim != 1:
            u = u + wd * p[name]
        p[name].add_(u, alpha=-lr)


def train_step(p: Params, m: Params, v: Params, cfg: dict, x: Tensor, b: Tensor, eps: Tensor, step: int,
               dropout_masks=None):
    """One bax.Trainer step as the reference drives it (train_pm_vae.py:85-102): value_and_grad of
    loss_fn, optimizer.update, apply_updates.  Returns (loss, aux, grads)."""
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in p.items()}
    loss, aux, _ = pm_vae_loss(leaves, cfg, x, b, eps, step, dropout_masks)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(leaves, grads)}
    adam_update(p, g, m, v, step, cfg)
    return loss.detach(), {k: (a.detach() if isinstance(a, Tensor) else a) for k, a in aux.items()}, g


# ----------------------------------------------------------------------------------------------
# Parameter specification and haiku-style initialisation
# ----------------------------------------------------------------------------------------------
def _conv_out(in_size: int, k: int, s: int, padding: str) -> int:
    return -(-in_size // s) if padding == "SAME" else (in_size - k) // s + 1


def param_shapes(model_cfg: dict, x_shape: Sequence[int]) -> Dict[str, Tuple[int, ...]]:
    """Shapes of every parameter PosteriorMatchingVAE.from_config(model_cfg) creates for inputs of
    shape ``x_shape`` ([H, W, C] images or [D] features), in creation order."""
    k = model_cfg["latent_dim"]
    shapes: Dict[str, Tuple[int, ...]] = {}

    def add_linear(name, fin, fout):
        shapes[f"{name}/w"] = (fin, fout)
        shapes[f"{name}/b"] = (fout,)

    def add_net(kind, cfg, prefix, in_shape):
        if kind == "ConvEncoder":
            h, w, c = in_shape
            layers = cfg["conv_layers"]
            for i, (f, ks, s) in enumerate(layers):
                pad = "VALID" if i == len(layers) - 1 else "SAME"
                shapes[f"{prefix}/conv_{i}/w"] = (ks, ks, c, f)
                shapes[f"{prefix}/conv_{i}/b"] = (f,)
                h, w, c = _conv_out(h, ks, s, pad), _conv_out(w, ks, s, pad), f
            return (h, w, c)
        if kind == "ConvDecoder":
            h, w, c = 1, 1, in_shape[-1]
            for i, (f, ks, s) in enumerate(cfg["conv_layers"]):
                shapes[f"{prefix}/conv_t_{i}/w"] = (ks, ks, f, c)
                shapes[f"{prefix}/conv_t_{i}/b"] = (f,)
                h, w = (h * s, w * s) if i > 0 else ((h - 1) * s + ks, (w - 1) * s + ks)
                c = f
            return (h, w, c)
        if kind == "ResidualMLP":
            fin = int(math.prod(in_shape))
            hu = cfg.get("hidden_units", 256)
            add_linear(f"{prefix}/linear_0", fin, hu)
            for j in range(cfg.get("residual_blocks", 2)):
                add_linear(f"{prefix}/block_{j}/linear_0", hu, hu)
                add_linear(f"{prefix}/block_{j}/linear_1", hu, hu)
            return (hu,)
        raise KeyError(kind)

    def add_dist(kind, cfg, prefix, feat_shape):
        fin = int(math.prod(feat_shape))
        if kind == "TriLGaussian":
            add_linear(f"{prefix}/linear", fin, k + k * (k + 1) // 2)
        elif kind == "DiagonalGaussian":
            add_linear(f"{prefix}/linear", fin, 2 * k)
        elif kind == "AutoregressiveGMM":
            hu = cfg.get("hidden_units", 256)
            add_net("ResidualMLP", {"hidden_units": hu, "residual_blocks": cfg.get("residual_blocks", 2)},
                    f"{prefix}/mlp", (2 * k + fin,))
            add_linear(f"{prefix}/gmm/linear", hu, 3 * cfg.get("num_components", 10) * k)
        elif kind == "IdentityGaussian":
            add_linear(f"{prefix}/linear", fin, cfg["event_size"])
            shapes[f"{prefix}/log_scale"] = ()
        elif kind == "Bernoulli":
            pass
        else:
            raise KeyError(kind)

    x_shape = tuple(x_shape)
    xb_shape = x_shape[:-1] + (2 * x_shape[-1],)
    enc_kind = model_cfg["encoder_net"]
    enc_cfg = model_cfg.get("encoder_net_config") or {}
    f = add_net(enc_kind, enc_cfg, "encoder_net", x_shape)
    add_dist(model_cfg["posterior_dist"], model_cfg.get("posterior_dist_config") or {}, "posterior_dist", f)
    f = add_net(model_cfg["decoder_net"], model_cfg.get("decoder_net_config") or {}, "decoder_net", (k,))
    add_dist(model_cfg["decoder_dist"], model_cfg.get("decoder_dist_config") or {}, "decoder_dist", f)
    f = add_net(model_cfg.get("partial_encoder_net", enc_kind),
                model_cfg.get("partial_encoder_net_config", enc_cfg) or {}, "partial_encoder_net", xb_shape)
    add_dist(model_cfg.get("partial_posterior_dist", model_cfg["posterior_dist"]),
             model_cfg.get("partial_posterior_dist_config", model_cfg.get("posterior_dist_config", {})) or {},
             "partial_posterior_dist", f)
    return shapes


def init_params(model_cfg: dict, x_shape: Sequence[int], seed: int = 1, dtype=torch.float64) -> Params:
    """haiku default init (Appendix A1/A2/A4): w ~ TruncatedNormal(+-2 sigma) * sigma with
    sigma = 1/sqrt(fan_in) (conv: kh*kw*Cin; conv-T: kh*kw*Cin with Cin = last weight axis;
    linear: in), biases and log_scale zero.  Uses numpy default_rng(seed) so that tests can
    regenerate identical parameters without any file."""
    import numpy as np
    from scipy.special import ndtr, ndtri

    def _tn(size, random_state):      # scipy.stats.truncnorm.rvs(-2, 2, ...): same uniform draws, inverse CDF by ndtri (1000x faster)
        lo, hi = ndtr(-2.0), ndtr(2.0)
        return ndtri(lo + random_state.uniform(size=size) * (hi - lo))

    rng = np.random.default_rng(seed)
    out: Params = {}
    for name, shp in param_shapes(model_cfg, x_shape).items():
        if name.endswith("/w"):
            if len(shp) == 4 and "/conv_t_" in name:
                fan_in = shp[0] * shp[1] * shp[3]
            elif len(shp) == 4:
                fan_in = shp[0] * shp[1] * shp[2]
            else:
                fan_in = shp[0]
            arr = _tn(shp, rng) / math.sqrt(fan_in)
        else:
            arr = np.zeros(shp)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out
