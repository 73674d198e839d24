"""CPU oracle for the VQ-VAE (stage 1) training step.  TEST INFRASTRUCTURE ONLY.

CPU restatement (torch tensors on the CPU, float64 by default) of the arithmetic behind
``train_vqvae.py`` of the reference: ``VQVAE.__call__`` (posterior_matching/models/vqvae.py:78-96),
its conv-residual encoder / decoder (vqvae.py:133-266), ``hk.nets.VectorQuantizerEMA`` (third
party, dm-haiku 0.0.5 - not under /root/reference; restated from its published algorithm,
SURVEY.md Appendix A5) and ``optax.adam`` (train_vqvae.py:82).  Imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg - never by the product package.

PARITY UNPINNED (see oracle/pm_vae_oracle.py's header for why: the reference cannot be imported
here and ships no tests or golden vectors).  Pinned instead by the self-derived known-answer
tests in tests/test_oracle_kat.py (quantised rows are codebook rows, straight-through gradient is
the identity, 1 <= perplexity <= K, decay-0 EMA equals one batch k-means step).

One deliberate reading of the third-party code, recorded because SURVEY.md A5 words it
differently: ``self.ema_dw.initialize(embeddings)`` in VectorQuantizerEMA only takes the SHAPE and
dtype of ``embeddings`` - haiku's ``ExponentialMovingAverage.initialize`` creates ``hidden`` and
``average`` as zeros ("sets the average to zeros_like the given value").  With zero-debiasing
(`average = hidden / (1 - decay**counter)`) a non-zero start would make the first average
``99 * embeddings + dw``, which cannot be the intent; zeros it is.

Conventions: activations NHWC, conv weights HWIO, transposed-conv weights HW(O)(I), codebook
``embeddings[D, K]``.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import torch

from .pm_vae_oracle import adam_update, conv2d, conv2d_transpose, normal_log_prob, relu

Tensor = torch.Tensor
Params = Dict[str, Tensor]

VQ_EPSILON = 1e-5          # hk.nets.VectorQuantizerEMA(epsilon=1e-5) default


# ----------------------------------------------------------------------------------------------
# networks (vqvae.py:133-266)
# ----------------------------------------------------------------------------------------------
def conv_residual_stack(p: Params, prefix: str, h: Tensor, residual_blocks: int, activate_final: bool = True) -> Tensor:
    """ConvResidualStack.__call__ (vqvae.py:148-181): h += conv1x1(relu(conv3x3(relu(h)))), final relu."""
    for i in range(residual_blocks):
        c3 = conv2d(relu(h), p[f"{prefix}/res3x3_{i}/w"], p[f"{prefix}/res3x3_{i}/b"], 1, "SAME")
        c1 = conv2d(relu(c3), p[f"{prefix}/res1x1_{i}/w"], p[f"{prefix}/res1x1_{i}/b"], 1, "SAME")
        h = h + c1
    return relu(h) if activate_final else h


def conv_residual_encoder(p: Params, prefix: str, x: Tensor, residual_blocks: int) -> Tensor:
    """ConvResidualEncoder.__call__ (vqvae.py:191-217): 4x4/2, 4x4/2, 3x3/1 convs (SAME, relu) + stack."""
    h = relu(conv2d(x, p[f"{prefix}/enc_1/w"], p[f"{prefix}/enc_1/b"], 2, "SAME"))
    h = relu(conv2d(h, p[f"{prefix}/enc_2/w"], p[f"{prefix}/enc_2/b"], 2, "SAME"))
    h = relu(conv2d(h, p[f"{prefix}/enc_3/w"], p[f"{prefix}/enc_3/b"], 1, "SAME"))
    return conv_residual_stack(p, prefix, h, residual_blocks)


def conv_residual_decoder(p: Params, prefix: str, z: Tensor, residual_blocks: int) -> Tuple[Tensor, Tensor]:
    """ConvResidualDecoder.__call__ (vqvae.py:235-266) -> (loc, scale) of the Normal it returns;
    scale = exp(log_scale) + 1e-5 with one scalar parameter (init 0)."""
    h = conv2d(z, p[f"{prefix}/dec_1/w"], p[f"{prefix}/dec_1/b"], 1, "SAME")
    h = conv_residual_stack(p, prefix, h, residual_blocks)
    h = relu(conv2d_transpose(h, p[f"{prefix}/dec_2/w"], p[f"{prefix}/dec_2/b"], 2, "SAME"))
    loc = conv2d_transpose(h, p[f"{prefix}/dec_3/w"], p[f"{prefix}/dec_3/b"], 2, "SAME")
    scale = torch.exp(p[f"{prefix}/log_scale"]) + 1e-5
    return loc, scale


# ----------------------------------------------------------------------------------------------
# hk.nets.VectorQuantizerEMA (third party)
# ----------------------------------------------------------------------------------------------
def vq_distances(flat: Tensor, embeddings: Tensor) -> Tensor:
    """|x|^2 - 2 x.E + |e|^2, the form haiku writes (not cdist), [N, K]."""
    return (flat ** 2).sum(1, keepdim=True) - 2.0 * flat @ embeddings + (embeddings ** 2).sum(0, keepdim=True)


def ema_update(state: Dict[str, Tensor], name: str, value: Tensor, decay: float) -> Tensor:
    """hk.ExponentialMovingAverage(decay, zero_debias=True).__call__: counter += 1;
    hidden = hidden*decay + value*(1-decay); average = hidden / (1 - decay**counter)."""
    counter = int(state[f"{name}/counter"]) + 1
    hidden = state[f"{name}/hidden"] * decay + value * (1.0 - decay)
    average = hidden / (1.0 - decay ** counter)
    state[f"{name}/counter"] = torch.tensor(counter)
    state[f"{name}/hidden"] = hidden
    state[f"{name}/average"] = average
    return average


def vector_quantizer_ema(state: Dict[str, Tensor], inputs: Tensor, commitment_cost: float, decay: float,
                         is_training: bool, epsilon: float = VQ_EPSILON) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """VectorQuantizerEMA.__call__ -> (outputs, new_state).  ``state`` is not modified."""
    E = state["vq/embeddings"]
    D, K = E.shape
    flat = inputs.reshape(-1, D)
    dist = vq_distances(flat, E)
    idx = torch.argmax(-dist, dim=1)                      # first maximum = nearest code, lowest index on ties
    enc = torch.nn.functional.one_hot(idx, K).to(flat.dtype)
    quantized = E.t()[idx].reshape(inputs.shape)          # self.quantize(indices): embedding lookup
    e_latent_loss = ((quantized.detach() - inputs) ** 2).mean()
    new_state = dict(state)
    if is_training:
        with torch.no_grad():
            cs = ema_update(new_state, "vq/ema_cluster_size", enc.sum(0), decay)
            dw = ema_update(new_state, "vq/ema_dw", flat.detach().t() @ enc, decay)
            n = cs.sum()
            cs = (cs + epsilon) / (n + K * epsilon) * n
            new_state["vq/embeddings"] = dw / cs.reshape(1, -1)
    loss = commitment_cost * e_latent_loss
    quantized_st = inputs + (quantized - inputs).detach()   # straight-through estimator
    avg = enc.mean(0)
    perplexity = torch.exp(-(avg * torch.log(avg + 1e-10)).sum())
    out = {"quantize": quantized_st, "loss": loss, "perplexity": perplexity, "encodings": enc,
           "encoding_indices": idx.reshape(inputs.shape[:-1])}
    return out, new_state


# ----------------------------------------------------------------------------------------------
# model / loss / train step
# ----------------------------------------------------------------------------------------------
def vqvae_forward(p: Params, state: Dict[str, Tensor], model_cfg: dict, x: Tensor, is_training: bool = False):
    """VQVAE.__call__ (vqvae.py:78-96) -> (out dict, new_state)."""
    if not model_cfg.get("use_ema", True):
        raise NotImplementedError("use_ema=False (hk.nets.VectorQuantizer) is not on the BASELINE configs")
    rb = model_cfg.get("residual_blocks", 2)
    h = conv_residual_encoder(p, "encoder", x, rb)
    z = conv2d(h, p["pre_vq_conv/w"], p["pre_vq_conv/b"], 1, "SAME")
    vq, new_state = vector_quantizer_ema(state, z, model_cfg.get("commitment_cost", 0.25),
                                         model_cfg.get("decay", 0.99), is_training)
    loc, scale = conv_residual_decoder(p, "decoder", vq["quantize"], rb)
    ll = normal_log_prob(x, loc, scale).reshape(x.shape[0], -1).sum(1)      # einops.reduce "b ... -> b", sum
    reconstruction_loss = -ll.mean()
    loss = reconstruction_loss + vq["loss"]
    out = {"loss": loss, "vq_output": vq, "z": z, "reconstruction": loc, "reconstruction_loss": reconstruction_loss}
    return out, new_state


def vqvae_loss(p: Params, state, cfg: dict, x: Tensor, is_training: bool = True):
    """loss_fn of train_vqvae.py:67-75 -> (loss, aux, out, new_state)."""
    out, new_state = vqvae_forward(p, state, cfg["model"], x, is_training)
    aux = {"perplexity": out["vq_output"]["perplexity"], "reconstruction_loss": out["reconstruction_loss"],
           "vq_loss": out["vq_output"]["loss"]}
    return out["loss"], aux, out, new_state


def optimizer_cfg(cfg: dict) -> dict:
    """optax.adam(config.learning_rate) (train_vqvae.py:82) in the terms of pm_vae_oracle.adam_update."""
    return {"lr_schedule": {"init_value": cfg["learning_rate"], "transition_steps": 1, "decay_rate": 1.0},
            "weight_decay": 0.0}


def train_step(p: Params, state, m: Params, v: Params, cfg: dict, x: Tensor, step: int):
    """One bax.Trainer step (train_vqvae.py:84-111): grads of loss_fn w.r.t. params, Adam update in
    place, haiku state replaced by the forward's new state.  Returns (loss, aux, grads, new_state)."""
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in p.items()}
    loss, aux, _, new_state = vqvae_loss(leaves, state, cfg, x, True)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(leaves, grads)}
    adam_update(p, g, m, v, step, optimizer_cfg(cfg))
    return loss.detach(), {k: a.detach() for k, a in aux.items()}, g, {k: t.detach() for k, t in new_state.items()}


# ----------------------------------------------------------------------------------------------
# parameter / state specification and haiku-style initialisation
# ----------------------------------------------------------------------------------------------
def _encoder_shapes(prefix: str, cin: int, hu: int, rb: int, rhu: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, k, ci, co):
        s[f"{prefix}/{name}/w"] = (k, k, ci, co)
        s[f"{prefix}/{name}/b"] = (co,)

    conv("enc_1", 4, cin, hu // 2)
    conv("enc_2", 4, hu // 2, hu)
    conv("enc_3", 3, hu, hu)
    for i in range(rb):
        conv(f"res3x3_{i}", 3, hu, rhu)
        conv(f"res1x1_{i}", 1, rhu, hu)
    return s


def param_shapes(model_cfg: dict, in_channels: int = 1) -> Dict[str, Tuple[int, ...]]:
    hu, rb = model_cfg.get("hidden_units", 128), model_cfg.get("residual_blocks", 2)
    rhu, D = model_cfg.get("residual_hidden_units", 128), model_cfg.get("embedding_dim", 64)
    oc = model_cfg.get("output_channels", 3)
    s = _encoder_shapes("encoder", in_channels, hu, rb, rhu)
    s["pre_vq_conv/w"], s["pre_vq_conv/b"] = (1, 1, hu, D), (D,)
    s["decoder/dec_1/w"], s["decoder/dec_1/b"] = (3, 3, D, hu), (hu,)
    for i in range(rb):
        s[f"decoder/res3x3_{i}/w"], s[f"decoder/res3x3_{i}/b"] = (3, 3, hu, rhu), (rhu,)
        s[f"decoder/res1x1_{i}/w"], s[f"decoder/res1x1_{i}/b"] = (1, 1, rhu, hu), (hu,)
    s["decoder/dec_2/w"], s["decoder/dec_2/b"] = (4, 4, hu // 2, hu), (hu // 2,)      # conv-T: [k,k,Cout,Cin]
    s["decoder/dec_3/w"], s["decoder/dec_3/b"] = (4, 4, oc, hu // 2), (oc,)
    s["decoder/log_scale"] = ()
    return s


def init_params(model_cfg: dict, in_channels: int = 1, seed: int = 1, dtype=torch.float64) -> Params:
    """haiku default init (Appendix A1/A2): TruncatedNormal(+-2) / sqrt(fan_in); conv fan_in =
    kh*kw*Cin, transposed conv fan_in = kh*kw*(last weight axis); biases / log_scale zero."""
    import numpy as np
    from scipy.special import ndtr, ndtri

    def _tn(size, random_state):      # scipy.stats.truncnorm.rvs(-2, 2, ...): same uniform draws, inverse CDF by ndtri (1000x faster)
        lo, hi = ndtr(-2.0), ndtr(2.0)
        return ndtri(lo + random_state.uniform(size=size) * (hi - lo))

    rng = np.random.default_rng(seed)
    out: Params = {}
    for name, shp in param_shapes(model_cfg, in_channels).items():
        if name.endswith("/w"):
            fan_in = shp[0] * shp[1] * (shp[3] if "/dec_2/" in name or "/dec_3/" in name else shp[2])
            arr = _tn(shp, rng) / math.sqrt(fan_in)
        else:
            arr = np.zeros(shp)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out


def init_state(model_cfg: dict, seed: int = 2, dtype=torch.float64) -> Dict[str, Tensor]:
    """VectorQuantizerEMA state: embeddings[D,K] ~ VarianceScaling(scale 1, fan_in, uniform) =
    U(-sqrt(3/D), sqrt(3/D)); both EMAs start with hidden = average = 0, counter = 0."""
    import numpy as np

    D, K = model_cfg.get("embedding_dim", 64), model_cfg.get("num_embeddings", 512)
    rng = np.random.default_rng(seed)
    lim = math.sqrt(3.0 / D)
    st = {"vq/embeddings": torch.tensor(rng.uniform(-lim, lim, size=(D, K)), dtype=dtype)}
    for name, shp in (("vq/ema_cluster_size", (K,)), ("vq/ema_dw", (D, K))):
        st[f"{name}/hidden"] = torch.zeros(shp, dtype=dtype)
        st[f"{name}/average"] = torch.zeros(shp, dtype=dtype)
        st[f"{name}/counter"] = torch.tensor(0)
    return st
