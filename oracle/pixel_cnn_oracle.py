"""CPU oracle for the PixelCNN partial posterior and the PM-VQVAE (stage 2) training step.
TEST INFRASTRUCTURE ONLY.

CPU restatement (torch on the CPU, float64 by default) of ``_PixelCNNNetwork.__call__``
(posterior_matching/models/pixel_cnn.py:372-553), ``PixelCNN.log_prob`` (:53-63), the masked
convolutions (:148-211, masks :556-562), ``VQVAEPartialEncoder`` (vqvae.py:99-130) and the loss /
optimizer / freeze predicate of ``train_pm_vqvae.py:81-123``.  Imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg - never by the product package.

PARITY UNPINNED (see oracle/pm_vae_oracle.py's header).  Pinned by the self-derived known-answer
tests in tests/test_oracle_kat.py: the autoregressive property (logits at a position do not depend on
indices at or after it in raster order), normalisation of the categorical, mask shapes.

Only ``num_hierarchies == 1`` is restated (both BASELINE configs, configs/pm_vqvae_mnist.py:22 and
configs/pm_vqvae_celeb_a.py, use 1): the strided down/up-sampling convolutions between hierarchies
(:462-483, :524-546) are then never built.

Conventions: NHWC activations, HWIO conv weights, dense weights [in, out].
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .pm_vae_oracle import adam_update, conv2d, linear
from .vqvae_oracle import _encoder_shapes, conv_residual_encoder, vqvae_forward

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def elu(x: Tensor) -> Tensor:
    """jax.nn.elu, alpha = 1: where(x > 0, x, expm1(x))"""
    return torch.where(x > 0, x, torch.expm1(torch.clamp(x, max=0.0)))


def concat_elu(x: Tensor) -> Tensor:
    """pixel_cnn.py:373-374"""
    return elu(torch.cat([x, -x], dim=-1))


def make_kernel_constraint(kernel_size, valid_rows, valid_columns) -> Tensor:
    """_make_kernel_constraint (pixel_cnn.py:556-562): [kh, kw, 1, 1] mask of ones in the given row / column ranges."""
    mask = np.zeros(kernel_size)
    mask[valid_rows[0]:valid_rows[1], valid_columns[0]:valid_columns[1]] = 1.0
    return torch.tensor(mask[:, :, None, None])


def kernel_plan(receptive_field_dims=(3, 3)):
    """The kernel sizes and masks of pixel_cnn.py:389-422 -> dict name -> (kernel_size, mask)."""
    rows, cols = receptive_field_dims
    valid = {"vertical": (rows - 1, cols), "horizontal": (2, cols // 2 + 1)}
    sizes = {"vertical": (2 * rows - 3, cols), "horizontal": (3, cols)}
    plan = {k: (sizes[k], make_kernel_constraint(sizes[k], (0, v[0]), (0, v[1]))) for k, v in valid.items()}
    plan["vertical_init"] = ((2 * rows - 1, cols), make_kernel_constraint((2 * rows - 1, cols), (0, rows - 1), (0, cols)))
    plan["horizontal_up"] = ((3, cols), make_kernel_constraint((3, cols), (0, 1), (0, cols)))
    plan["horizontal_left"] = ((3, cols), make_kernel_constraint((3, cols), (0, 2), (0, cols // 2)))
    return plan


def masked_conv(p: Params, name: str, x: Tensor, mask: Tensor) -> Tensor:
    """_ConvND.__call__ (pixel_cnn.py:148-211): w *= mask; SAME, stride 1."""
    return conv2d(x, p[f"{name}/w"] * mask.to(x.dtype), p[f"{name}/b"], 1, "SAME")


def _resnet_block(p: Params, name: str, stack: str, input_x: Tensor, extra: Optional[Tensor], cond: Optional[Tensor],
                  plan, drop: Optional[Tensor]) -> Tensor:
    """One gated block of the down pass (:429-460) or the up pass (:491-522).  extra: the tensor whose
    concat_elu goes through the block's hk.Linear and is added after the first conv (None: no Linear)."""
    F = input_x.shape[-1]
    mask = plan[stack][1]
    x = masked_conv(p, f"{name}/conv1", concat_elu(input_x), mask)
    if extra is not None:
        x = x + linear(concat_elu(extra), p[f"{name}/linear/w"], p[f"{name}/linear/b"])
    x = concat_elu(x)
    if drop is not None:                                   # hk.dropout: x * keep / (1 - rate), mask given pre-scaled
        x = x * drop
    x = masked_conv(p, f"{name}/conv2", x, mask)
    if cond is not None:                                   # _build_and_apply_h_projection (:565-568)
        h = linear(cond.reshape(cond.shape[0], -1), p[f"{name}/cond/w"], p[f"{name}/cond/b"])
        x = x + h[:, None, None, :]
    act, gate = x[..., :F], x[..., F:]                     # _apply_sigmoid_gating (:571-574)
    return input_x + torch.sigmoid(gate) * act


def pixel_cnn_logits(p: Params, prefix: str, image_input: Tensor, cfg: dict, conditional_input: Optional[Tensor] = None,
                     dropout_masks: Optional[Sequence[Tensor]] = None) -> Tensor:
    """_PixelCNNNetwork.__call__ (pixel_cnn.py:372-553) -> logits [B, H, W, num_indices].
    dropout_masks: one pre-scaled keep mask [B,H,W,2F] per gated block in execution order
    (down pass: block 0 vertical, block 0 horizontal, ...; then the up pass), or None (training=False)."""
    if cfg.get("num_hierarchies", 1) != 1:
        raise NotImplementedError("num_hierarchies != 1")
    R = cfg.get("num_resnet", 5)
    plan = kernel_plan(tuple(cfg.get("receptive_field_dims", (3, 3))))
    drops = iter(dropout_masks) if dropout_masks is not None else None
    nxt = (lambda: next(drops)) if drops is not None else (lambda: None)
    P = prefix
    x = p[f"{P}/embed/embeddings"][image_input.long()]                         # hk.Embed
    v_init = masked_conv(p, f"{P}/vertical_init", x, plan["vertical_init"][1])
    h_init = masked_conv(p, f"{P}/horizontal_up", x, plan["horizontal_up"][1]) + \
        masked_conv(p, f"{P}/horizontal_left", x, plan["horizontal_left"][1])
    stacks = {"vertical": [v_init], "horizontal": [h_init]}
    for i in range(R):                                                          # down pass (:427-460)
        v = _resnet_block(p, f"{P}/down_{i}/vertical", "vertical", stacks["vertical"][-1], None, conditional_input,
                          plan, nxt())
        stacks["vertical"].append(v)
        h = _resnet_block(p, f"{P}/down_{i}/horizontal", "horizontal", stacks["horizontal"][-1], v, conditional_input,
                          plan, nxt())
        stacks["horizontal"].append(h)
    up = {k: s.pop() for k, s in stacks.items()}
    for i in range(R):                                                          # up pass (:487-522)
        sym_v = stacks["vertical"].pop()
        up["vertical"] = _resnet_block(p, f"{P}/up_{i}/vertical", "vertical", up["vertical"], sym_v, conditional_input,
                                       plan, nxt())
        sym_h = torch.cat([up["vertical"], stacks["horizontal"].pop()], dim=-1)
        up["horizontal"] = _resnet_block(p, f"{P}/up_{i}/horizontal", "horizontal", up["horizontal"], sym_h,
                                         conditional_input, plan, nxt())
    x_out = elu(up["horizontal"])
    return conv2d(x_out, p[f"{P}/out_conv/w"], p[f"{P}/out_conv/b"], 1, "SAME")


def pixel_cnn_log_prob(p: Params, prefix: str, value: Tensor, cfg: dict, conditional_input=None,
                       dropout_masks=None) -> Tensor:
    """PixelCNN.log_prob (pixel_cnn.py:53-63): Categorical(logits).log_prob(value) summed per example."""
    logits = pixel_cnn_logits(p, prefix, value, cfg, conditional_input, dropout_masks)
    lp = torch.log_softmax(logits, dim=-1).gather(-1, value.long().unsqueeze(-1)).squeeze(-1)
    return lp.reshape(lp.shape[0], -1).sum(1)


def vqvae_partial_encoder(p: Params, prefix: str, x_o_b: Tensor, vqvae_cfg: dict) -> Tensor:
    """VQVAEPartialEncoder (vqvae.py:99-130): ConvResidualEncoder -> Flatten -> Linear(conditional_dim)."""
    h = conv_residual_encoder(p, f"{prefix}/encoder", x_o_b, vqvae_cfg.get("residual_blocks", 2))
    return linear(h.reshape(h.shape[0], -1), p[f"{prefix}/linear/w"], p[f"{prefix}/linear/b"])


def pm_vqvae_loss(p: Params, vq_params: Params, vq_state, cfg: dict, vqvae_cfg: dict, x: Tensor, b: Tensor,
                  is_training: bool, dropout_masks=None):
    """loss_fn of train_pm_vqvae.py:81-99 -> (loss, encoding_indices, per-example log-prob)."""
    with torch.no_grad():
        out, _ = vqvae_forward(vq_params, vq_state, vqvae_cfg, x, False)       # frozen, is_training=False
    idx = out["vq_output"]["encoding_indices"]
    x_o_b = torch.cat([x * b, b], dim=-1)
    cond = vqvae_partial_encoder(p, "partial_encoder", x_o_b, vqvae_cfg)
    pc = dict(cfg["pixel_cnn"])
    pc["num_indices"] = vqvae_cfg["num_embeddings"]
    lp = pixel_cnn_log_prob(p, "pixel_cnn", idx, pc, cond, dropout_masks if is_training else None)
    return -lp.mean(), idx, lp


def train_step(p: Params, vq_params: Params, vq_state, m: Params, v: Params, cfg: dict, vqvae_cfg: dict, x: Tensor,
               b: Tensor, step: int, dropout_masks=None):
    """One bax.Trainer step of train_pm_vqvae.py: only the non-"vqvae/" parameters (here: `p`) are trained;
    optimizer = scale_by_adam -> scale_by_schedule(exponential_decay) -> scale(-1), no weight decay (:115-120)."""
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in p.items()}
    loss, idx, _ = pm_vqvae_loss(leaves, vq_params, vq_state, cfg, vqvae_cfg, x, b, True, dropout_masks)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(leaves, grads)}
    adam_update(p, g, m, v, step, {"lr_schedule": cfg["lr_schedule"], "adam": cfg.get("adam"), "weight_decay": 0.0})
    return loss.detach(), g


# ----------------------------------------------------------------------------------------------
# parameter specification / haiku-style init
# ----------------------------------------------------------------------------------------------
def pixel_cnn_param_shapes(prefix: str, cfg: dict, cond_dim: Optional[int]) -> Dict[str, Tuple[int, ...]]:
    F, K, R = cfg.get("num_filters", 160), cfg["num_indices"], cfg.get("num_resnet", 5)
    plan = kernel_plan(tuple(cfg.get("receptive_field_dims", (3, 3))))
    s: Dict[str, Tuple[int, ...]] = {f"{prefix}/embed/embeddings": (K, F)}

    def conv(name, ksz, ci, co):
        s[f"{name}/w"], s[f"{name}/b"] = (ksz[0], ksz[1], ci, co), (co,)

    def lin(name, ci, co):
        s[f"{name}/w"], s[f"{name}/b"] = (ci, co), (co,)

    for name in ("vertical_init", "horizontal_up", "horizontal_left"):
        conv(f"{prefix}/{name}", plan[name][0], F, F)
    for phase in ("down", "up"):
        for i in range(R):
            for stack in ("vertical", "horizontal"):
                base = f"{prefix}/{phase}_{i}/{stack}"
                conv(f"{base}/conv1", plan[stack][0], 2 * F, F)
                if phase == "down" and stack == "horizontal":
                    lin(f"{base}/linear", 2 * F, F)
                elif phase == "up":
                    lin(f"{base}/linear", 2 * F if stack == "vertical" else 4 * F, F)
                conv(f"{base}/conv2", plan[stack][0], 2 * F, 2 * F)
                if cond_dim is not None:
                    lin(f"{base}/cond", cond_dim, 2 * F)
    conv(f"{prefix}/out_conv", (1, 1), F, K)
    return s


def partial_encoder_param_shapes(prefix: str, vqvae_cfg: dict, in_channels: int, grid: Tuple[int, int],
                                 cond_dim: int) -> Dict[str, Tuple[int, ...]]:
    hu = vqvae_cfg.get("hidden_units", 128)
    s = _encoder_shapes(f"{prefix}/encoder", in_channels, hu, vqvae_cfg.get("residual_blocks", 2),
                        vqvae_cfg.get("residual_hidden_units", 128))
    s[f"{prefix}/linear/w"], s[f"{prefix}/linear/b"] = (grid[0] * grid[1] * hu, cond_dim), (cond_dim,)
    return s


def init_params(cfg: dict, vqvae_cfg: dict, in_channels: int = 2, seed: int = 3, dtype=torch.float64) -> Params:
    """Trainable parameters of stage 2.  haiku defaults: conv / linear TruncatedNormal(+-2)/sqrt(fan_in)
    (conv fan_in = kh*kw*Cin over the FULL kernel, pixel_cnn.py:181-184), Embed TruncatedNormal(stddev 1),
    the conditional projections RandomNormal(stddev 1) (:567), biases zero."""
    from scipy.special import ndtr, ndtri

    def _tn(size, random_state):      # scipy.stats.truncnorm.rvs(-2, 2, ...): same uniform draws, inverse CDF by ndtri (1000x faster)
        lo, hi = ndtr(-2.0), ndtr(2.0)
        return ndtri(lo + random_state.uniform(size=size) * (hi - lo))

    pc = dict(cfg["pixel_cnn"])
    pc["num_indices"] = vqvae_cfg["num_embeddings"]
    grid = tuple(pc["image_shape"])
    shapes = partial_encoder_param_shapes("partial_encoder", vqvae_cfg, in_channels, grid, cfg["conditional_dim"])
    shapes.update(pixel_cnn_param_shapes("pixel_cnn", pc, cfg["conditional_dim"]))
    rng = np.random.default_rng(seed)
    out: Params = {}
    for name, shp in shapes.items():
        if name.endswith("/embeddings"):
            arr = _tn(shp, rng)
        elif name.endswith("/cond/w"):
            arr = rng.normal(size=shp)
        elif name.endswith("/w"):
            arr = _tn(shp, rng) / math.sqrt(int(np.prod(shp[:-1])))
        else:
            arr = np.zeros(shp)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out


# ----------------------------------------------------------------------------------------------
# ancestral sampling, imputation and PSNR (pixel_cnn.py:76-146, vqvae.py:269-312, eval_pm_vqvae.py:120-138)
# ----------------------------------------------------------------------------------------------
def pixel_cnn_sample(p: Params, prefix: str, cfg: dict, conditional_input: Tensor, num_samples: int,
                     gumbel: Tensor) -> Tensor:
    """PixelCNN._sample_n with conditioning (pixel_cnn.py:102-124): for every conditioning vector, draw
    `num_samples` index grids position by position in raster order.  Returns [num_samples, B, H, W].
    jax.random.categorical is the Gumbel-max trick; the noise is explicit here: gumbel [H*W, B*num_samples, K]
    (row b*num_samples + s belongs to sample s of example b)."""
    H, W = cfg["image_shape"]
    B = conditional_input.shape[0]
    cond = conditional_input.reshape(B, -1).repeat_interleave(num_samples, dim=0)       # jnp.tile per vmap lane
    x = torch.zeros((B * num_samples, H, W), dtype=torch.long)
    for i in range(H * W):
        logits = pixel_cnn_logits(p, prefix, x, cfg, cond)
        r, c = divmod(i, W)
        x[:, r, c] = torch.argmax(logits[:, r, c, :] + gumbel[i], dim=-1)
    return x.reshape(B, num_samples, H, W).permute(1, 0, 2, 3)


def pixel_cnn_sample_unconditional(p: Params, prefix: str, cfg: dict, num_samples: int, gumbel: Tensor) -> Tensor:
    """PixelCNN._sample_n without conditioning (pixel_cnn.py:82-100): `num_samples` chains from an all-zero grid; position
    i of every chain is replaced by the draw from the network's distribution at that position (`jnp.where(one_hot(i), samples,
    x)`).  gumbel [H*W, num_samples, K] makes the categorical draws explicit.  Returns [num_samples, H, W]."""
    H, W = cfg["image_shape"]
    x = torch.zeros((num_samples, H, W), dtype=torch.long)
    for i in range(H * W):
        logits = pixel_cnn_logits(p, prefix, x, cfg, None)
        r, c = divmod(i, W)
        x[:, r, c] = torch.argmax(logits[:, r, c, :] + gumbel[i], dim=-1)
    return x


def vqvae_impute(p: Params, vq_params: Params, vq_state, cfg: dict, vqvae_cfg: dict, x: Tensor, b: Tensor,
                 num_samples: int, gumbel: Tensor) -> Tensor:
    """vqvae_impute (vqvae.py:269-312) -> [B, num_samples, H, W, C] with observed pixels kept, clipped to [0,1]."""
    from .vqvae_oracle import conv_residual_decoder

    x_o_b = torch.cat([x * b, b], dim=-1)
    cond = vqvae_partial_encoder(p, "partial_encoder", x_o_b, vqvae_cfg)
    pc = dict(cfg["pixel_cnn"])
    pc["num_indices"] = vqvae_cfg["num_embeddings"]
    samples = pixel_cnn_sample(p, "pixel_cnn", pc, cond, num_samples, gumbel)           # [S, B, H, W]
    E = vq_state["vq/embeddings"]
    out = []
    for s in range(num_samples):
        q = E.t()[samples[s]]                                                        # vq.quantize(indices)
        loc, _ = conv_residual_decoder(vq_params, "decoder", q, vqvae_cfg.get("residual_blocks", 2))
        out.append(loc)
    imp = torch.stack(out, dim=1)                                                     # [B, S, H, W, C]
    imp = torch.where(b[:, None].bool(), x[:, None].expand_as(imp), imp)
    return imp.clamp(0.0, 1.0)


def imputation_psnr(imputations: Tensor, x: Tensor) -> Tensor:
    """eval_pm_vqvae.py:133-136: PSNR of the mean imputation, per example."""
    mean_imp = imputations.mean(1)
    mse = ((mean_imp - x) ** 2).reshape(x.shape[0], -1).mean(1)
    return -10.0 * torch.log10(mse)
