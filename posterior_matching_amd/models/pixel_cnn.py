"""Host-side mirror of posterior_matching/models/pixel_cnn.py of the reference: the PixelCNN
distribution used as partial posterior over VQ-VAE code indices (stage 2, train_pm_vqvae.py).

Same constructor as the reference's `PixelCNN` (pixel_cnn.py:26-47) and the same `log_prob(value,
training, conditional_input)` contract (:53-63); the arithmetic runs in libpmhip.so: the masked
convolutions walk only the taps their mask keeps (sub-kernel form of the gather-GEMM engine), the
per-block conditional projections (a new hk.Linear per gated block, :565-568) run as ONE grouped
GEMM, and the row-wise glue (concat_elu, dropout, gating, categorical log-prob) are fused HBM-bound
kernels (csrc/pm_pixelcnn.hip).  There is no autodiff: `backward(g_ll)` walks the stored buffers.

Only num_hierarchies == 1 is implemented (every BASELINE config uses 1).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from ..ops import ACT_NONE, LayerGeom
from .core import Module, ParamStore


def _kernel_plan(receptive_field_dims=(3, 3)):
    """kernel sizes and valid (top-left) extents of pixel_cnn.py:389-422: name -> (kh, kw, valid_rows, valid_cols)"""
    rows, cols = receptive_field_dims
    return {
        "vertical": (2 * rows - 3, cols, rows - 1, cols),
        "horizontal": (3, cols, 2, cols // 2 + 1),
        "vertical_init": (2 * rows - 1, cols, rows - 1, cols),
        "horizontal_up": (3, cols, 1, cols),
        "horizontal_left": (3, cols, 2, cols // 2),
    }


class _Layer:
    def __init__(self, store: ParamStore, name: str, geom: LayerGeom, fan_in: int, split: bool = True):
        self.g, self.w, self.b = geom, f"{name}/w", f"{name}/b"
        store.add(self.w, geom.weight_shape, fan_in=fan_in)
        store.add(self.b, (geom.CO,))
        self.ws_f = store.request_split(self.w, geom, "fwd") if split else None
        self.ws_d = store.request_split(self.w, geom, "dgrad") if split else None


class _Block:
    """one gated resnet block: names, layers, which tensors feed it"""

    def __init__(self, name: str, stack: str, group: int):
        self.name, self.stack, self.group = name, stack, group
        self.conv1 = self.conv2 = self.linear = None


class PixelCNN(Module):
    """reference pixel_cnn.py:26-63 (+ the network :339-553)."""

    def __init__(self, num_indices, image_shape, dropout=0.5, num_resnet=15, num_hierarchies=1, num_filters=128,
                 receptive_field_dims=(3, 3), name: Optional[str] = None):
        super().__init__(name)
        if num_hierarchies != 1:
            raise NotImplementedError("PixelCNN(num_hierarchies != 1) has no HIP path (no reference config uses it)")
        self._event_shape = tuple(image_shape)
        self._num_indices, self._dropout = num_indices, float(dropout)
        self._num_resnet, self._num_filters = num_resnet, num_filters
        self._receptive_field_dims = tuple(receptive_field_dims)

    two_streams = True       # forward: the vertical stack on its own stream (see logits)

    @property
    def event_shape(self) -> Tuple[int, ...]:
        return self._event_shape

    # ------------------------------------------------------------------------------------------
    def build(self, store: ParamStore, prefix: str, cond_dim: Optional[int]) -> None:
        self.attach(store, prefix)
        H, W = self._event_shape
        F, K, R = self._num_filters, self._num_indices, self._num_resnet
        plan = _kernel_plan(self._receptive_field_dims)
        self._cond_dim = cond_dim
        store.add(f"{prefix}/embed/embeddings", (K, F), fan_in=-1)             # hk.Embed: TruncatedNormal(stddev 1)

        def mconv(name, kind, ci, co):
            kh, kw, vr, vc = plan[kind]
            return _Layer(store, f"{prefix}/{name}", LayerGeom.masked_conv(H, W, ci, co, kh, kw, vr, vc), kh * kw * ci)

        self.v_init = mconv("vertical_init", "vertical_init", F, F)
        self.h_up = mconv("horizontal_up", "horizontal_up", F, F)
        self.h_left = mconv("horizontal_left", "horizontal_left", F, F)
        self.blocks: List[_Block] = []
        for phase in ("down", "up"):
            for i in range(R):
                for stack in ("vertical", "horizontal"):
                    b = _Block(f"{phase}_{i}/{stack}", stack, len(self.blocks))
                    b.conv1 = mconv(f"{b.name}/conv1", stack, 2 * F, F)
                    lin_in = 0
                    if phase == "down" and stack == "horizontal":
                        lin_in = 2 * F
                    elif phase == "up":
                        lin_in = 2 * F if stack == "vertical" else 4 * F
                    if lin_in:
                        b.linear = _Layer(store, f"{prefix}/{b.name}/linear", LayerGeom.dense(lin_in, F), lin_in)
                    b.conv2 = mconv(f"{b.name}/conv2", stack, 2 * F, 2 * F)
                    self.blocks.append(b)
        self.out_conv = _Layer(store, f"{prefix}/out_conv", LayerGeom.conv(H, W, F, K, 1, 1, "SAME"), F)
        # the conditional projections: added back to back so that the G weight matrices [cond, 2F] (and
        # the G biases) are adjacent in the flat parameter buffer = one grouped GEMM
        if cond_dim is not None:
            self.g_cond = LayerGeom.dense(cond_dim, 2 * F)
            for b in self.blocks:
                store.add(f"{prefix}/{b.name}/cond/w", (cond_dim, 2 * F), fan_in=-2)   # RandomNormal(stddev 1)
            for b in self.blocks:
                store.add(f"{prefix}/{b.name}/cond/b", (2 * F,))

    # -- small helpers ------------------------------------------------------------------------
    def _fwd(self, L: _Layer, x, out, res=None):
        ops.layer_forward(L.g, x, self.store.p[L.w], self.store.p[L.b], out, res=res,
                          wsplit=self.store.split_view(L.ws_f) if L.ws_f is not None else None)

    def _wg(self, L: _Layer, x, dy):
        self.wgrad(L.g, x, dy, self.store.g[L.w], self.store.g[L.b])

    def _dg(self, L: _Layer, dy, dx, res=None):
        ops.layer_dgrad(L.g, dy, self.store.p[L.w], dx, res=res,
                        wsplit=self.store.split_view(L.ws_d) if L.ws_d is not None else None)

    def _cond_kw(self, B):
        F, cd, G = self._num_filters, self._cond_dim, len(self.blocks)
        return dict(B=B, groups=G, in_gs=0, w_gs=cd * 2 * F, out_gs=B * 2 * F, bias_gs=2 * F)

    def _cond_params(self, which):
        first = self.blocks[0].name
        src = self.store.p if which == "p" else self.store.g
        return src[f"{self.prefix}/{first}/cond/w"], src[f"{self.prefix}/{first}/cond/b"]

    # ------------------------------------------------------------------------------------------
    def logits(self, value: torch.Tensor, training: bool = False, conditional_input: Optional[torch.Tensor] = None,
               dropout_masks: Optional[Sequence[torch.Tensor]] = None, seed: int = 0, step_dev=None) -> torch.Tensor:
        """_PixelCNNNetwork.__call__ (:372-553) -> logits [B,H,W,K].  value: int32 [B,H,W] code indices.
        training=True applies hk.dropout with rate `dropout`: keep masks are drawn on the device (Philox,
        keyed by seed / step / block) unless `dropout_masks` (one pre-scaled [B,H,W,2F] mask per gated
        block in execution order) is given."""
        H, W = self._event_shape
        F, K = self._num_filters, self._num_indices
        B = value.shape[0]
        R, P = B * H * W, H * W
        if (conditional_input is None) != (self._cond_dim is None):
            raise ValueError("conditional_input must be given iff the network was built with a conditional_dim")
        self._B, self._value = B, value
        idx = value.reshape(-1)
        sh = lambda c: (B, H, W, c)   # noqa: E731
        emb = self.buf("embed", sh(F))
        ops.embed_fwd(idx, self.P("embed/embeddings"), emb)
        v = self.buf("v_init", sh(F))
        self._fwd(self.v_init, emb, v)
        h_up = self.buf("h_up", sh(F))
        self._fwd(self.h_up, emb, h_up)
        h = self.buf("h_init", sh(F))
        self._fwd(self.h_left, emb, h, res=h_up)
        hproj = None
        if conditional_input is not None:
            cond = conditional_input.reshape(B, -1)
            self._cond = cond
            hproj = self.buf("hproj", (len(self.blocks), B, 2 * F))
            cw, cb = self._cond_params("p")
            ops.layer_forward(self.g_cond, cond, cw, cb, hproj, **self._cond_kw(B))
        self._hproj = hproj
        rate = self._dropout if training else 0.0
        self._drops: List[Optional[torch.Tensor]] = []
        self._io: List[Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]] = []

        # the block that opens with concat_elu(out of THIS block): the next block of the same stack, in execution order
        # (vertical: blocks 0, 2, ..; horizontal: 1, 3, ..; the up pass continues where the down pass ended).  Its ce1 comes
        # out of this block's gate launch (ops.gate_fwd_ce) - one link less in each block's dependent chain.
        fuse_ce = not os.environ.get("PM_NO_GATE_CE")
        nxt = {b_.group: (self.blocks[i + 2] if i + 2 < len(self.blocks) else None) for i, b_ in enumerate(self.blocks)}
        have_ce1 = set()

        self._ce_e: Dict[int, torch.Tensor] = {}
        # concat_elu(t) that already exists for a block output t: the gate launch of the block that produced t wrote it as the
        # ce1 of the next block of its stack.  A block whose ONLY extra input is t (the down pass's horizontal blocks: t = the
        # vertical output of their level; the up pass's vertical blocks: t = a down-pass vertical output) reads that buffer
        # instead of launching its own concat_elu (same tensor, no dropout, no second source).  PM_PX_NO_CE_ALIAS=1: A/B
        ce_of: Dict[int, torch.Tensor] = {}
        no_alias = bool(os.environ.get("PM_PX_NO_CE_ALIAS"))

        def run_block(blk: _Block, input_x, extra_a=None, extra_b=None):
            n = blk.name
            ce_e_alias = None if (no_alias or extra_a is None or extra_b is not None) else ce_of.get(extra_a.data_ptr())
            ce1 = self.buf(f"{n}/ce1", sh(2 * F))
            if n not in have_ce1:
                ops.concat_elu_fwd(input_x, None, None, ce1)
            x1 = self.buf(f"{n}/x1", sh(F))
            if blk.linear is None:
                self._fwd(blk.conv1, ce1, x1)
            else:
                x1a = self.buf(f"{n}/x1a", sh(F))
                self._fwd(blk.conv1, ce1, x1a)
                if ce_e_alias is not None and extra_b is None and ce_e_alias.numel() == R * blk.linear.g.CI:
                    ce_e = ce_e_alias.view(R, blk.linear.g.CI)   # concat_elu(extra_a) already exists: see the down pass below
                else:
                    ce_e = self.buf(f"{n}/ce_e", (R, blk.linear.g.CI))
                    ops.concat_elu_fwd(extra_a, extra_b, None, ce_e)
                self._ce_e[blk.group] = ce_e
                self._fwd(blk.linear, ce_e, x1.view(R, F), res=x1a.view(R, F))
            drop = None
            if rate > 0.0:
                if dropout_masks is not None:
                    drop = dropout_masks[blk.group]
                elif ops.PhiloxDrop.usable(x1) and not os.environ.get("PM_DROPOUT_MASK_TENSOR"):
                    # the keep mask is drawn inside concat_elu forward AND backward (same Philox counters): it never exists
                    # in HBM (was: one mask launch + 12.8 MB written and read twice per gated block at the mnist size)
                    drop = ops.PhiloxDrop(rate, seed, step_dev, blk.group)
                else:
                    drop = self.buf(f"{n}/drop", sh(2 * F))
                    ops.dropout_mask(drop, rate, seed, step_dev, stream_id=blk.group)
            ce2 = self.buf(f"{n}/ce2", sh(2 * F))
            ops.concat_elu_fwd(x1, None, drop, ce2)
            y = self.buf(f"{n}/y", sh(2 * F))
            self._fwd(blk.conv2, ce2, y)
            out = self.buf(f"{n}/out", sh(F))
            hp = hproj[blk.group] if hproj is not None else None
            follower = nxt[blk.group]
            ce_next = self.buf(f"{follower.name}/ce1", sh(2 * F)) if (fuse_ce and follower is not None) else None
            if ce_next is not None and ops.gate_fwd_ce_ok(y, hp, input_x, out, ce_next):
                ops.gate_fwd_ce(y, hp, input_x, out, ce_next, P)
                have_ce1.add(follower.name)
                ce_of[out.data_ptr()] = ce_next
            else:
                ops.gate_fwd(y, hp, input_x, out, P)
            self._drops.append(drop)
            self._io.append((input_x, extra_a, extra_b))
            return out

        nres = self._num_resnet
        # The vertical stack never reads the horizontal one (a horizontal block reads the vertical block of its level), so
        # the two are two dependent chains of small launches: the vertical chain runs ahead on its own stream and every
        # horizontal block waits for the event of the vertical output it reads.  (One stream: 2 x the chain length.)
        main = torch.cuda.current_stream(value.device)
        two = self.two_streams and not os.environ.get("PM_PIXELCNN_ONE_STREAM")
        vs = main
        if two:
            vs = self.side_stream(value.device)
            ops.wait_stream(vs, main)                           # embeddings, v_init, conditional projections

        def vblock(k, blk, *args, **kw):
            if not two:
                return run_block(blk, *args, **kw)
            with torch.cuda.stream(vs):
                out = run_block(blk, *args, **kw)
                ops.record_event(self._vev[k], vs)
            return out

        V, Hs = [v], [h]
        for i in range(nres):                                   # down pass (:427-460)
            v = vblock(i, self.blocks[2 * i], V[-1])
            V.append(v)
            if two:
                ops.wait_event(main, self._vev[i])
            h = run_block(self.blocks[2 * i + 1], Hs[-1], extra_a=v)
            Hs.append(h)
        up_v, up_h = V.pop(), Hs.pop()
        for i in range(nres):                                   # up pass (:487-522)
            up_v = vblock(nres + i, self.blocks[2 * nres + 2 * i], up_v, extra_a=V.pop())
            if two:
                ops.wait_event(main, self._vev[nres + i])
            up_h = run_block(self.blocks[2 * nres + 2 * i + 1], up_h, extra_a=up_v, extra_b=Hs.pop())
        self._up_h = up_h
        self._slab_zeroed = None
        if two and training and not os.environ.get("PM_NO_GRAD_SLAB") and not os.environ.get("PM_PX_LATE_SLAB_ZERO"):
            # the backward pass's gradient accumulators (one slab) are zeroed here, on the second stream, which has finished
            # its chain while the main stream still runs the last horizontal block, the output layer and the loss
            n_zero = getattr(self, "_gslab_used", {}).get(B)
            if n_zero:
                if getattr(self, "_zev", None) is None:
                    self._zev = torch.cuda.Event()
                with torch.cuda.stream(vs):
                    ops.fill_zero(self.buf("grad_slab", (2 * len(self.blocks) + 4,) + sh(F))[:n_zero])
                    ops.record_event(self._zev, vs)
                self._slab_zeroed = (B, n_zero)
        x_out = self.buf("x_out", sh(F))
        ops.elu_fwd(up_h, x_out)
        logits = self.buf("logits", sh(K))
        self._fwd(self.out_conv, x_out, logits)
        self._emb, self._v0, self._h0 = emb, self.buf("v_init", sh(F)), self.buf("h_init", sh(F))
        return logits

    def side_stream(self, device) -> Optional["torch.cuda.Stream"]:
        """the second stream of the two-chain schedule (None when the network runs on one stream)"""
        if not self.two_streams or os.environ.get("PM_PIXELCNN_ONE_STREAM"):
            return None
        if getattr(self, "_vstream", None) is None:
            self._vstream = torch.cuda.Stream(device=device)
            self._vev = [torch.cuda.Event() for _ in range(2 * self._num_resnet)]
        return self._vstream

    def log_prob(self, value: torch.Tensor, training: bool = False, conditional_input: Optional[torch.Tensor] = None,
                 dropout_masks=None, seed: int = 0, step_dev=None) -> torch.Tensor:
        """reference pixel_cnn.py:53-63: Categorical(logits).log_prob(value) summed per example -> [B]."""
        H, W = self._event_shape
        logits = self.logits(value, training, conditional_input, dropout_masks, seed, step_dev)
        B = value.shape[0]
        self._lse = self.buf("lse", (B * H * W,))
        ll = self.buf("ll", (B,))
        ops.categorical_ll_fwd(logits.view(B * H * W, -1), value.reshape(-1), self._lse, ll, H * W)
        self._logits = logits
        return ll

    def sample(self, *, seed: int = 0, sample_shape=(), conditional_input: Optional[torch.Tensor] = None,
               gumbel: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference pixel_cnn.py:76-146 with conditioning: ancestral sampling in raster order, one full
        network evaluation per position.  Returns int32 [*sample_shape, B, H, W].  The categorical draw is
        the Gumbel-max trick (what jax.random.categorical does) with device Philox noise, or the explicit
        `gumbel` [H*W, B*n, K] (row b*n + s = sample s of conditioning vector b) in parity mode."""
        H, W = self._event_shape
        K, P = self._num_indices, H * W
        shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
        n = 1
        for v in shape:
            n *= int(v)
        if conditional_input is None:
            # the unconditional branch (:82-100): n chains from an all-zero grid; gumbel (parity mode): [H*W, n, K]
            if self._cond_dim is not None:
                raise ValueError("this network was built with a conditional_dim: conditional_input is required")
            dev = self.store.device
            x = torch.zeros((n, H, W), dtype=torch.int32, device=dev)
            noise = self.buf("gumbel_uncond", (n, K)) if gumbel is None else None
            for i in range(P):
                logits = self.logits(x, False, None)
                if gumbel is None:
                    ops.gumbel_fill(noise, seed, None, stream_id=i)
                ops.categorical_sample(logits.view(n * P, K), noise if gumbel is None else gumbel[i], x.view(-1), P, i)
            return x.view(*shape, H, W) if shape else x[0]
        B = conditional_input.shape[0]
        cond = conditional_input.reshape(B, -1).repeat_interleave(n, dim=0).contiguous()   # jnp.tile per vmap lane
        x = torch.zeros((B * n, H, W), dtype=torch.int32, device=cond.device)
        noise = self.buf("gumbel", (B * n, K)) if gumbel is None else None
        for i in range(P):
            logits = self.logits(x, False, cond)
            if gumbel is None:
                ops.gumbel_fill(noise, seed, None, stream_id=i)
            ops.categorical_sample(logits.view(B * n * P, K), noise if gumbel is None else gumbel[i], x.view(-1), P, i)
        out = x.view(B, n, H, W).permute(1, 0, 2, 3).contiguous()
        return out.view(*shape, B, H, W) if shape else out[0]

    # ------------------------------------------------------------------------------------------
    def _block_backward(self, blk: "_Block", gbuf, gbuf_h, gname, sh, dh_all, R: int, P: int, F: int, two: bool, from_h) -> None:
        """data gradients of one gated block (its weight gradients go to the grouped launches or the companion stream)"""
        n = blk.name
        input_x, extra_a, extra_b = self._io[blk.group]
        out = self.buf(f"{n}/out", sh(F))
        dout = gbuf(out, n)
        if two and out.data_ptr() in from_h:                                   # the horizontal reader's share, kept apart
            ops.axpy1(from_h[out.data_ptr()], dout)
        d_in = gbuf(input_x, gname(input_x))                                  # (+= dout, the residual branch: folded into the
        y = self.buf(f"{n}/y", sh(2 * F))                                      #  last concat_elu_bwd of the block)
        dy = self.buf(f"{n}/dy", sh(2 * F))
        hp = self._hproj[blk.group] if self._hproj is not None else None
        if dh_all is not None and ops.gate_bwd_rows_sum_ok(dout, R // P) and not os.environ.get("PM_NO_GATE_ROWS_SUM"):
            ops.gate_bwd_rows_sum(y, hp, dout, dy, dh_all[blk.group], P)     # dy and its sum over positions in one pass
        else:
            ops.gate_bwd(y, hp, dout, dy, P)
            if dh_all is not None:
                if self._dy_pending is not None:
                    self._dy_pending.append((blk.group, dy.view(R, 2 * F)))   # summed over positions by ONE launch at the tail
                else:
                    ops.rows_sum(dy.view(R, 2 * F), dh_all[blk.group], P)
        ce2 = self.buf(f"{n}/ce2", sh(2 * F))
        self._wg(blk.conv2, ce2, dy)
        dce2 = self.buf(f"{n}/dce2", sh(2 * F))
        self._dg(blk.conv2, dy, dce2)
        x1 = self.buf(f"{n}/x1", sh(F))
        dx1 = self.buf(f"{n}/dx1", sh(F))
        ops.concat_elu_bwd(x1, None, self._drops[blk.group], dce2, dx1, None, accumulate=False)
        if blk.linear is not None:
            ce_e = self._ce_e[blk.group]              # its own buffer, or the next vertical block's ce1 (forward pass)
            self._wg(blk.linear, ce_e, dx1.view(R, F))
            dce_e = self.buf(f"{n}/dce_e", (R, blk.linear.g.CI))
            self._dg(blk.linear, dx1.view(R, F), dce_e)
            # a horizontal block's extra_a is a VERTICAL tensor: with two chains its gradient goes to that tensor's own buffer
            to_h = two and blk.stack == "horizontal"
            da = gbuf_h(extra_a, gname(extra_a)) if to_h else gbuf(extra_a, gname(extra_a))
            db = gbuf(extra_b, gname(extra_b)) if extra_b is not None else None
            ops.concat_elu_bwd(extra_a, extra_b, None, dce_e, da, db, accumulate=True)
        ce1 = self.buf(f"{n}/ce1", sh(2 * F))
        self._wg(blk.conv1, ce1, dx1)
        dce1 = self.buf(f"{n}/dce1", sh(2 * F))
        self._dg(blk.conv1, dx1, dce1)
        ops.concat_elu_bwd(input_x, None, None, dce1, d_in, None, accumulate=True, add_a=dout)
        if self.store.reducer is not None and self.ws.wgrad_batch is None:   # data-parallel: this block's weight gradients are final
            self.ws.join_aux()
            self.store.grads_ready([f"{self.prefix}/{n}/{leaf}" for leaf in ("conv1", "linear", "conv2")])

    def backward(self, g_ll: torch.Tensor, overlap_tail: bool = False) -> Optional[torch.Tensor]:
        """g_ll [B] = d loss / d log_prob.  Accumulates parameter gradients; returns d loss / d
        conditional_input [B, cond_dim] (None without conditioning).  overlap_tail (grouped weight gradients + two chains
        only): the queued weight gradients are launched on the current stream once the blocks' data gradients are done, and
        the rest of the pass - with the returned tensor - is issued on `self.tail_stream`, which the caller must join."""
        self.tail_stream = None
        H, W = self._event_shape
        F, K = self._num_filters, self._num_indices
        B = self._B
        R, P = B * H * W, H * W
        sh = lambda c: (B, H, W, c)   # noqa: E731
        nres, G = self._num_resnet, len(self.blocks)
        dlogits = self.buf("dlogits", sh(K))
        ops.categorical_ll_bwd(self._logits.view(R, K), self._value.reshape(-1), self._lse, g_ll, dlogits.view(R, K), P)
        self._wg(self.out_conv, self.buf("x_out", sh(F)), dlogits)
        if self.ws.wgrad_batch is None:
            self.ws.join_aux()
            self.store.grads_ready([f"{self.prefix}/out_conv"])
        dx_out = self.buf("dx_out", sh(F))
        self._dg(self.out_conv, dlogits, dx_out)

        # gradient accumulators of every tensor that has several consumers: block outputs + the two inits.  They all have the
        # shape [B, H, W, F] and live in ONE slab that a single launch zeroes when the pass starts (was: one zero-fill launch
        # per accumulator, ~50 per pm_vqvae_mnist step, each in front of its first use on the dependent chains)
        grads: Dict[int, torch.Tensor] = {}
        cap = 2 * G + 4
        slab = self.buf("grad_slab", (cap,) + sh(F))
        used = [0]
        n_zero = getattr(self, "_gslab_used", {}).get(B, cap)       # entries the previous pass of this batch size handed out
        if os.environ.get("PM_NO_GRAD_SLAB"):                       # A/B switch: one zero-fill launch per accumulator, at first use
            n_zero = 0
        elif getattr(self, "_slab_zeroed", None) == (B, n_zero):    # done by the forward pass on the second stream
            ops.wait_event(torch.cuda.current_stream(dlogits.device), self._zev)
            self._slab_zeroed = None
        else:
            ops.fill_zero(slab[:n_zero])

        def take(t: torch.Tensor) -> torch.Tensor:
            assert tuple(t.shape) == sh(F) and used[0] < cap, (tuple(t.shape), used[0])
            gb = slab[used[0]]
            used[0] += 1
            if used[0] > n_zero:                                    # first pass only: beyond what the slab fill covered
                ops.fill_zero(gb)
            return gb

        def gbuf(t: torch.Tensor, name: str) -> torch.Tensor:
            key = t.data_ptr()
            if key not in grads:
                grads[key] = take(t)
            return grads[key]

        names = {self._v0.data_ptr(): "v_init", self._h0.data_ptr(): "h_init"}
        for blk in self.blocks:
            names[self.buf(f"{blk.name}/out", sh(F)).data_ptr()] = blk.name
        gname = lambda t: names[t.data_ptr()]   # noqa: E731

        ops.elu_bwd(self._up_h, dx_out, gbuf(self._up_h, gname(self._up_h)), accumulate=True)
        dh_all = self.buf("dhproj", (G, B, 2 * F)) if self._hproj is not None else None
        # blocks that do not take the fused gate + row-sum launch leave their dy here; one pm_rows_sum_multi at the tail
        self._dy_pending = None if os.environ.get("PM_NO_ROWS_SUM_MULTI") else []

        # Two chains again (see logits): a horizontal block's backward writes gradients of horizontal tensors and of ONE
        # vertical tensor (its extra_a); a vertical block only touches vertical tensors.  With the horizontal blocks'
        # contribution to a vertical tensor kept in its own buffer, the horizontal chain never waits for the vertical one: it
        # runs ahead on the main stream, the vertical chain follows on its stream and waits, block by block, for the event
        # of the horizontal block that read its output.  (Weight gradients are collected for the grouped launches at the
        # end, so only data gradients are on the chains; without the batch the single-stream order is kept.)
        main = torch.cuda.current_stream(dlogits.device)
        two = (self.two_streams and self.ws.wgrad_batch is not None and getattr(self, "_vstream", None) is not None
               and not os.environ.get("PM_PIXELCNN_ONE_STREAM"))
        vs = self._vstream if two else main
        from_h: Dict[int, torch.Tensor] = {}

        def gbuf_h(t: torch.Tensor, name: str) -> torch.Tensor:
            key = t.data_ptr()
            if key not in from_h:
                from_h[key] = take(t)
            return from_h[key]

        if two:
            if getattr(self, "_hev", None) is None:
                self._hev = [torch.cuda.Event() for _ in range(G)]
            ops.wait_stream(vs, main)

        fs = None
        for blk in reversed(self.blocks):
            vertical = blk.stack == "vertical"
            if (two and fs is None and blk.name.startswith("down")
                    and (os.environ.get("PM_PIXELCNN_MID_FLUSH") or self.store.reducer is not None
                         or getattr(self, "early_update", None) is not None)):
                # (measured neutral on one GPU: celeb_a 1638 -> 1622 img/s, mnist 20.3k -> 20.5k; off unless asked for.  Data-
                # parallel: ON - the up pass's weights (half of the 140 / 273 MB gradient) are final after this flush, so
                # their buckets are all-reduced beside the whole down pass instead of after the last launch)
                # every up-pass block has left its operands in HBM: their grouped weight gradients (throughput-bound) start
                # now on a third stream, beside the two latency-bound chains of the down pass
                if getattr(self, "_fstream", None) is None:
                    self._fstream = torch.cuda.Stream(device=dlogits.device)
                fs = self._fstream
                ops.wait_stream(fs, main)
                ops.wait_stream(fs, vs)
                with torch.cuda.stream(fs):
                    self.ws.wgrad_batch.flush()
                    if getattr(self, "early_update", None) is not None:
                        # one GPU: the up pass's WEIGHTS (half of the network) have their final gradients and no later launch
                        # of this step reads them - the train step runs their optimizer update here, on this third stream,
                        # beside the two latency-bound chains of the down pass (engine._PlannedStep._early_adam)
                        self.early_update([f"{self.prefix}/{b_.name}" for b_ in self.blocks if not b_.name.startswith("down")])
            if two and vertical:
                # its output was read (as extra_a) by the horizontal block right behind it in execution order
                ops.wait_event(vs, self._hev[blk.group + 1])
                with torch.cuda.stream(vs):
                    self._block_backward(blk, gbuf, gbuf_h, gname, sh, dh_all, R, P, F, two, from_h)
                continue
            self._block_backward(blk, gbuf, gbuf_h, gname, sh, dh_all, R, P, F, two, from_h)
            if two:
                ops.record_event(self._hev[blk.group], main)
        if two:
            ops.wait_stream(main, vs)
        if fs is not None:
            ops.wait_stream(main, fs)
        dv0, dh0 = gbuf(self._v0, "v_init"), gbuf(self._h0, "h_init")
        emb = self._emb
        self._wg(self.v_init, emb, dv0)
        self._wg(self.h_up, emb, dh0)
        self._wg(self.h_left, emb, dh0)
        if overlap_tail and two:
            self.tail_stream = vs
            ops.wait_stream(vs, main)
            self.ws.wgrad_batch.flush()                      # main: every convolution's weight gradient of this network
            with torch.cuda.stream(vs):
                return self._backward_tail(dv0, dh0, dh_all, used, sh, B, F, G)
        return self._backward_tail(dv0, dh0, dh_all, used, sh, B, F, G)

    def _backward_tail(self, dv0, dh0, dh_all, used, sh, B: int, F: int, G: int) -> Optional[torch.Tensor]:
        """input convolutions' data gradients -> embedding table gradient; conditional projection's gradients"""
        de1, de2 = self.buf("demb1", sh(F)), self.buf("demb2", sh(F))
        self._dg(self.v_init, dv0, de1)
        self._dg(self.h_up, dh0, de2, res=de1)
        self._dg(self.h_left, dh0, de1, res=de2)
        ops.embed_bwd(self._value.reshape(-1), de1, self.G("embed/embeddings"))
        if getattr(self, "_gslab_used", None) is None:
            self._gslab_used = {}
        self._gslab_used[B] = used[0]
        if dh_all is None:
            return None
        pend = self._dy_pending or []
        if pend:
            P_ = self._event_shape[0] * self._event_shape[1]
            pend.sort(key=lambda t: t[0])
            if len(pend) == G and [g_ for g_, _ in pend] == list(range(G)) and G <= ops.ROWS_SUM_MULTI_MAX:
                ops.rows_sum_multi([t for _, t in pend], dh_all, P_)
            else:
                for g_, t in pend:
                    ops.rows_sum(t, dh_all[g_], P_)
            self._dy_pending = []
        gw, gb_ = self._cond_params("g")
        kw = self._cond_kw(B)
        ops.layer_wgrad(self.g_cond, self._cond, dh_all, gw, gb_, bf16=False, **kw)
        dcond_g = self.buf("dcond_groups", (G, B, self._cond_dim))
        cw, _ = self._cond_params("p")
        dkw = dict(kw)
        dkw["in_gs"], dkw["out_gs"] = B * self._cond_dim, B * 2 * F      # layer_dgrad swaps them: dy groups -> dx groups
        ops.layer_dgrad(self.g_cond, dh_all, cw, dcond_g, **dkw)
        dcond = self.buf("dcond", (B, self._cond_dim))
        ops.groups_sum(dcond_g, dcond, G)
        return dcond
