"""Host-side mirrors of posterior_matching/models/networks.py of the reference.

Same class names, constructor arguments and registry (`get_network`, reference
networks.py:138-162); the arithmetic runs in libpmhip.so.  Because there is no autodiff here,
every network also has an explicit `backward`.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from ..ops import ACT_LEAKY, ACT_NONE, ACT_RELU, LayerGeom
from .core import Feat, Module, ParamStore


class ConvEncoder(Module):
    """reference networks.py:9-38: SAME convs (last one VALID), leaky_relu after every layer."""

    def __init__(self, conv_layers: Sequence[Tuple[int, int, int]], name: Optional[str] = None):
        super().__init__(name)
        self._conv_layers = [tuple(l) for l in conv_layers]

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        if len(in_shape) != 3:
            raise ValueError(f"ConvEncoder expects [H, W, C] inputs, got {tuple(in_shape)}")  # chex.assert_rank(x, 4)
        self.attach(store, prefix)
        h, w, c = in_shape
        self.geoms: List[LayerGeom] = []
        self._ws = []
        n = len(self._conv_layers)
        for i, (f, k, s) in enumerate(self._conv_layers):
            g = LayerGeom.conv(h, w, c, f, k, s, "VALID" if i == n - 1 else "SAME")
            store.add(f"{prefix}/conv_{i}/w", g.weight_shape, fan_in=k * k * c)
            store.add(f"{prefix}/conv_{i}/b", (f,))
            self.geoms.append(g)
            self._ws.append((store.request_split(f"{prefix}/conv_{i}/w", g, "fwd"),
                             store.request_split(f"{prefix}/conv_{i}/w", g, "dgrad")))
            h, w, c = g.OH, g.OW, f
        return (h, w, c)

    def __call__(self, x: Feat, is_training: bool = False) -> Feat:
        assert x.in_act == ACT_NONE
        B = x.t.shape[0]
        self._x = x
        self._outs = []
        h = x.t
        for i, g in enumerate(self.geoms):
            out = self.buf(f"out_{i}", (B, g.OH, g.OW, g.CO))
            ops.layer_forward(g, h, self.P(f"conv_{i}/w"), self.P(f"conv_{i}/b"), out, out_act=ACT_LEAKY,
                              wsplit=self.store.split_view(self._ws[i][0]))
            self._outs.append(out)
            h = out
        return Feat(h, ACT_NONE, ACT_LEAKY)

    def backward(self, dpre: torch.Tensor, need_input_grad: bool = False) -> Optional[torch.Tensor]:
        """dpre: gradient w.r.t. the last layer's pre-activation."""
        B = dpre.shape[0]
        for i in reversed(range(len(self.geoms))):
            g = self.geoms[i]
            inp = self._outs[i - 1] if i > 0 else self._x.t
            lent = self.ws.wgrad_stream
            if i in getattr(self, "own_stream_wgrad", ()):     # this layer's weight gradient stays on the data-gradient stream
                self.ws.wgrad_stream = None
            self.wgrad(g, inp, dpre, self.G(f"conv_{i}/w"), self.G(f"conv_{i}/b"))
            self.ws.wgrad_stream = lent
            if i > 0:
                dprev = self.buf(f"dpre_{i - 1}", (B, g.IH, g.IW, g.CI))
                ops.layer_dgrad(g, dpre, self.P(f"conv_{i}/w"), dprev, aux=inp, aux_act=ACT_LEAKY,
                                wsplit=self.store.split_view(self._ws[i][1]))
                dpre = dprev
            elif need_input_grad:
                dprev = self.buf("dx", (B, g.IH, g.IW, g.CI))
                ops.layer_dgrad(g, dpre, self.P(f"conv_{i}/w"), dprev, aux=self._x.t, aux_act=self._x.grad_act,
                                wsplit=self.store.split_view(self._ws[i][1]))
                return dprev
        return None


class ConvDecoder(Module):
    """reference networks.py:41-72: z -> [B,1,1,Z] -> transposed convs (first VALID, rest SAME),
    leaky_relu after EVERY layer including the last."""

    def __init__(self, conv_layers: Sequence[Tuple[int, int, int]], name: Optional[str] = None):
        super().__init__(name)
        self._conv_layers = [tuple(l) for l in conv_layers]

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        if len(in_shape) != 1:
            raise ValueError(f"ConvDecoder expects [Z] inputs, got {tuple(in_shape)}")  # chex.assert_rank(x, 2)
        self.attach(store, prefix)
        h, w, c = 1, 1, in_shape[0]
        self.geoms = []
        self._ws = []
        for i, (f, k, s) in enumerate(self._conv_layers):
            g = LayerGeom.conv_t(h, w, c, f, k, s, "VALID" if i == 0 else "SAME")
            store.add(f"{prefix}/conv_t_{i}/w", g.weight_shape, fan_in=k * k * c)
            store.add(f"{prefix}/conv_t_{i}/b", (f,))
            self.geoms.append(g)
            self._ws.append((store.request_split(f"{prefix}/conv_t_{i}/w", g, "fwd"),
                             store.request_split(f"{prefix}/conv_t_{i}/w", g, "dgrad")))
            h, w, c = g.OH, g.OW, f
        return (h, w, c)

    def __call__(self, x: Feat, is_training: bool = False) -> Feat:
        B = x.t.shape[0]
        self._x = x
        self._outs = []
        h = x.t
        for i, g in enumerate(self.geoms):
            out = self.buf(f"out_{i}", (B, g.OH, g.OW, g.CO))
            tmp = self.buf(f"taps_{i}", (B, g.IH, g.IW, g.k * g.k)) if (g.CO == 1 and g.s == 1 and i > 0) else None
            ops.layer_forward(g, h, self.P(f"conv_t_{i}/w"), self.P(f"conv_t_{i}/b"), out,
                              in_act=x.in_act if i == 0 else ACT_NONE, out_act=ACT_LEAKY,
                              wsplit=self.store.split_view(self._ws[i][0]), tmp=tmp)
            self._outs.append(out)
            h = out
        return Feat(h, ACT_NONE, ACT_LEAKY)

    def backward(self, dpre: torch.Tensor, need_input_grad: bool = True, lend=None) -> Optional[torch.Tensor]:
        """lend = (stream, n): the weight gradients of layers 0 .. n-1 (the last n of this pass) are queued on `stream`"""
        B = dpre.shape[0]
        # the bias gradients (column sums of every layer's dpre, read only by the optimizer) wait for ONE launch at the end of
        # this pass: the dpre_i buffers are per layer and stay untouched until then
        ops.colsum_defer_begin()
        try:
            return self._backward(dpre, need_input_grad, lend, B)
        finally:
            ops.colsum_defer_flush()

    def _backward(self, dpre, need_input_grad, lend, B):
        for i in reversed(range(len(self.geoms))):
            g = self.geoms[i]
            inp = self._outs[i - 1] if i > 0 else self._x.t
            if lend is not None and i < lend[1]:
                self.ws.wgrad_stream = lend[0]
            # the bias gradient (column sums of dpre) comes out of the data-gradient launch when that one stages the dpre images
            # in LDS anyway (ops.dgrad_insum_ok): no pm_colsum launch, no second read of dpre
            wsd = self.store.split_view(self._ws[i][1])
            db_in_dgrad = i > 0 and ops.dgrad_insum_ok(g, B, wsd)
            self.wgrad(g, inp, dpre, self.G(f"conv_t_{i}/w"), None if db_in_dgrad else self.G(f"conv_t_{i}/b"),
                            in_act=ACT_NONE if i > 0 else self._x.in_act)
            if lend is not None:
                self.ws.wgrad_stream = None
            if i > 0:
                dprev = self.buf(f"dpre_{i - 1}", (B, g.IH, g.IW, g.CI))
                ops.layer_dgrad(g, dpre, self.P(f"conv_t_{i}/w"), dprev, aux=inp, aux_act=ACT_LEAKY, wsplit=wsd,
                                in_colsum=self.G(f"conv_t_{i}/b") if db_in_dgrad else None)
                dpre = dprev
            elif need_input_grad:
                dz = self.buf("dz", (B, g.CI))
                ops.layer_dgrad(g, dpre, self.P(f"conv_t_{i}/w"), dz, aux=self._x.t, aux_act=self._x.grad_act,
                                wsplit=self.store.split_view(self._ws[i][1]))
                return dz
        return None


class ResidualMLP(Module):
    """reference networks.py:75-135 (activation relu): Linear -> N x [act, Linear, act, dropout,
    Linear, +residual] -> act.  The pre-activation h is what is stored; consumers apply relu on load."""

    def __init__(self, residual_blocks: int = 2, hidden_units: int = 256, activation: Any = "relu",
                 activate_final: bool = True, dropout: float = 0.0, w_init: Any = None, layer_norm: bool = False,
                 name: Optional[str] = None):
        super().__init__(name)
        if activation not in ("relu", None) and getattr(activation, "__name__", "") != "relu":
            raise NotImplementedError("ResidualMLP: only the reference default activation (relu) has a HIP path")
        if w_init is not None:
            raise NotImplementedError("custom w_init")
        self._residual_blocks = residual_blocks
        self._hidden_units = hidden_units
        self._activate_final = activate_final
        self._layer_norm, self._dropout = bool(layer_norm), float(dropout or 0.0)
        # parity mode: explicit keep masks (one [rows, hidden] tensor per block, already scaled by 1 / (1 - rate)) replace
        # the device Philox draws of hk.dropout(hk.next_rng_key(), ...)
        self.dropout_masks = None
        self.dropout_seed, self.dropout_step_dev, self.dropout_stream_base = 0, None, 0

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        if len(in_shape) != 1:
            raise ValueError(f"ResidualMLP expects [D] inputs, got {tuple(in_shape)}")  # chex.assert_rank(x, 2)
        self.attach(store, prefix)
        fin, hu = int(in_shape[0]), self._hidden_units
        self.g_in = LayerGeom.dense(fin, hu)
        self.g_hid = LayerGeom.dense(hu, hu)
        store.add(f"{prefix}/linear_0/w", (fin, hu), fan_in=fin)
        store.add(f"{prefix}/linear_0/b", (hu,))
        self._ws = {"linear_0": (store.request_split(f"{prefix}/linear_0/w", self.g_in, "fwd"),
                                 store.request_split(f"{prefix}/linear_0/w", self.g_in, "dgrad"))}
        for k in range(self._residual_blocks):
            for j in range(2):
                store.add(f"{prefix}/block_{k}/linear_{j}/w", (hu, hu), fan_in=hu)
                store.add(f"{prefix}/block_{k}/linear_{j}/b", (hu,))
                self._ws[f"block_{k}/linear_{j}"] = (
                    store.request_split(f"{prefix}/block_{k}/linear_{j}/w", self.g_hid, "fwd"),
                    store.request_split(f"{prefix}/block_{k}/linear_{j}/w", self.g_hid, "dgrad"))
        return (hu,)

    def _wsf(self, name):
        return self.store.split_view(self._ws[name][0])

    def _wsd(self, name):
        return self.store.split_view(self._ws[name][1])

    # The 2*N hidden linears (hu -> hu) all see [rows, hu] operands and their weights / biases are adjacent in the flat
    # parameter buffer, so their weight gradients run as ONE grouped launch at the end of backward (4 launches of 256
    # workgroups -> 1 launch of 1024: the per-launch floor is paid once and the chip holds 4 waves per SIMD).  For that
    # the layer inputs and the output gradients live in two strided buffers:
    #   xs[2k] = h_k, xs[2k+1] = u_k          (inputs of block_k/linear_0, block_k/linear_1; relu applied on load)
    #   dys[2k] = du_k, dys[2k+1] = dh_{k+1}  (gradients w.r.t. the outputs of the same two layers)
    def _grouped(self) -> bool:
        if self._residual_blocks == 0:
            return False
        if getattr(self, "_grouped_ok", None) is None:
            hu, off = self._hidden_units, self.store.offsets
            names = [f"{self.prefix}/block_{k}/linear_{j}" for k in range(self._residual_blocks) for j in range(2)]
            w0, b0 = off[names[0] + "/w"][0], off[names[0] + "/b"][0]
            self._grouped_ok = all(off[n + "/w"][0] == w0 + i * hu * hu and off[n + "/b"][0] == b0 + i * hu
                                   for i, n in enumerate(names))
        return self._grouped_ok

    def _pair_fused(self) -> bool:
        """hidden width 256 on the bf16x3 path: a block's two layers run as one launch (pm_mlp_pair_bf16)"""
        import os

        if os.environ.get("PM_NO_MLP_PAIR") or self._hidden_units != 256 or self._residual_blocks == 0:
            return False
        return all(self._wsf(f"block_{k}/linear_{j}") is not None and self._wsd(f"block_{k}/linear_{j}") is not None
                   for k in range(self._residual_blocks) for j in range(2))

    def _chain_ok(self, rows: int) -> bool:
        import os

        return not os.environ.get("PM_NO_MLP_CHAIN") and ops.mlp_chain_ok(rows, self._hidden_units)

    def out_grad_buffer(self, rows: int) -> torch.Tensor:
        """where the caller should write the gradient w.r.t. this network's output so that backward() needs no copy"""
        nb, hu = self._residual_blocks, self._hidden_units
        if nb == 0:
            return self.buf("dh_out", (rows, hu))
        return self.buf("dys_all", (2 * nb, rows, hu))[2 * nb - 1]

    def _plain(self, is_training: bool) -> bool:
        """the relu / no-LayerNorm / no-dropout form (pm_vae_gas, the AR-GMM networks): fused kernels below"""
        return not self._layer_norm and not (self._dropout > 0.0 and is_training)

    def _call_general(self, x: Feat, is_training: bool) -> Feat:
        """networks.py:111-135 with LayerNorm after every Linear and / or dropout between a block's two Linears
        (configs/pm_vae_miniboone.py:32-39, pm_vae_bsds.py:32-37): the Linears are the GEMM engine's, LayerNorm and
        relu -> dropout are row-wise kernels (csrc/pm_mlp.hip).  Stored per block for the backward pass: h (block input,
        pre-activation), n1 = LN(t1), d = relu(n1) * mask, n2 = LN(t2), the two rstd vectors."""
        rows, hu, nb, ln = x.t.shape[0], self._hidden_units, self._residual_blocks, self._layer_norm
        rate = self._dropout if is_training else 0.0
        self._x, self._general = x, True
        st = self._st = []
        t0 = self.buf("g/t0", (rows, hu))
        ops.layer_forward(self.g_in, x.t, self.P("linear_0/w"), self.P("linear_0/b"), t0, in_act=x.in_act,
                          wsplit=self._wsf("linear_0"))
        if ln:
            h, self._rstd0 = self.buf("g/h0", (rows, hu)), self.buf("g/rstd0", (rows,))
            ops.layernorm_fwd(t0, None, h, None, self._rstd0)
        else:
            h = t0
        self._h = [h]
        for k in range(nb):
            t1 = self.buf(f"g/t1_{k}", (rows, hu))
            ops.layer_forward(self.g_hid, h, self.P(f"block_{k}/linear_0/w"), self.P(f"block_{k}/linear_0/b"), t1,
                              in_act=ACT_RELU, wsplit=self._wsf(f"block_{k}/linear_0"))
            n1, rstd1 = t1, None
            if ln:
                n1, rstd1 = self.buf(f"g/n1_{k}", (rows, hu)), self.buf(f"g/rstd1_{k}", (rows,))
                ops.layernorm_fwd(t1, None, n1, None, rstd1)
            mask, d = None, None
            if rate > 0.0:
                if self.dropout_masks is not None:
                    mask = self.dropout_masks[k]
                else:
                    mask = self.buf(f"g/mask_{k}", (rows, hu))
                    ops.dropout_mask(mask, rate, self.dropout_seed, self.dropout_step_dev, stream_id=self.dropout_stream_base + k)
                d = self.buf(f"g/d_{k}", (rows, hu))
                ops.relu_mask_fwd(n1, mask, d)
            hn = self.buf(f"g/h_{k + 1}", (rows, hu))
            n2, rstd2 = None, None
            lin1_in, lin1_act = (d, ACT_NONE) if d is not None else (n1, ACT_RELU)
            if ln:
                t2 = self.buf(f"g/t2_{k}", (rows, hu))
                ops.layer_forward(self.g_hid, lin1_in, self.P(f"block_{k}/linear_1/w"), self.P(f"block_{k}/linear_1/b"), t2,
                                  in_act=lin1_act, wsplit=self._wsf(f"block_{k}/linear_1"))
                n2, rstd2 = self.buf(f"g/n2_{k}", (rows, hu)), self.buf(f"g/rstd2_{k}", (rows,))
                ops.layernorm_fwd(t2, h, n2, hn, rstd2)                       # h += LN(t2)
            else:
                ops.layer_forward(self.g_hid, lin1_in, self.P(f"block_{k}/linear_1/w"), self.P(f"block_{k}/linear_1/b"), hn,
                                  in_act=lin1_act, res=h, wsplit=self._wsf(f"block_{k}/linear_1"))
            st.append((n1, rstd1, mask, d, n2, rstd2))
            self._h.append(hn)
            h = hn
        if self._activate_final:
            return Feat(h, ACT_RELU, ACT_RELU)
        return Feat(h, ACT_NONE, ACT_NONE)

    def _backward_general(self, dh: torch.Tensor, need_input_grad: bool) -> Optional[torch.Tensor]:
        rows, hu, nb, ln = dh.shape[0], self._hidden_units, self._residual_blocks, self._layer_norm
        for k in reversed(range(nb)):
            n1, rstd1, mask, d, n2, rstd2 = self._st[k]
            h = self._h[k]
            dt2 = dh
            if ln:
                dt2 = self.buf(f"g/dt2_{k}", (rows, hu))
                ops.layernorm_bwd(n2, rstd2, dh, dt2)
            lin1_in, lin1_act = (d, ACT_NONE) if d is not None else (n1, ACT_RELU)
            self.wgrad(self.g_hid, lin1_in, dt2, self.G(f"block_{k}/linear_1/w"), self.G(f"block_{k}/linear_1/b"), in_act=lin1_act)
            dn1 = self.buf(f"g/dn1_{k}", (rows, hu))
            if d is not None:
                dd = self.buf(f"g/dd_{k}", (rows, hu))
                ops.layer_dgrad(self.g_hid, dt2, self.P(f"block_{k}/linear_1/w"), dd, wsplit=self._wsd(f"block_{k}/linear_1"))
                ops.relu_mask_bwd(n1, mask, dd, dn1)
            else:
                ops.layer_dgrad(self.g_hid, dt2, self.P(f"block_{k}/linear_1/w"), dn1, aux=n1, aux_act=ACT_RELU,
                                wsplit=self._wsd(f"block_{k}/linear_1"))
            dt1 = dn1
            if ln:
                dt1 = self.buf(f"g/dt1_{k}", (rows, hu))
                ops.layernorm_bwd(n1, rstd1, dn1, dt1)
            self.wgrad(self.g_hid, h, dt1, self.G(f"block_{k}/linear_0/w"), self.G(f"block_{k}/linear_0/b"), in_act=ACT_RELU)
            dprev = self.buf(f"g/dh_{k}", (rows, hu))
            ops.layer_dgrad(self.g_hid, dt1, self.P(f"block_{k}/linear_0/w"), dprev, aux=h, aux_act=ACT_RELU, res=dh,
                            wsplit=self._wsd(f"block_{k}/linear_0"))
            dh = dprev
        if ln:
            dt0 = self.buf("g/dt0", (rows, hu))
            ops.layernorm_bwd(self._h[0], self._rstd0, dh, dt0)
            dh = dt0
        self.wgrad(self.g_in, self._x.t, dh, self.G("linear_0/w"), self.G("linear_0/b"), in_act=self._x.in_act)
        if need_input_grad:
            dx = self.buf("dx", (rows, self.g_in.CI))
            ops.layer_dgrad(self.g_in, dh, self.P("linear_0/w"), dx, aux=self._x.t, aux_act=self._x.grad_act,
                            wsplit=self._wsd("linear_0"))
            return dx
        return None

    def __call__(self, x: Feat, is_training: bool = False) -> Feat:
        if not self._plain(is_training):
            return self._call_general(x, is_training)
        self._general = False
        rows, hu, nb = x.t.shape[0], self._hidden_units, self._residual_blocks
        self._x = x
        xs = self.buf("xs_all", (2 * nb + 1, rows, hu))
        h = xs[0]
        ops.layer_forward(self.g_in, x.t, self.P("linear_0/w"), self.P("linear_0/b"), h, in_act=x.in_act,
                          wsplit=self._wsf("linear_0"))
        self._h, self._u = [h], []
        fused = self._pair_fused()
        if fused and self._chain_ok(rows):
            # up to two blocks (four layers) per launch, activations of a 64-row tile kept in LDS (pm_mlp_chain_bf16)
            for k0 in range(0, nb, 2):
                ks = list(range(k0, min(k0 + 2, nb)))
                names = [f"block_{k}/linear_{j}" for k in ks for j in range(2)]
                outs = [xs[2 * k + 1 + j] for k in ks for j in range(2)]
                ops.mlp_chain_bf16(xs[2 * k0], [self._wsf(n) for n in names], [self.P(n + "/b") for n in names], None, outs,
                                   ACT_RELU, ACT_RELU, ACT_NONE)
            for k in range(nb):
                self._u.append(xs[2 * k + 1])
                self._h.append(xs[2 * k + 2])
            h = xs[2 * nb]
            nb = 0                                      # nothing left for the per-block loop below
        for k in range(nb):
            u = xs[2 * k + 1]
            hn = xs[2 * k + 2]
            if fused:      # both layers of the block in one launch (csrc/pm_mlp.hip)
                ops.mlp_pair_bf16(h, self._wsf(f"block_{k}/linear_0"), self._wsf(f"block_{k}/linear_1"),
                                  self.P(f"block_{k}/linear_0/b"), self.P(f"block_{k}/linear_1/b"), None, None, u, hn,
                                  ACT_RELU, ACT_RELU, ACT_NONE, ACT_NONE)
                self._u.append(u)
                self._h.append(hn)
                h = hn
                continue
            ops.layer_forward(self.g_hid, h, self.P(f"block_{k}/linear_0/w"), self.P(f"block_{k}/linear_0/b"), u,
                              in_act=ACT_RELU, wsplit=self._wsf(f"block_{k}/linear_0"))
            ops.layer_forward(self.g_hid, u, self.P(f"block_{k}/linear_1/w"), self.P(f"block_{k}/linear_1/b"), hn,
                              in_act=ACT_RELU, res=h, wsplit=self._wsf(f"block_{k}/linear_1"))
            self._u.append(u)
            self._h.append(hn)
            h = hn
        self._xs = xs
        if self._activate_final:
            return Feat(h, ACT_RELU, ACT_RELU)
        return Feat(h, ACT_NONE, ACT_NONE)

    def backward(self, dh: torch.Tensor, need_input_grad: bool = False) -> Optional[torch.Tensor]:
        """dh: gradient w.r.t. the stored pre-activation h of the last block."""
        if getattr(self, "_general", False):
            return self._backward_general(dh.reshape(dh.shape[0], self._hidden_units), need_input_grad)
        rows, hu, nb = dh.shape[0], self._hidden_units, self._residual_blocks
        grouped = self._grouped()
        if nb > 0:
            dys = self.buf("dys_all", (2 * nb, rows, hu))
            if grouped and dh.data_ptr() != dys[2 * nb - 1].data_ptr():
                ops.copy_cols(dh.reshape(rows, hu), dys[2 * nb - 1], 0)   # callers that did not use out_grad_buffer()
                dh = dys[2 * nb - 1]
        nloop = nb
        if grouped and self._pair_fused() and self._chain_ok(rows):
            # all data gradients of up to two blocks per launch; outputs land in dys (du_k, dh_k) for the grouped wgrad
            dh0 = self.buf("dh_0", (rows, hu))
            k_hi = nb - 1
            while k_hi >= 0:
                ks = [k for k in (k_hi, k_hi - 1) if k >= 0]
                names = [f"block_{k}/linear_{j}" for k in ks for j in (1, 0)]
                auxs = [t for k in ks for t in (self._u[k], self._h[k])]
                outs = [t for k in ks for t in (dys[2 * k], dys[2 * k - 1] if k > 0 else dh0)]
                ops.mlp_chain_bf16(dh, [self._wsd(n) for n in names], None, auxs, outs, ACT_NONE, ACT_NONE, ACT_RELU)
                dh = outs[-1]
                k_hi -= 2
            nloop = 0
        for k in reversed(range(nloop)):
            h, u = self._h[k], self._u[k]
            if not grouped:
                self.wgrad(self.g_hid, u, dh, self.G(f"block_{k}/linear_1/w"), self.G(f"block_{k}/linear_1/b"),
                           in_act=ACT_RELU)
            du = dys[2 * k]
            dprev = dys[2 * k - 1] if k > 0 else self.buf("dh_0", (rows, hu))
            if grouped and self._pair_fused():      # both data gradients of the block in one launch
                ops.mlp_pair_bf16(dh, self._wsd(f"block_{k}/linear_1"), self._wsd(f"block_{k}/linear_0"), None, None, u, h,
                                  du, dprev, ACT_NONE, ACT_NONE, ACT_RELU, ACT_RELU)
                dh = dprev
                continue
            ops.layer_dgrad(self.g_hid, dh, self.P(f"block_{k}/linear_1/w"), du, aux=u, aux_act=ACT_RELU,
                            wsplit=self._wsd(f"block_{k}/linear_1"))
            if not grouped:
                self.wgrad(self.g_hid, h, du, self.G(f"block_{k}/linear_0/w"), self.G(f"block_{k}/linear_0/b"),
                           in_act=ACT_RELU)
            ops.layer_dgrad(self.g_hid, du, self.P(f"block_{k}/linear_0/w"), dprev, aux=h, aux_act=ACT_RELU, res=dh,
                            wsplit=self._wsd(f"block_{k}/linear_0"))
            dh = dprev
        if grouped:
            self.wgrad(self.g_hid, self._xs, dys, self.G("block_0/linear_0/w"), self.G("block_0/linear_0/b"),
                       in_act=ACT_RELU, B=rows, groups=2 * nb, in_gs=rows * hu, out_gs=rows * hu, w_gs=hu * hu, bias_gs=hu)
        self.wgrad(self.g_in, self._x.t, dh, self.G("linear_0/w"), self.G("linear_0/b"), in_act=self._x.in_act)
        if need_input_grad:
            dx = self.buf("dx", (rows, self.g_in.CI))
            ops.layer_dgrad(self.g_in, dh, self.P("linear_0/w"), dx, aux=self._x.t, aux_act=self._x.grad_act,
                            wsplit=self._wsd("linear_0"))
            return dx
        return None


_NETWORKS = {
    "ConvEncoder": ConvEncoder,
    "ConvDecoder": ConvDecoder,
    "ResidualMLP": ResidualMLP,
}


def get_network(network_type: str, network_config: Optional[Dict[str, Any]] = None, name: Optional[str] = None):
    """reference networks.py:145-162 (KeyError on unknown names, like the reference's dict lookup)."""
    network_config = dict(network_config or {})
    return _NETWORKS[network_type](**network_config, name=name)
