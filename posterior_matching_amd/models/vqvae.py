"""Host-side mirror of posterior_matching/models/vqvae.py of the reference (VQ-VAE, stage 1).

Same class names and constructor arguments (reference vqvae.py:13-266); the arithmetic runs in
libpmhip.so.  haiku keeps the codebook and its EMA accumulators in *state* (updated by the forward
pass when is_training, never by the optimizer) - here that is `VQVAE.state`, a dict of device
tensors next to the flat parameter buffer.  There is no autodiff: `VQVAE.backward()` walks the
buffers the forward left in HBM.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Optional, Tuple

import math

import numpy as np
import torch

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, AUX_AFTER_RES, LayerGeom
from .core import Feat, Module, ParamStore, Workspace

VQ_EPSILON = 1e-5   # hk.nets.VectorQuantizerEMA(epsilon=1e-5)


class _Conv:
    """One hk.Conv2D / hk.Conv2DTranspose of a conv-residual network: geometry, parameter names and
    the pre-split bf16 weight handles."""

    def __init__(self, store: ParamStore, prefix: str, name: str, geom: LayerGeom):
        base = f"{prefix}/{name}" if prefix else name
        self.g, self.w, self.b = geom, f"{base}/w", f"{base}/b"
        # haiku: stddev = 1/sqrt(fan_in); conv fan_in = kh*kw*Cin, and the same product for the
        # transposed conv's [kh,kw,Cout,Cin] weight
        store.add(self.w, geom.weight_shape, fan_in=geom.k * geom.k * geom.CI)
        store.add(self.b, (geom.CO,))
        self.ws_f = store.request_split(self.w, geom, "fwd")
        self.ws_d = store.request_split(self.w, geom, "dgrad")


class _ConvResidualNet(Module):
    """Shared machinery of ConvResidualEncoder / ConvResidualDecoder: layer calls and the
    ConvResidualStack (reference vqvae.py:133-181)."""

    def __init__(self, hidden_units, residual_blocks, residual_hidden_units, name=None):
        super().__init__(name)
        if residual_blocks < 1:
            raise NotImplementedError("ConvResidualStack with residual_blocks=0 has no HIP path")
        self._hidden_units = hidden_units
        self._residual_blocks = residual_blocks
        self._residual_hidden_units = residual_hidden_units

    # -- layer helpers ------------------------------------------------------------------------
    def _fwd(self, L: _Conv, x, out, in_act=ACT_NONE, out_act=ACT_NONE, res=None):
        ops.layer_forward(L.g, x, self.store.p[L.w], self.store.p[L.b], out, in_act=in_act, out_act=out_act, res=res,
                          wsplit=self.store.split_view(L.ws_f))

    def _wg(self, L: _Conv, x, dy, in_act=ACT_NONE):
        self.wgrad(L.g, x, dy, self.store.g[L.w], self.store.g[L.b], in_act=in_act)

    def _dg(self, L: _Conv, dy, dx, aux=None, aux_act=ACT_NONE, res=None):
        ops.layer_dgrad(L.g, dy, self.store.p[L.w], dx, aux=aux, aux_act=aux_act, res=res,
                        wsplit=self.store.split_view(L.ws_d))

    # -- ConvResidualStack (vqvae.py:148-181) ---------------------------------------------------
    def _build_stack(self, store, prefix, h, w):
        hu, rhu = self._hidden_units, self._residual_hidden_units
        self._stack = []
        for i in range(self._residual_blocks):
            self._stack.append((_Conv(store, prefix, f"res3x3_{i}", LayerGeom.conv(h, w, hu, rhu, 3, 1, "SAME")),
                                _Conv(store, prefix, f"res1x1_{i}", LayerGeom.conv(h, w, rhu, hu, 1, 1, "SAME"))))

    def _stack_forward(self, h: torch.Tensor) -> torch.Tensor:
        """h: the stack's input as stored (relu is applied on load, which is idempotent for an
        already activated tensor).  Blocks keep their pre-activation sums; the LAST block fuses the
        stack's final relu into its store.  Returns relu(h_last)."""
        B = h.shape[0]
        self._sh, self._sc = [h], []
        last = len(self._stack) - 1
        for i, (c3, c1) in enumerate(self._stack):
            u = self.buf(f"res_c3_{i}", (B, c3.g.OH, c3.g.OW, c3.g.CO))
            self._fwd(c3, h, u, in_act=ACT_RELU)
            hn = self.buf(f"res_h_{i}", (B, c1.g.OH, c1.g.OW, c1.g.CO))
            self._fwd(c1, u, hn, in_act=ACT_RELU, res=h, out_act=ACT_RELU if i == last else ACT_NONE)
            self._sc.append(u)
            self._sh.append(hn)
            h = hn
        return h

    def _stack_backward(self, dh: torch.Tensor, input_is_activated: bool) -> torch.Tensor:
        """dh: gradient w.r.t. the last block's pre-activation sum.  Returns the gradient w.r.t. the
        stack input's pre-activation.  input_is_activated: the stored input is relu(pre) and is both
        the conv path's input and the skip path (encoder), so relu' multiplies the skip gradient too."""
        B = dh.shape[0]
        for i in reversed(range(len(self._stack))):
            c3, c1 = self._stack[i]
            h_in, u = self._sh[i], self._sc[i]
            self._wg(c1, u, dh, in_act=ACT_RELU)
            du = self.buf(f"res_du_{i}", u.shape)
            self._dg(c1, dh, du, aux=u, aux_act=ACT_RELU)
            self._wg(c3, h_in, du, in_act=ACT_RELU)
            dprev = self.buf(f"res_dh_{i}", h_in.shape)
            flag = AUX_AFTER_RES if (i == 0 and input_is_activated) else 0
            self._dg(c3, du, dprev, aux=h_in, aux_act=ACT_RELU | flag, res=dh)
            dh = dprev
        return dh


class ConvResidualEncoder(_ConvResidualNet):
    """reference vqvae.py:184-217: 4x4/2 (hidden/2) -> 4x4/2 -> 3x3 convs, relu each, then the stack."""

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        if len(in_shape) != 3:
            raise ValueError(f"ConvResidualEncoder expects [H, W, C] inputs, got {tuple(in_shape)}")
        self.attach(store, prefix)
        h, w, c = in_shape
        hu = self._hidden_units
        self.enc = []
        for name, co, k, s in (("enc_1", hu // 2, 4, 2), ("enc_2", hu, 4, 2), ("enc_3", hu, 3, 1)):
            g = LayerGeom.conv(h, w, c, co, k, s, "SAME")
            self.enc.append(_Conv(store, prefix, name, g))
            h, w, c = g.OH, g.OW, co
        self._build_stack(store, prefix, h, w)
        return (h, w, hu)

    def __call__(self, x: Feat, is_training: bool = False) -> Feat:
        assert x.in_act == ACT_NONE
        B = x.t.shape[0]
        self._x = x
        self._outs = []
        h = x.t
        for i, L in enumerate(self.enc):
            out = self.buf(f"enc_out_{i}", (B, L.g.OH, L.g.OW, L.g.CO))
            self._fwd(L, h, out, out_act=ACT_RELU)
            self._outs.append(out)
            h = out
        return Feat(self._stack_forward(h), ACT_NONE, ACT_RELU)

    def backward(self, dpre: torch.Tensor, need_input_grad: bool = False) -> Optional[torch.Tensor]:
        """dpre: gradient w.r.t. the pre-activation of the stack's final relu."""
        B = dpre.shape[0]
        dpre = self._stack_backward(dpre, input_is_activated=True)
        for i in reversed(range(len(self.enc))):
            L = self.enc[i]
            inp = self._outs[i - 1] if i > 0 else self._x.t
            self._wg(L, inp, dpre)
            if i > 0:
                dprev = self.buf(f"enc_dpre_{i - 1}", inp.shape)
                self._dg(L, dpre, dprev, aux=inp, aux_act=ACT_RELU)
                dpre = dprev
            elif need_input_grad:
                dx = self.buf("dx", inp.shape)
                self._dg(L, dpre, dx, aux=self._x.t, aux_act=self._x.grad_act)
                return dx
        return None


class ConvResidualDecoder(_ConvResidualNet):
    """reference vqvae.py:220-266: 3x3 conv -> stack -> ConvT 4x4/2 (hidden/2) + relu -> ConvT 4x4/2
    (output_channels) = loc of Normal(loc, exp(log_scale) + 1e-5)."""

    SCALE_EPS = 1e-5

    def __init__(self, hidden_units, residual_blocks, residual_hidden_units, output_channels, name=None):
        super().__init__(hidden_units, residual_blocks, residual_hidden_units, name)
        self._output_channels = output_channels

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        self.attach(store, prefix)
        h, w, c = in_shape
        hu = self._hidden_units
        self.dec_1 = _Conv(store, prefix, "dec_1", LayerGeom.conv(h, w, c, hu, 3, 1, "SAME"))
        self._build_stack(store, prefix, h, w)
        g2 = LayerGeom.conv_t(h, w, hu, hu // 2, 4, 2, "SAME")
        self.dec_2 = _Conv(store, prefix, "dec_2", g2)
        g3 = LayerGeom.conv_t(g2.OH, g2.OW, hu // 2, self._output_channels, 4, 2, "SAME")
        self.dec_3 = _Conv(store, prefix, "dec_3", g3)
        store.add(f"{prefix}/log_scale", ())
        return (g3.OH, g3.OW, self._output_channels)

    def __call__(self, z: Feat, is_training: bool = False) -> torch.Tensor:
        """-> loc [B,H,W,output_channels]; the scale is `self.P('log_scale')` (see log_prob_sum)."""
        assert z.in_act == ACT_NONE
        B = z.t.shape[0]
        self._z = z
        g1, g2, g3 = self.dec_1.g, self.dec_2.g, self.dec_3.g
        h0 = self.buf("dec_1_out", (B, g1.OH, g1.OW, g1.CO))
        self._fwd(self.dec_1, z.t, h0)
        hs = self._stack_forward(h0)
        self._h2 = self.buf("dec_2_out", (B, g2.OH, g2.OW, g2.CO))
        self._fwd(self.dec_2, hs, self._h2, out_act=ACT_RELU)
        self._loc = self.buf("loc", (B, g3.OH, g3.OW, g3.CO))
        self._fwd(self.dec_3, self._h2, self._loc)
        return self._loc

    def log_prob_sum(self, x: torch.Tensor) -> torch.Tensor:
        """einops.reduce(Normal(loc, scale).log_prob(x), 'b ... -> b', 'sum') (vqvae.py:83-85)."""
        self._xt = x
        ll = self.buf("ll", (x.shape[0],))
        ops.normal_ll_fwd(self._loc, x, self.P("log_scale"), ll, self.SCALE_EPS)
        return ll

    def backward(self, g_ll: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """g_ll [B]: d loss / d ll.  Returns d loss / d z (+ res, the commitment gradient)."""
        dloc = self.buf("dloc", self._loc.shape)
        ops.normal_ll_bwd(self._loc, self._xt, self.P("log_scale"), g_ll, dloc, self.G("log_scale"), self.SCALE_EPS)
        hs = self._sh[-1]
        self._wg(self.dec_3, self._h2, dloc)
        d2 = self.buf("dec_2_dpre", self._h2.shape)
        self._dg(self.dec_3, dloc, d2, aux=self._h2, aux_act=ACT_RELU)
        self._wg(self.dec_2, hs, d2)
        dhs = self.buf("stack_dpre", hs.shape)
        self._dg(self.dec_2, d2, dhs, aux=hs, aux_act=ACT_RELU)
        dh0 = self._stack_backward(dhs, input_is_activated=False)
        self._wg(self.dec_1, self._z.t, dh0)
        dz = self.buf("dz", self._z.t.shape)
        self._dg(self.dec_1, dh0, dz, res=res)
        return dz


class VQVAEPartialEncoder(Module):
    """reference vqvae.py:99-130: ConvResidualEncoder on [x*b | b] -> Flatten -> Linear(conditional_dim)."""

    def __init__(self, conditional_dim: int, vqvae_config: Dict[str, Any], name: Optional[str] = None):
        super().__init__(name)
        self._conditional_dim = conditional_dim
        self.encoder = ConvResidualEncoder(vqvae_config["hidden_units"], vqvae_config["residual_blocks"],
                                           vqvae_config["residual_hidden_units"])

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        self.attach(store, prefix)
        self.encoder.ws = self.ws
        h, w, c = self.encoder.build(store, f"{prefix}/encoder", in_shape)
        self._flat = h * w * c
        self.g_lin = LayerGeom.dense(self._flat, self._conditional_dim)
        store.add(f"{prefix}/linear/w", (self._flat, self._conditional_dim), fan_in=self._flat)
        store.add(f"{prefix}/linear/b", (self._conditional_dim,))
        self._ws = (store.request_split(f"{prefix}/linear/w", self.g_lin, "fwd"),
                    store.request_split(f"{prefix}/linear/w", self.g_lin, "dgrad"))
        return (self._conditional_dim,)

    def __call__(self, x_o_b: torch.Tensor, is_training: bool = False) -> torch.Tensor:
        B = x_o_b.shape[0]
        feat = self.encoder(Feat(x_o_b), is_training=is_training)
        self._feat = feat
        out = self.buf("cond", (B, self._conditional_dim))
        ops.layer_forward(self.g_lin, feat.t.view(B, self._flat), self.P("linear/w"), self.P("linear/b"), out,
                          in_act=feat.in_act, wsplit=self.store.split_view(self._ws[0]))
        return out

    def backward(self, dcond: torch.Tensor) -> None:
        B = dcond.shape[0]
        feat = self._feat
        flat = feat.t.view(B, self._flat)
        self.wgrad(self.g_lin, flat, dcond, self.G("linear/w"), self.G("linear/b"), in_act=feat.in_act)
        dfeat = self.buf("dfeat", (B, self._flat))
        ops.layer_dgrad(self.g_lin, dcond, self.P("linear/w"), dfeat, aux=flat, aux_act=feat.grad_act,
                        wsplit=self.store.split_view(self._ws[1]))
        self.encoder.backward(dfeat.view(feat.t.shape), need_input_grad=False)


class VectorQuantizerEMA(Module):
    """hk.nets.VectorQuantizerEMA (third party) as constructed at reference
    vqvae.py:66-72.  State tensors (haiku names): embeddings [D,K], ema_cluster_size/{hidden,average},
    ema_dw/{hidden,average}, counter."""

    def __init__(self, embedding_dim, num_embeddings, commitment_cost, decay, epsilon: float = VQ_EPSILON,
                 cross_replica_axis: Optional[str] = None, name: Optional[str] = None):
        super().__init__(name)
        self.embedding_dim, self.num_embeddings = embedding_dim, num_embeddings
        self.commitment_cost, self.decay, self.epsilon = commitment_cost, decay, epsilon
        self.cross_replica_axis = cross_replica_axis
        self.state: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def init_state(self, device, seed: int = 2) -> None:
        """embeddings ~ hk.initializers.VarianceScaling(distribution='uniform') = U(+-sqrt(3/D));
        both EMAs start at zero (hk.ExponentialMovingAverage.initialize only takes the shape)."""
        D, K = self.embedding_dim, self.num_embeddings
        lim = math.sqrt(3.0 / D)
        rng = np.random.default_rng(seed)
        emb = torch.from_numpy(rng.uniform(-lim, lim, size=(D, K)).astype(np.float32)).to(device)
        z = lambda *s: torch.zeros(s, device=device)   # noqa: E731
        self.state = OrderedDict([
            ("embeddings", emb),
            ("ema_cluster_size/hidden", z(K)), ("ema_cluster_size/average", z(K)),
            ("ema_dw/hidden", z(D, K)), ("ema_dw/average", z(D, K)),
            ("counter", torch.zeros(1, dtype=torch.int32, device=device)),
        ])
        self._geom = LayerGeom.dense(D, K)

    def __call__(self, z: torch.Tensor, is_training: bool, commit_grad_scale: float = 1.0) -> Dict[str, torch.Tensor]:
        """z [..., D] -> quantize / encoding_indices / per-row squared error / code counts.  When
        is_training the EMA update replaces `embeddings` AFTER the lookup (haiku order)."""
        D, K = self.embedding_dim, self.num_embeddings
        N = z.numel() // D
        st = self.state
        flat = z.view(N, D)
        dots = self.buf("dots", (N, K))
        d = self._geom._desc(N, "fwd")
        ops.gather_gemm(d, flat, st["embeddings"], None, None, None, dots)      # f32: argmin sees exact products
        idx = self.ws.get(f"{self.prefix}/idx", (N,), dtype=torch.int32)
        quant = self.buf("quantize", tuple(z.shape))
        sqerr, counts, e2 = self.buf("sqerr", (N,)), self.buf("counts", (K,)), self.buf("e2", (K,))
        cgrad = self.buf("commit_grad", tuple(z.shape)) if is_training else None
        dw = self.buf("dw", (D, K)) if is_training else None
        # d/dz of commitment_cost * mean((sg(q) - z)^2) = 2*commitment_cost/(N*D) * (z - q)
        coef = 2.0 * self.commitment_cost / (N * D) * commit_grad_scale
        ops.vq_select(flat, st["embeddings"], dots, e2, idx, quant, cgrad, sqerr, counts, dw, coef)
        if is_training:
            if self.cross_replica_axis is not None:
                from ..parallel import allreduce_sum_

                ops.host_call(allreduce_sum_, counts)           # jax.lax.psum over the pmap axis
                ops.host_call(allreduce_sum_, dw)
            ops.vq_ema_update(counts, dw, st["ema_cluster_size/hidden"], st["ema_cluster_size/average"],
                              st["ema_dw/hidden"], st["ema_dw/average"], st["embeddings"], st["counter"],
                              self.decay, self.epsilon)
        return {"quantize": quant, "encoding_indices": idx.view(tuple(z.shape[:-1])), "sqerr": sqerr, "counts": counts,
                "commit_grad": cgrad}

    def quantize(self, encoding_indices: torch.Tensor) -> torch.Tensor:
        """embedding lookup (reference vqvae.py:302)."""
        out = torch.empty(tuple(encoding_indices.shape) + (self.embedding_dim,), device=encoding_indices.device)
        ops.vq_lookup(encoding_indices.contiguous().view(-1).to(torch.int32), self.state["embeddings"], out)
        return out


class _LazyDict(dict):
    """A dict whose listed keys are computed on first access.  The training step never touches
    them, so no torch kernel (and no allocation) happens inside a captured launch sequence."""

    def __init__(self, lazy, **kw):
        super().__init__(**kw)
        self._lazy = dict(lazy)

    def __missing__(self, key):
        if key in self._lazy:
            v = self._lazy[key]()
            self[key] = v
            return v
        raise KeyError(key)

    def keys(self):
        return list(super().keys()) + [k for k in self._lazy if not super().__contains__(k)]

    def __contains__(self, key):
        return super().__contains__(key) or key in self._lazy


class VQVAE(Module):
    """The Vector-Quantized VAE (reference vqvae.py:13-96), EMA codebook only."""

    def __init__(self, output_channels: int = 3, embedding_dim: int = 64, num_embeddings: int = 512,
                 hidden_units: int = 128, residual_blocks: int = 2, residual_hidden_units: int = 128,
                 decay: float = 0.99, commitment_cost: float = 0.25, cross_replica_axis: Optional[str] = None,
                 use_ema: bool = True, name: Optional[str] = None, device: Optional[str] = None, seed: int = 1):
        super().__init__(name)
        if not use_ema:
            raise NotImplementedError("use_ema=False (hk.nets.VectorQuantizer) has no HIP path; every reference "
                                      "config sets use_ema=True")
        self.config = dict(output_channels=output_channels, embedding_dim=embedding_dim, num_embeddings=num_embeddings,
                           hidden_units=hidden_units, residual_blocks=residual_blocks,
                           residual_hidden_units=residual_hidden_units, decay=decay, commitment_cost=commitment_cost,
                           use_ema=use_ema)
        self.encoder = ConvResidualEncoder(hidden_units, residual_blocks, residual_hidden_units)
        self.decoder = ConvResidualDecoder(hidden_units, residual_blocks, residual_hidden_units, output_channels)
        self.vq = VectorQuantizerEMA(embedding_dim, num_embeddings, commitment_cost, decay,
                                     cross_replica_axis=cross_replica_axis)
        self._device, self._seed = device, seed
        self.store: Optional[ParamStore] = None

    def init(self, x_shape, device=None, seed: Optional[int] = None) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("posterior_matching_amd needs an MI355X: there is no CPU fallback path")
        from .. import _lib

        _lib.load()
        device = torch.device(device or self._device or "cuda:0")
        store, ws = ParamStore(), Workspace(device)
        for m in (self.encoder, self.decoder, self.vq):
            m.ws = ws
        x_shape = tuple(int(s) for s in x_shape)
        h, w, c = self.encoder.build(store, "encoder", x_shape)
        D = self.config["embedding_dim"]
        self.pre_vq = _Conv(store, "", "pre_vq_conv", LayerGeom.conv(h, w, c, D, 1, 1, "SAME"))
        self.decoder.build(store, "decoder", (h, w, D))
        seed = self._seed if seed is None else seed
        store.allocate(device, seed)
        self.vq.attach(store, "vq")
        self.vq.init_state(device, seed + 1)
        self.store, self.ws, self._x_shape = store, ws, x_shape
        self.attach(store, "")
        self.metrics = torch.zeros(8, device=device)

    @property
    def state(self) -> Dict[str, torch.Tensor]:
        return self.vq.state

    @property
    def num_params(self) -> int:
        return self.store.num_params

    def buf(self, name, shape):
        return self.ws.get(f"vqvae/{name}", shape)

    def __call__(self, inputs: torch.Tensor, is_training: bool = False, grad_scale: Optional[float] = None):
        """reference vqvae.py:78-96.  Returns the reference's dict (device tensors owned by the model,
        overwritten by the next call); `decoder_dist` is replaced by its two parameters
        `reconstruction` (= mean) and `scale`.  grad_scale: d(total loss)/d(this loss), default 1."""
        if self.store is None:
            self.init(inputs.shape[1:], inputs.device)
        B = inputs.shape[0]
        self._B = B
        feat = self.encoder(Feat(inputs), is_training=is_training)
        g = self.pre_vq.g
        z = self.buf("z", (B, g.OH, g.OW, g.CO))
        ops.layer_forward(g, feat.t, self.store.p[self.pre_vq.w], self.store.p[self.pre_vq.b], z, in_act=feat.in_act,
                          wsplit=self.store.split_view(self.pre_vq.ws_f))
        self._feat = feat
        gs = 1.0 if grad_scale is None else grad_scale
        vq = self.vq(z, is_training, commit_grad_scale=gs)
        loc = self.decoder(Feat(vq["quantize"]), is_training=is_training)
        ll = self.decoder.log_prob_sum(inputs)
        g_ll = self.buf("g_ll", (B,)) if is_training else None
        ops.vqvae_loss(ll, vq["sqerr"], vq["counts"], self.config["embedding_dim"], self.config["commitment_cost"],
                       gs / B, self.metrics, g_ll)
        self._vq, self._g_ll = vq, g_ll
        m = self.metrics
        K, idx = self.config["num_embeddings"], vq["encoding_indices"]
        # `encodings` (one-hot [N,K]) and `scale` are derived views for callers; the step itself only
        # needs the code counts and log_scale
        vq_out = _LazyDict(
            {"encodings": lambda: torch.nn.functional.one_hot(idx.reshape(-1).long(), K).to(torch.float32)},
            quantize=vq["quantize"], loss=m[2], perplexity=m[3], encoding_indices=idx)
        log_scale = self.store.p["decoder/log_scale"]
        return _LazyDict({"scale": lambda: torch.exp(log_scale) + ConvResidualDecoder.SCALE_EPS},
                         loss=m[0], vq_output=vq_out, z=z, reconstruction=loc, reconstruction_loss=m[1], ll=ll)

    def encoding_indices(self, inputs: torch.Tensor) -> torch.Tensor:
        """`self(inputs, is_training=False)["vq_output"]["encoding_indices"]` without the decoder and the losses (what XLA's
        dead-code elimination leaves of the frozen model in train_pm_vqvae.py:81-86): encoder -> pre_vq -> nearest code."""
        if self.store is None:
            self.init(inputs.shape[1:], inputs.device)
        B = inputs.shape[0]
        feat = self.encoder(Feat(inputs), is_training=False)
        g = self.pre_vq.g
        z = self.buf("z", (B, g.OH, g.OW, g.CO))
        ops.layer_forward(g, feat.t, self.store.p[self.pre_vq.w], self.store.p[self.pre_vq.b], z, in_act=feat.in_act,
                          wsplit=self.store.split_view(self.pre_vq.ws_f))
        return self.vq(z, False)["encoding_indices"]

    def backward(self) -> None:
        """Accumulates d loss / d params of the last is_training=True call into the flat gradient
        buffer (zero it first).  Straight-through estimator: d/dz = d/dquantize + commitment term."""
        dz = self.decoder.backward(self._g_ll, res=self._vq["commit_grad"])
        feat = self._feat
        self.encoder.wgrad(self.pre_vq.g, feat.t, dz, self.store.g[self.pre_vq.w], self.store.g[self.pre_vq.b],
                           in_act=feat.in_act)
        dfeat = self.buf("dfeat", tuple(feat.t.shape))
        ops.layer_dgrad(self.pre_vq.g, dz, self.store.p[self.pre_vq.w], dfeat, aux=feat.t, aux_act=feat.grad_act,
                        wsplit=self.store.split_view(self.pre_vq.ws_d))
        self.encoder.backward(dfeat, need_input_grad=False)
        self.ws.join_aux()

    def zero_grad(self) -> None:
        self.store.zero_grad()

    def params_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.to_dict("p")

    def grads_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.to_dict("g")

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return OrderedDict((k, v.detach().clone()) for k, v in self.vq.state.items())

    def load_params(self, values) -> None:
        self.store.load_dict(values)

    def load_state(self, values) -> None:
        for k, v in values.items():
            t = torch.as_tensor(np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v))
            self.vq.state[k].copy_(t.reshape(self.vq.state[k].shape).to(self.vq.state[k].dtype))


def build_partial_posterior(vqvae: VQVAE, conditional_dim: int, pixel_cnn_config: Dict[str, Any], x_shape, seed: int = 0):
    """Builds the two trainable modules of stage 2 (train_pm_vqvae.py:82-84) on their own parameter
    store: VQVAEPartialEncoder on [x*b | b] and the conditional PixelCNN over the code grid."""
    from .pixel_cnn import PixelCNN

    if vqvae.store is None:
        vqvae.init(x_shape)
    dev = vqvae.store.device
    store, ws = ParamStore(), Workspace(dev)
    penc = VQVAEPartialEncoder(conditional_dim, vqvae.config)
    pc_cfg = dict(pixel_cnn_config)
    pc_cfg["num_indices"] = vqvae.config["num_embeddings"]            # train_pm_vqvae.py:78
    pcnn = PixelCNN(**pc_cfg)
    penc.ws = pcnn.ws = ws
    xb_shape = tuple(x_shape[:-1]) + (x_shape[-1] + 1,)             # [x*b | b]: image masks have ONE channel (masking.py:346)
    (cond_dim,) = penc.build(store, "partial_encoder", xb_shape)
    pcnn.build(store, "pixel_cnn", cond_dim)
    store.allocate(dev, seed)
    return penc, pcnn, store


def vqvae_impute(vqvae: VQVAE, partial_encoder: VQVAEPartialEncoder, partial_posterior, x: torch.Tensor,
                 b: torch.Tensor, num_samples: int = 5, seed: int = 0, gumbel: Optional[torch.Tensor] = None):
    """reference vqvae.py:269-312: imputations [B, num_samples, H, W, C] - codes sampled from the partially
    observed posterior, decoded by the VQ-VAE, observed pixels kept, clipped to [0, 1]."""
    B = x.shape[0]
    ws = partial_encoder.ws
    xob = ws.get("impute/x_o_b", tuple(x.shape[:-1]) + (x.shape[-1] + b.shape[-1],))
    ops.mask_concat(x, b, xob)
    cond = partial_encoder(xob)
    samples = partial_posterior.sample(seed=seed, sample_shape=num_samples, conditional_input=cond, gumbel=gumbel)
    idx = samples.permute(1, 0, 2, 3).contiguous()                            # [B, S, H, W]
    q = vqvae.vq.quantize(idx.view((B * num_samples,) + tuple(idx.shape[2:])))
    loc = vqvae.decoder(Feat(q))
    imp = loc.view((B, num_samples) + tuple(loc.shape[1:])).clone()
    ops.impute_blend(x, b, imp)
    return imp


def imputation_psnr(imputations: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """eval_pm_vqvae.py:133-136: PSNR [B] of the mean imputation against the full image."""
    psnr = torch.empty(x.shape[0], device=x.device)
    ops.imputation_psnr(imputations, x, psnr)
    return psnr
