"""Host-side mirror of posterior_matching/models/vdvae.py of the reference: PosteriorMatchingVDVAE.

Same constructor arguments and `__call__(x, b)` outputs (reference vdvae.py:41-94); the arithmetic
runs in libpmhip.so: every 1x1 / 3x3 convolution in the gather-GEMM engine (gelu' of a Block's
hidden activations fused into the data-gradient epilogue), the rest in the row-wise kernels of
csrc/pm_vdvae.hip.  There is no autodiff: `backward()` walks the buffers the forward left in HBM,
in reverse block order, accumulating the gradients of tensors with several consumers (the shared
encoder activations of a resolution, the per-resolution decoder state, the mix-in sources).

Not implemented here (no reference config needs them): custom_width_string (per-resolution widths
with channel padding), multi-channel logistic mixtures (coefficients), is_log_probs / impute / sample.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from ..ops import ACT_GELU, ACT_NONE, LayerGeom
from .core import Module, ParamStore, Workspace


def parse_layer_string(s: str) -> List[Tuple[int, Optional[int]]]:
    """reference vdvae.py:213-229"""
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


class _Conv:
    def __init__(self, store: ParamStore, name: str, geom: LayerGeom, fan_in: int):
        self.g, self.w, self.b = geom, f"{name}/w", f"{name}/b"
        store.add(self.w, geom.weight_shape, fan_in=fan_in)
        store.add(self.b, (geom.CO,))
        self.ws_f = store.request_split(self.w, geom, "fwd")
        self.ws_d = store.request_split(self.w, geom, "dgrad")
        self.ws_fd = self.ws_dd = None

    def request_dense_k(self, store: ParamStore) -> None:
        """second pair of split copies with K = (tap, c) unpadded per tap - what the fused Block reads for its 3x3 layers"""
        self.ws_fd = store.request_split(self.w, self.g, "fwd", dense_k=True)
        self.ws_dd = store.request_split(self.w, self.g, "dgrad", dense_k=True)


class Block(Module):
    """reference vdvae.py:263-299 without the pooling (the caller pools): four convolutions with gelu
    in front of each.  `forward` takes the ALREADY activated input gelu(x) (the callers build it, possibly
    from two concatenated sources) and an optional residual tensor added to the last conv's output."""

    def __init__(self, store: ParamStore, ws: Workspace, name: str, H: int, W: int, cin: int, mid: int, cout: int,
                 use_3x3: bool, zero_last: bool = False, out_init_div: int = 1):
        super().__init__(name)
        self.attach(store, name)
        self.ws = ws
        k = 3 if use_3x3 else 1
        pad = "SAME"
        self.c1 = _Conv(store, f"{name}/c1", LayerGeom.conv(H, W, cin, mid, 1, 1, pad), cin)
        self.c2 = _Conv(store, f"{name}/c2", LayerGeom.conv(H, W, mid, mid, k, 1, pad), k * k * mid)
        self.c3 = _Conv(store, f"{name}/c3", LayerGeom.conv(H, W, mid, mid, k, 1, pad), k * k * mid)
        # get_1x1(zero_last / init_multiple = sqrt(1/N)) (:193-205): stddev * sqrt(1/N) <=> fan_in * N
        self.c4 = _Conv(store, f"{name}/c4", LayerGeom.conv(H, W, mid, cout, 1, 1, pad), 0 if zero_last else mid * out_init_div)
        self.H, self.W, self.cin, self.mid, self.cout = H, W, cin, mid, cout
        # mid = 48: a tap's second 32-chunk is half empty in the per-tap layout (18 k-steps per 3x3 layer); K over (tap, c) is
        # 13.5 chunks = 14 k-steps.  PM_VB_NO_DENSE=1: the per-tap copies (A/B)
        self._dense = k == 3 and mid % 32 != 0 and not os.environ.get("PM_VB_NO_DENSE")
        if self._dense:
            self.c2.request_dense_k(store)
            self.c3.request_dense_k(store)

    def _f(self, L: _Conv, x, out, res=None, out2=None):
        """out2: gelu(out) from the same launch (pm_gather_gemm_bf16_dual) - the next convolution's input"""
        ops.layer_forward(L.g, x, self.store.p[L.w], self.store.p[L.b], out, res=res, wsplit=self.store.split_view(L.ws_f),
                          out2=out2, act2=ops.ACT_GELU if out2 is not None else ops.ACT_NONE)

    def _fused(self):
        """split views (forward, data-gradient) of c1..c4 when the one-launch form applies: default arithmetic (bf16x3),
        shapes inside pm_vdvae_block_fwd's preconditions; PM_NO_VDVAE_FUSED=1 restores the layer-by-layer path"""
        import os

        if os.environ.get("PM_NO_VDVAE_FUSED") or not self.store.use_bf16:
            return None
        layers = (self.c1, self.c2, self.c3, self.c4)
        dn = lambda L: self._dense and L in (self.c2, self.c3)   # noqa: E731
        f = [self.store.split_view(L.ws_fd if dn(L) else L.ws_f) for L in layers]
        d = [self.store.split_view(L.ws_dd if dn(L) else L.ws_d) for L in layers]
        if any(v is None for v in f + d):
            return None
        if not ops.vdvae_block_fused_ok(1, self.H, self.W, self.cin, self.cout, self.mid, self.c2.g.k):
            return None
        return f, d

    # -- the one-launch form: operands of this Block for ops.vdvae_blocks_fwd / _bwd (several independent Blocks of one
    #    geometry - the three Blocks of a decoder block, the same Block of the two encoders - share a launch) ---------------
    def fwd_io(self, x: torch.Tensor, x2: Optional[torch.Tensor], res: Optional[torch.Tensor], raw: bool):
        B, H, W = x.shape[0], self.H, self.W
        sh = lambda c: (B, H, W, c)   # noqa: E731
        fused = self._fused()
        if raw:
            self._xg = self.buf("xg", sh(self.cin))
        else:
            self._xg = x
        self._h = [self.buf(f"h{i + 1}", sh(self.mid)) for i in range(3)]
        self._g = [self.buf(f"g{i + 1}", sh(self.mid)) for i in range(3)]
        out = self.buf("out", sh(self.cout))
        biases = [self.store.p[L.b] for L in (self.c1, self.c2, self.c3, self.c4)]
        return ops.vdvae_block_io(x, self._h, self._g, out, fused[0], biases, x2=x2, res=res,
                                  xg_out=self._xg if raw else None, dense_k3=self._dense), out

    def bwd_io(self, dout: torch.Tensor, dx: torch.Tensor, x_pre: Optional[torch.Tensor], res: Optional[torch.Tensor]):
        B = dout.shape[0]
        self._dhs = [self.buf(f"dh{i + 1}", (B, self.H, self.W, self.mid)) for i in range(3)]
        self._dout = dout
        return ops.vdvae_block_io(dout, self._h, self._dhs, dx, self._fused()[1], None, res=res if x_pre is not None else None,
                                  xpre=x_pre, backward=True, dense_k3=self._dense)

    def weight_grads(self) -> None:
        """the four weight / bias gradients after bwd_io's launch: deferred to the end of the backward pass (one launch per
        geometry) when the model collects them, else launched now"""
        dhs, dout = self._dhs, self._dout
        pairs = ((self.c4, self._g[2], dout), (self.c3, self._g[1], dhs[2]), (self.c2, self._g[0], dhs[1]),
                 (self.c1, self._xg, dhs[0]))
        if self.ws.wgrad_batch is not None:
            for L, xin, dy in pairs:
                self.ws.wgrad_batch.add(L.g, xin, dy, self.store.g[L.w], self.store.g[L.b], self.store.use_bf16)
            return
        cur, aux = torch.cuda.current_stream(self.ws.device), self.ws.aux_stream()
        if aux is not cur and aux != cur:
            ops.wait_stream(aux, cur)               # ONE dependency per Block: its four weight gradients share a stream
        with torch.cuda.stream(aux):
            for L, xin, dy in pairs:
                ops.layer_wgrad(L.g, xin, dy, self.store.g[L.w], self.store.g[L.b], bf16=self.store.use_bf16)

    def forward_raw(self, x: torch.Tensor, x2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """the Block applied to the RAW input [x | x2] (reference :283: `x_hat = self.c1(jax.nn.gelu(x))`): in the one-launch form
        the gelu runs inside the kernel as the rows are loaded and gelu([x | x2]) is written out for the weight gradient;
        otherwise a gelu launch builds it first"""
        B, H, W = x.shape[0], self.H, self.W
        if self._fused() is None or (x2 is not None and x.shape[-1] % 32 != 0):
            xg = self.buf("xg", (B, H, W, self.cin))
            ops.gelu_fwd(x, x2, xg)
            return self.forward(xg, res=res)
        io, out = self.fwd_io(x, x2, res, raw=True)
        ops.vdvae_blocks_fwd([io], B, H, W, self.mid, self.c2.g.k)
        return out

    def forward(self, xg: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, H, W = xg.shape[0], self.H, self.W
        self._xg = xg
        sh = lambda c: (B, H, W, c)   # noqa: E731
        if self._fused() is not None:    # the four convolutions and the three gelus between them in ONE launch
            io, out = self.fwd_io(xg, None, res, raw=False)
            ops.vdvae_blocks_fwd([io], B, H, W, self.mid, self.c2.g.k)
            return out
        self._h, self._g = [], []
        x = xg
        for i, L in enumerate((self.c1, self.c2, self.c3)):
            h = self.buf(f"h{i + 1}", sh(self.mid))
            g = self.buf(f"g{i + 1}", sh(self.mid))
            self._f(L, x, h, out2=g)          # pre-activation (for gelu' in the backward pass) and gelu(h) in one launch
            self._h.append(h)
            self._g.append(g)
            x = g
        out = self.buf("out", sh(self.cout))
        self._f(self.c4, x, out, res=res)
        return out

    def backward(self, dout: torch.Tensor, dx: torch.Tensor, x_pre: Optional[torch.Tensor] = None,
                 res: Optional[torch.Tensor] = None) -> None:
        """dout: gradient w.r.t. the last conv's output.  Writes into dx the gradient w.r.t. the activated
        input gelu(x) - or, when x_pre (the single-source pre-activation x) is given, w.r.t. x itself:
        dx = dgrad * gelu'(x_pre) + res (one fused epilogue)."""
        B = dout.shape[0]
        sh = lambda c: (B, self.H, self.W, c)   # noqa: E731
        d = dout
        layers = (self.c1, self.c2, self.c3, self.c4)
        if self._fused() is not None:
            # the four data gradients (with the gelu' factors between them) in ONE launch; the four weight gradients follow:
            # they read the stored g1..g3 / dh1..dh3 and only feed the optimizer
            ops.vdvae_blocks_bwd([self.bwd_io(dout, dx, x_pre, res)], B, self.H, self.W, self.mid, self.c2.g.k)
            self.weight_grads()
            return
        for i in (3, 2, 1):
            L = layers[i]
            self.wgrad(L.g, self._g[i - 1], d, self.store.g[L.w], self.store.g[L.b])
            dh = self.buf(f"dh{i}", sh(self.mid))
            ops.layer_dgrad(L.g, d, self.store.p[L.w], dh, aux=self._h[i - 1], aux_act=ACT_GELU,
                            wsplit=self.store.split_view(L.ws_d))
            d = dh
        self.wgrad(self.c1.g, self._xg, d, self.store.g[self.c1.w], self.store.g[self.c1.b])
        if x_pre is not None:
            ops.layer_dgrad(self.c1.g, d, self.store.p[self.c1.w], dx, aux=x_pre, aux_act=ACT_GELU, res=res,
                            wsplit=self.store.split_view(self.c1.ws_d))
        else:
            ops.layer_dgrad(self.c1.g, d, self.store.p[self.c1.w], dx, wsplit=self.store.split_view(self.c1.ws_d))


class Encoder(Module):
    """reference vdvae.py:302-348: 3x3 stem + residual bottleneck blocks, AvgPool at the `d` blocks;
    returns the last activation of every resolution."""

    def __init__(self, width: int, blocks: str, bottleneck_multiple: float, custom_width_string: Optional[str] = None,
                 name: Optional[str] = None):
        super().__init__(name)
        if custom_width_string:
            raise NotImplementedError("custom_width_string has no HIP path (the reference configs leave it None)")
        self.width, self.blocks_spec, self.mid = width, parse_layer_string(blocks), int(width * bottleneck_multiple)

    def build(self, store: ParamStore, prefix: str, in_shape) -> None:
        self.attach(store, prefix)
        H, W, C = in_shape
        assert H == W
        self.stem = _Conv(store, f"{prefix}/stem", LayerGeom.conv(H, W, C, self.width, 3, 1, "SAME"), 9 * C)
        self.blocks: List[Tuple[Block, Optional[int], int]] = []
        n = len(self.blocks_spec)
        res = H
        for i, (r, down) in enumerate(self.blocks_spec):
            assert r == res, f"encoder block {i} declared at resolution {r} but the running resolution is {res}"
            blk = Block(store, self.ws, f"{prefix}/block_{i}", res, res, self.width, self.mid, self.width, res > 2,
                        out_init_div=n)
            self.blocks.append((blk, down, res))
            if down is not None:
                res = res // down

    # The pass is written as steps (stem, per-Block operands, per-Block bookkeeping) so that the two encoders of the model can
    # walk their Blocks in lockstep and share ONE launch per Block index (run_encoders / run_encoders_backward below).
    def _begin(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        self._x = x
        g = self.stem.g
        h = self.buf("stem_out", (B, g.OH, g.OW, g.CO))
        ops.layer_forward(g, x, self.store.p[self.stem.w], self.store.p[self.stem.b], h,
                          wsplit=self.store.split_view(self.stem.ws_f))
        self._acts = {h.shape[1]: h}
        self._ins, self._outs = [], []
        return h

    def _after_block(self, i: int, h: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        blk, down, res = self.blocks[i]
        self._ins.append(h)
        if down is not None:
            pooled = self.buf(f"block_{i}/pooled", (h.shape[0], res // down, res // down, self.width))
            ops.avgpool_fwd(out, pooled, down)
            out = pooled
        self._outs.append(out)
        self._acts[out.shape[1]] = out
        return out

    def __call__(self, x: torch.Tensor) -> Dict[int, torch.Tensor]:
        h = self._begin(x)
        for i, (blk, down, res) in enumerate(self.blocks):
            h = self._after_block(i, h, blk.forward_raw(h, None, res=h))
        return self._acts

    def _bwd_begin(self, dacts: Dict[int, torch.Tensor]) -> None:
        self._exported = {t.data_ptr(): r for r, t in self._acts.items()}
        self._dacts, self._dh = dacts, None

    def _bwd_block_args(self, i: int):
        """-> (dout, dx, x_pre) of Block i: the gradient arriving at its output (exported-activation gradients and the
        pooling undone), where its input gradient goes, its raw input"""
        blk, down, res = self.blocks[i]
        out = self._outs[i]
        r_out = self._exported.get(out.data_ptr())
        dh = self._dh
        if r_out is not None:
            if dh is None:
                dh = self._dacts[r_out]
            else:
                ops.axpy1(self._dacts[r_out], dh)
        if down is not None:
            dfull = self.buf(f"block_{i}/dfull", tuple(blk._xg.shape[:3]) + (self.width,))
            ops.avgpool_bwd(dh, dfull, down)
            dh = dfull
        dprev = self.buf(f"block_{i}/dx", tuple(self._ins[i].shape))
        self._dh = dprev
        return dh, dprev, self._ins[i]

    def _bwd_end(self) -> None:
        dh = self._dh
        r0 = self._exported.get(self._ins[0].data_ptr())
        if r0 is not None:
            ops.axpy1(self._dacts[r0], dh)
        self.wgrad(self.stem.g, self._x, dh, self.store.g[self.stem.w], self.store.g[self.stem.b])

    def backward(self, dacts: Dict[int, torch.Tensor]) -> None:
        """dacts[res]: gradient w.r.t. the exported activation of each resolution (from the decoder)."""
        self._bwd_begin(dacts)
        for i in reversed(range(len(self.blocks))):
            dout, dx, x_pre = self._bwd_block_args(i)
            self.blocks[i][0].backward(dout, dx, x_pre=x_pre, res=dout)
        self._bwd_end()


def _pairable(e: "Encoder", m: "Encoder") -> bool:
    return (len(e.blocks) == len(m.blocks) and all(be._fused() is not None and bm._fused() is not None and be.H == bm.H
                                                   for (be, _, _), (bm, _, _) in zip(e.blocks, m.blocks)))


def run_encoders(e: "Encoder", m: "Encoder", x: torch.Tensor, xm: torch.Tensor):
    """both encoders of the model (reference vdvae.py:77-80) walked in lockstep: Block i of the two is ONE launch"""
    he, hm = e._begin(x), m._begin(xm)
    B = x.shape[0]
    for i in range(len(e.blocks)):
        be, bm = e.blocks[i][0], m.blocks[i][0]
        io_e, oe = be.fwd_io(he, None, he, raw=True)
        io_m, om = bm.fwd_io(hm, None, hm, raw=True)
        ops.vdvae_blocks_fwd([io_e, io_m], B, be.H, be.W, be.mid, be.c2.g.k)
        he, hm = e._after_block(i, he, oe), m._after_block(i, hm, om)
    return e._acts, m._acts


def run_encoders_backward(e: "Encoder", m: "Encoder", dacts, dmacts) -> None:
    e._bwd_begin(dacts)
    m._bwd_begin(dmacts)
    for i in reversed(range(len(e.blocks))):
        be, bm = e.blocks[i][0], m.blocks[i][0]
        de, dxe, xe = e._bwd_block_args(i)
        dm, dxm, xm = m._bwd_block_args(i)
        ops.vdvae_blocks_bwd([be.bwd_io(de, dxe, xe, de), bm.bwd_io(dm, dxm, xm, dm)], de.shape[0], be.H, be.W, be.mid,
                             be.c2.g.k)
        be.weight_grads()
        bm.weight_grads()
    e._bwd_end()
    m._bwd_end()


class PosteriorMatchingDecoderBlock(Module):
    """reference vdvae.py:479-687 (forward_posterior path)."""

    def __init__(self, store: ParamStore, ws: Workspace, name: str, latent_dim: int, res: int, mixin: Optional[int],
                 num_blocks: int, width: int, mid: int):
        super().__init__(name)
        self.attach(store, name)
        self.ws = ws
        self.base, self.mixin, self.Z, self.width = res, mixin, latent_dim, width
        Z, u3 = latent_dim, res > 2
        self.posterior = Block(store, ws, f"{name}/posterior", res, res, 2 * width, mid, 2 * Z, u3)
        self.masked_posterior = Block(store, ws, f"{name}/masked_posterior", res, res, 2 * width, mid,
                                      Z + Z * (Z + 1) // 2, u3)
        self.prior = Block(store, ws, f"{name}/prior", res, res, width, mid, 2 * Z + width, u3, zero_last=True)
        self.z_proj = _Conv(store, f"{name}/z_proj", LayerGeom.conv(res, res, Z, width, 1, 1, "SAME"), Z * num_blocks)
        self.resnet = Block(store, ws, f"{name}/resnet", res, res, width, mid, width, u3, out_init_div=num_blocks)

    def forward(self, x_in: torch.Tensor, acts: torch.Tensor, macts: torch.Tensor, eps: torch.Tensor,
                kl: torch.Tensor, pm_kl: torch.Tensor, streams=None) -> torch.Tensor:
        """streams = (s1, s2): the posterior, masked-posterior and prior Blocks only share x_in, so the latter two
        run on companion streams (every kernel here is a few workgroups: latency, not throughput, bounds the step)."""
        B, r, W, Z = x_in.shape[0], self.base, self.width, self.Z
        P = r * r
        sh = lambda c: (B, r, r, c)   # noqa: E731
        self._x_in, self._acts, self._macts, self._eps = x_in, acts, macts, eps
        main = torch.cuda.current_stream(x_in.device)
        self._grouped = all(b._fused() is not None for b in (self.posterior, self.masked_posterior, self.prior)) and W % 32 == 0
        if self._grouped:
            # the three Blocks only share x_in: ONE launch (blockIdx.y picks the Block), no companion streams.  The
            # masked-posterior Block only feeds pm_kl (and, backward, the masked encoder): with a side stream it is a chain
            # of its own and the main chain's launch holds two Blocks (28x28 at per-GPU 16: 224 workgroups, one round on 256
            # CUs instead of 336 in two)
            io_p, self._pp = self.posterior.fwd_io(x_in, acts, None, raw=True)
            io_m, self._mp = self.masked_posterior.fwd_io(x_in, macts, None, raw=True)   # stop_gradient(x): see backward
            io_r, self._pr = self.prior.fwd_io(x_in, None, None, raw=True)
            self._mp_split = self.kl_stream is not None and not os.environ.get("PM_VDVAE_GROUP3")
            if self._mp_split:
                ops.wait_stream(self.kl_stream, main)            # x_in is the previous block's output
                with torch.cuda.stream(self.kl_stream):
                    ops.vdvae_blocks_fwd([io_m], B, r, r, self.posterior.mid, self.posterior.c2.g.k)
                ops.vdvae_blocks_fwd([io_p, io_r], B, r, r, self.posterior.mid, self.posterior.c2.g.k)
            else:
                ops.vdvae_blocks_fwd([io_p, io_m, io_r], B, r, r, self.posterior.mid, self.posterior.c2.g.k)
            streams = None
        s1, s2 = streams if streams is not None else (main, main)
        if streams is not None:
            ops.wait_stream(s1, main)
            ops.wait_stream(s2, main)
        if not self._grouped:
            with torch.cuda.stream(s1):
                self._mp = self.masked_posterior.forward_raw(x_in, macts)   # stop_gradient(x): handled in backward (:536-538)
            with torch.cuda.stream(s2):
                self._pr = self.prior.forward_raw(x_in)
            self._pp = self.posterior.forward_raw(x_in, acts)
        if streams is not None:
            ops.wait_stream(main, s1)
            ops.wait_stream(main, s2)
        self._z = self.buf("z", sh(Z))
        # one launch for the three steps.  Its first form (four rows per wave, one after the other) was no faster than the three
        # launches (891 vs 897 img/s at per-GPU 8); with one row per wave and 16 waves per workgroup it is: 8.49 -> 8.24 ms
        # (PM_VDVAE_NO_SAMPLE_PROJECT=1: the three launches, A/B)
        fuse_sp = not os.environ.get("PM_VDVAE_NO_SAMPLE_PROJECT")
        x2 = self.buf("x2", sh(W))
        if fuse_sp:     # x += h (:558), sample + KL (:559-561), x += z_proj(z) (:562): one launch
            ops.sample_project_fwd(self._pp, self._pr, eps, x_in, self.store.p[self.z_proj.w].view(Z, W),
                                   self.store.p[self.z_proj.b], self._z, x2, kl, P)
        else:
            x1 = self.buf("x1", sh(W))
            ops.add_cols(x_in, self._pr, 2 * Z, x1)              # x += h (:558)
            ops.diag_sample_kl_fwd(self._pp, self._pr, eps, self._z, kl, P)
        ks = self.kl_stream          # pm_kl only feeds the loss: off the chain, beside z_proj and the resnet Block
        if ks is not None:
            ops.wait_stream(ks, main)
            with torch.cuda.stream(ks):
                ops.diag_tril_kl_fwd(self._pp, self._mp, pm_kl, Z, P)
        else:
            ops.diag_tril_kl_fwd(self._pp, self._mp, pm_kl, Z, P)
        if not fuse_sp:
            ops.layer_forward(self.z_proj.g, self._z, self.store.p[self.z_proj.w], self.store.p[self.z_proj.b], x2, res=x1,
                              wsplit=self.store.split_view(self.z_proj.ws_f))
        self._x2 = x2
        return self.resnet.forward_raw(x2, None, res=x2)

    def forward_partial(self, x_in: torch.Tensor, macts: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        """forward_partial_posterior / sample_partial_posterior (reference :689-703, :573-590): z is drawn from the
        masked (TriL) posterior, z = loc + L eps; no KL terms."""
        B, r, W, Z = x_in.shape[0], self.base, self.width, self.Z
        sh = lambda c: (B, r, r, c)   # noqa: E731
        am = self.buf("am", sh(2 * W))
        ops.gelu_fwd(x_in, macts, am)
        mp = self.masked_posterior.forward(am)
        ap = self.buf("ap", sh(W))
        ops.gelu_fwd(x_in, None, ap)
        pr = self.prior.forward(ap)
        x1 = self.buf("x1", sh(W))
        ops.add_cols(x_in, pr, 2 * Z, x1)
        z, scratch = self.buf("z", sh(Z)), self.buf("tril_kl_scratch", (B * r * r,))
        ops.tril_sample_kl_fwd(mp.view(B * r * r, -1), eps.view(B * r * r, Z), z.view(B * r * r, Z), scratch)
        x2 = self.buf("x2", sh(W))
        ops.layer_forward(self.z_proj.g, z, self.store.p[self.z_proj.w], self.store.p[self.z_proj.b], x2, res=x1,
                          wsplit=self.store.split_view(self.z_proj.ws_f))
        x2g = self.buf("x2g", sh(W))
        ops.gelu_fwd(x2, None, x2g)
        return self.resnet.forward(x2g, res=x2)

    def forward_prior(self, x_in: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        """forward_prior / sample_prior (reference :705-723, :593-607): z = prior_loc + prior_scale * eps."""
        B, r, W, Z = x_in.shape[0], self.base, self.width, self.Z
        sh = lambda c: (B, r, r, c)   # noqa: E731
        ap = self.buf("ap", sh(W))
        ops.gelu_fwd(x_in, None, ap)
        pr = self.prior.forward(ap)
        x1 = self.buf("x1", sh(W))
        ops.add_cols(x_in, pr, 2 * Z, x1)
        # the prior's (loc | raw scale) columns as a [rows, 2Z] posterior-shaped operand of the sampling kernel
        pq = self.buf("prior_as_post", sh(2 * Z))
        zeros = self.buf("zeros_2z", sh(2 * Z))
        ops.fill_zero(zeros)
        ops.add_cols(zeros, pr, 0, pq)
        z, scratch = self.buf("z", sh(Z)), self.buf("kl_scratch", (B,))
        ops.fill_zero(scratch)
        ops.diag_sample_kl_fwd(pq, pr, eps, z, scratch, r * r)
        x2 = self.buf("x2", sh(W))
        ops.layer_forward(self.z_proj.g, z, self.store.p[self.z_proj.w], self.store.p[self.z_proj.b], x2, res=x1,
                          wsplit=self.store.split_view(self.z_proj.ws_f))
        x2g = self.buf("x2g", sh(W))
        ops.gelu_fwd(x2, None, x2g)
        return self.resnet.forward(x2g, res=x2)

    def forward_lls(self, x_all: torch.Tensor, acts: torch.Tensor, macts: torch.Tensor, eps: torch.Tensor,
                    eps_m: torch.Tensor, stats: Dict[str, torch.Tensor]) -> torch.Tensor:
        """forward_lls / sample_lls (reference :725-754, :609-660).  x_all [2B, r, r, W] stacks the two decoder states
        the reference carries - rows [0, B): x (full posterior), rows [B, 2B): masked_x (masked posterior) - so the
        shared prior / z_proj / resnet Blocks run once on 2B rows.  stats[name] [B] accumulate pz, qzx, masked_pz,
        masked_qzx over the blocks."""
        B2, r, W, Z = x_all.shape[0], self.base, self.width, self.Z
        B, P = B2 // 2, r * r
        sh = lambda n, c: (n, r, r, c)   # noqa: E731
        x, mx = x_all[:B], x_all[B:]
        a = self.buf("a", sh(B, 2 * W))
        ops.gelu_fwd(x, acts, a)
        pp = self.posterior.forward(a)
        am = self.buf("am", sh(B, 2 * W))
        ops.gelu_fwd(mx, macts, am)
        mp = self.masked_posterior.forward(am)
        ap = self.buf("ap_lls", sh(B2, W))
        ops.gelu_fwd(x_all, None, ap)
        pr = self.prior.forward(ap)                         # rows [0,B): prior(x), rows [B,2B): prior(masked_x)
        x1 = self.buf("x1_lls", sh(B2, W))
        ops.add_cols(x_all, pr, 2 * Z, x1)
        z_all = self.buf("z_lls", sh(B2, Z))
        z, mz = z_all[:B], z_all[B:]
        scratch, scratch_p = self.buf("kl_scratch", (B,)), self.buf("tril_scratch", (B * P,))
        ops.fill_zero(scratch)
        ops.diag_sample_kl_fwd(pp, pr[:B], eps, z, scratch, P)
        ops.tril_sample_kl_fwd(mp.view(B * P, -1), eps_m.view(B * P, Z), mz.view(B * P, Z), scratch_p)
        ops.diag_logprob_acc(pr[:B], z, stats["pz"], P)
        ops.diag_logprob_acc(pp, z, stats["qzx"], P)
        ops.diag_logprob_acc(pr[B:], mz, stats["masked_pz"], P)
        ops.tril_logprob_fwd(mp.view(B * P, -1), mz.view(B * P, Z), scratch_p)
        ops.segment_wsum(scratch_p, None, stats["masked_qzx"], accumulate=True)
        x2 = self.buf("x2_lls", sh(B2, W))
        ops.layer_forward(self.z_proj.g, z_all, self.store.p[self.z_proj.w], self.store.p[self.z_proj.b], x2, res=x1,
                          wsplit=self.store.split_view(self.z_proj.ws_f))
        x2g = self.buf("x2g_lls", sh(B2, W))
        ops.gelu_fwd(x2, None, x2g)
        return self.resnet.forward(x2g, res=x2)

    kl_stream = None             # set by the model: side stream of the posterior-matching KL kernels
    _kl_ev_pending = False
    _mp_split = False            # forward ran the masked-posterior Block as its own launch on the side stream
    _mp_bwd_done = False

    def masked_backward(self, g_pm: float, dmacts: torch.Tensor) -> None:
        """everything behind pm_kl, on the side stream: d pm_kl / d(masked-posterior parameters) (forward values and a
        constant), the masked-posterior Block's data gradients, its gelu' into the masked encoder's activation gradients.
        Nothing here reads the main chain, so the model issues it for every decoder block when the backward pass starts."""
        B, r, W, Z = self._x_in.shape[0], self.base, self.width, self.Z
        dmp = self.buf("dmp", tuple(self._mp.shape))
        ops.diag_tril_kl_bwd(self._pp, self._mp, g_pm, dmp, Z, r * r)
        dam = self.buf("dam", (B, r, r, 2 * W))
        ops.vdvae_blocks_bwd([self.masked_posterior.bwd_io(dmp, dam, None, None)], B, r, r, self.posterior.mid,
                             self.posterior.c2.g.k)
        self.masked_posterior.weight_grads()
        ops.gelu_bwd(self._x_in, self._macts, dam, None, dmacts, accumulate=True)   # no gradient into stop_gradient(x)
        self._mp_bwd_done = True

    def pm_kl_backward(self, g_pm: float) -> None:
        """d pm_kl / d(masked-posterior parameters) depends on forward values only (the loss weight is a constant): the model
        launches it for every decoder block on the side stream when the backward pass starts, so that the 20 launches are off
        the dependent chain; backward() then only waits for the event"""
        dmp = self.buf("dmp", tuple(self._mp.shape))
        ops.diag_tril_kl_bwd(self._pp, self._mp, g_pm, dmp, self.Z, self.base * self.base)
        if getattr(self, "_kl_ev", None) is None:
            self._kl_ev = torch.cuda.Event()
        ops.record_event(self._kl_ev, torch.cuda.current_stream(dmp.device))
        self._kl_ev_pending = True

    def backward(self, dx3: torch.Tensor, dacts: torch.Tensor, dmacts: torch.Tensor, g_kl: float, g_pm: float,
                 streams=None) -> torch.Tensor:
        """dx3: gradient w.r.t. this block's output.  Accumulates into dacts / dmacts (the encoder
        activations of this resolution; dmacts only ever on streams[1]) and returns the gradient w.r.t. x_in."""
        B, r, W, Z = dx3.shape[0], self.base, self.width, self.Z
        P = r * r
        sh = lambda c: (B, r, r, c)   # noqa: E731
        main = torch.cuda.current_stream(dx3.device)
        s1, s2 = streams if streams is not None else (main, main)
        dx2 = self.buf("dx2", sh(W))
        self.resnet.backward(dx3, dx2, x_pre=self._x2, res=dx3)                 # dx2 = dx1
        self.wgrad(self.z_proj.g, self._z, dx2, self.store.g[self.z_proj.w], self.store.g[self.z_proj.b])
        dpp, dpr = self.buf("dpp", sh(2 * Z)), self.buf("dpr", sh(2 * Z + W))
        if not os.environ.get("PM_VDVAE_NO_SAMPLE_PROJECT"):
            # dz = z_proj^T(dx2), the sample / KL gradients, d h = d x1 = dx2: one launch
            ops.sample_project_bwd(self._pp, self._pr, self._eps, dx2, self.store.p[self.z_proj.w].view(Z, W), g_kl, dpp, dpr)
        else:
            dz = self.buf("dz", sh(Z))
            ops.layer_dgrad(self.z_proj.g, dx2, self.store.p[self.z_proj.w], dz, wsplit=self.store.split_view(self.z_proj.ws_d))
            ops.diag_sample_kl_bwd(self._pp, self._pr, self._eps, dz, g_kl, dpp, dpr)
            ops.copy_cols(dx2, dpr, 2 * Z)                                            # d h = d x1
        dmp = self.buf("dmp", tuple(self._mp.shape))
        if self._mp_bwd_done:
            pass                     # dmp and everything behind it were handled on the side stream
        elif self._kl_ev_pending:    # launched on the side stream when the backward pass started (pm_kl_backward)
            ops.wait_event(main, self._kl_ev)
            self._kl_ev_pending = False
        else:
            ops.diag_tril_kl_bwd(self._pp, self._mp, g_pm, dmp, Z, P)
        if getattr(self, "_grouped", False) and self._mp_bwd_done:
            # the masked-posterior Block's whole backward already ran on the side stream (masked_backward)
            self._mp_bwd_done = False
            da, dxin = self.buf("da", sh(2 * W)), self.buf("dxin", sh(W))
            ios = [self.posterior.bwd_io(dpp, da, None, None), self.prior.bwd_io(dpr, dxin, self._x_in, dx2)]
            ops.vdvae_blocks_bwd(ios, B, r, r, self.posterior.mid, self.posterior.c2.g.k)
            self.posterior.weight_grads()
            self.prior.weight_grads()
            ops.gelu_bwd(self._x_in, self._acts, da, dxin, dacts, accumulate=True)
            return dxin
        if getattr(self, "_grouped", False):
            da, dam, dxin = self.buf("da", sh(2 * W)), self.buf("dam", sh(2 * W)), self.buf("dxin", sh(W))
            ios = [self.posterior.bwd_io(dpp, da, None, None), self.masked_posterior.bwd_io(dmp, dam, None, None),
                   self.prior.bwd_io(dpr, dxin, self._x_in, dx2)]           # prior: + the direct x1 = x_in + h path
            ops.vdvae_blocks_bwd(ios, B, r, r, self.posterior.mid, self.posterior.c2.g.k)
            for blk in (self.posterior, self.masked_posterior, self.prior):
                blk.weight_grads()
            ks = self.kl_stream
            if ks is not None:       # dmacts only feeds the masked encoder's backward at the very end: off the chain
                ops.wait_stream(ks, main)
                with torch.cuda.stream(ks):
                    ops.gelu_bwd(self._x_in, self._macts, dam, None, dmacts, accumulate=True)
            else:
                ops.gelu_bwd(self._x_in, self._macts, dam, None, dmacts, accumulate=True)   # no gradient into stop_gradient(x)
            ops.gelu_bwd(self._x_in, self._acts, da, dxin, dacts, accumulate=True)
            return dxin
        if streams is not None:
            ops.wait_stream(s1, main)
            ops.wait_stream(s2, main)
        with torch.cuda.stream(s1):
            da = self.buf("da", sh(2 * W))
            self.posterior.backward(dpp, da)
        with torch.cuda.stream(s2):
            dam = self.buf("dam", sh(2 * W))
            self.masked_posterior.backward(dmp, dam)
            ops.gelu_bwd(self._x_in, self._macts, dam, None, dmacts, accumulate=True)   # no gradient into stop_gradient(x)
        dxin = self.buf("dxin", sh(W))
        self.prior.backward(dpr, dxin, x_pre=self._x_in, res=dx2)               # + the direct x1 = x_in + h path
        if streams is not None:
            ops.wait_stream(main, s1)
        ops.gelu_bwd(self._x_in, self._acts, da, dxin, dacts, accumulate=True)
        return dxin


class PosteriorMatchingVDVAE(Module):
    """A "Very Deep VAE" modified for Posterior Matching (reference vdvae.py:15-94)."""

    def __init__(self, image_shape: Tuple[int, int, int], encoder_blocks: str, decoder_blocks: str, latent_dim: int = 16,
                 width: int = 128, bottleneck_multiple: float = 0.25, no_bias_above: int = 64, num_mixtures: int = 10,
                 custom_width_string: Optional[str] = None, name: Optional[str] = None, device: Optional[str] = None,
                 seed: int = 1):
        super().__init__(name)
        image_shape = tuple(image_shape)
        if image_shape[-1] != 1:
            raise NotImplementedError("LogisticMixture with num_channels > 1 (channel coefficients) has no HIP path")
        if custom_width_string:
            raise NotImplementedError("custom_width_string has no HIP path")
        self.config = dict(image_shape=image_shape, encoder_blocks=encoder_blocks, decoder_blocks=decoder_blocks,
                           latent_dim=latent_dim, width=width, bottleneck_multiple=bottleneck_multiple,
                           no_bias_above=no_bias_above, num_mixtures=num_mixtures, custom_width_string=custom_width_string)
        self.encoder = Encoder(width, encoder_blocks, bottleneck_multiple)
        self.masked_encoder = Encoder(width, encoder_blocks, bottleneck_multiple)
        self._device, self._seed = device, seed
        self.store: Optional[ParamStore] = None
        self.concurrent = True      # independent chains (the two encoders; the three Blocks of a decoder block) on companion streams
        self._streams = None

    def _kl_stream(self, device):
        if not self.concurrent or os.environ.get("PM_VDVAE_KL_INLINE"):
            return None
        if getattr(self, "_kls", None) is None:
            self._kls = torch.cuda.Stream(device=device)
        return self._kls

    def _branch_streams(self, device):
        if not self.concurrent:
            return None
        if self._streams is None:
            self._streams = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        return self._streams

    def init(self, device=None, seed: Optional[int] = None) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("posterior_matching_amd needs an MI355X: there is no CPU fallback path")
        from .. import _lib

        _lib.load()
        c = self.config
        device = torch.device(device or self._device or "cuda:0")
        store, ws = ParamStore(), Workspace(device)
        self.ws = ws
        H, W_, C = c["image_shape"]
        width, Z = c["width"], c["latent_dim"]
        mid = int(width * c["bottleneck_multiple"])
        self.encoder.ws = self.masked_encoder.ws = ws
        self.encoder.build(store, "encoder", (H, W_, C))
        self.masked_encoder.build(store, "masked_encoder", (H, W_, 2 * C))
        spec = parse_layer_string(c["decoder_blocks"])
        self.dec_blocks = [PosteriorMatchingDecoderBlock(store, ws, f"decoder/block_{i}", Z, r, m, len(spec), width, mid)
                           for i, (r, m) in enumerate(spec)]
        self.resolutions = sorted({r for r, _ in spec})
        self.bias_res = [r for r in self.resolutions if r <= c["no_bias_above"]]
        for r in self.bias_res:
            store.add(f"decoder/x_bias_{r}", (1, r, r, width))      # reference name: "x_bias_{res}]" (stray bracket, :797)
        nm = c["num_mixtures"]
        self.out_net = _Conv(store, "decoder/out_net", LayerGeom.conv(H, W_, width, nm * 3, 1, 1, "SAME"), width)
        store.add("decoder/gain", (1, 1, 1, width), fan_in=-3)      # Constant(1.0)
        store.add("decoder/bias", (1, 1, 1, width))
        store.allocate(device, self._seed if seed is None else seed)
        self.store = store
        self.attach(store, "decoder")
        self.metrics = torch.zeros(8, device=device)

    @property
    def num_params(self) -> int:
        return self.store.num_params

    def eps_shapes(self, B: int) -> List[Tuple[int, ...]]:
        Z = self.config["latent_dim"]
        return [(B, blk.base, blk.base, Z) for blk in self.dec_blocks]

    def __call__(self, x: torch.Tensor, b: torch.Tensor, eps: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """reference vdvae.py:76-94.  x: raw pixel values 0..255 [B,H,W,1]; b: mask (1 = observed);
        eps: one N(0,1) draw [B,res,res,Z] per decoder block (the caller owns the RNG).  Returns per-example
        reconstruction_ll, kl, pm_kl (device tensors owned by the model)."""
        if self.store is None:
            self.init(x.device)
        c = self.config
        B, H, W_, _ = x.shape
        width, nm = c["width"], c["num_mixtures"]
        self._B, self._x = B, x
        xn = self.ws.get("vdvae/xn", tuple(x.shape))
        ops.scale_shift(x, 1.0 / 127.5, -1.0, xn)
        xob = self.ws.get("vdvae/x_o_b", (B, H, W_, 2))
        ops.mask_concat(xn, b, xob)
        streams = self._branch_streams(x.device)
        main = torch.cuda.current_stream(x.device)
        self._paired = _pairable(self.encoder, self.masked_encoder)
        if self._paired:
            acts, macts = run_encoders(self.encoder, self.masked_encoder, xn, xob)
        elif streams is not None:
            ops.wait_stream(streams[0], main)
            with torch.cuda.stream(streams[0]):
                macts = self.masked_encoder(xob)
            acts = self.encoder(xn)
            ops.wait_stream(main, streams[0])
        else:
            acts = self.encoder(xn)
            macts = self.masked_encoder(xob)
        self._acts, self._macts = acts, macts
        kl, pm_kl = self.ws.get("vdvae/kl", (B,)), self.ws.get("vdvae/pm_kl", (B,))
        ops.fill_zero(kl)
        ops.fill_zero(pm_kl)
        xs: Dict[int, torch.Tensor] = {}
        self._first_use: Dict[int, int] = {}
        self._x_ins: List[torch.Tensor] = []
        kls = self._kl_stream(x.device)
        for i, blk in enumerate(self.dec_blocks):
            r = blk.base
            if r in xs:
                x_in = xs[r]
            else:
                x_in = self._start_state(xs, r, B)     # jnp.repeat(x_bias, B, axis=0) (:669-670) or zeros_like(acts) (:667-668)
                self._first_use[r] = i
            if blk.mixin is not None:
                ops.resize_nearest_add(xs[blk.mixin], x_in)
            blk.kl_stream = kls
            xs[r] = blk.forward(x_in, acts[r], macts[r], eps[i], kl, pm_kl, streams=streams)
        if kls is not None:
            ops.wait_stream(main, kls)                 # pm_kl is complete before the loss reads it
        top = xs[H]
        self._top = top
        px_z = self.ws.get("decoder/px_z", tuple(top.shape))
        ops.affine_fwd(top, self.store.p["decoder/gain"], self.store.p["decoder/bias"], px_z)
        self._px_z = px_z
        params = self.ws.get("decoder/dmol_params", (B, H, W_, 3 * nm))
        ops.layer_forward(self.out_net.g, px_z, self.store.p[self.out_net.w], self.store.p[self.out_net.b], params,
                          wsplit=self.store.split_view(self.out_net.ws_f))
        self._params = params
        rec = self.ws.get("vdvae/rec_ll", (B,))
        ops.dmol_ll_fwd(params, x, rec, nm, H * W_)
        ops.vdvae_loss(rec, kl, pm_kl, float(H * W_ * c["image_shape"][-1]), self.metrics)
        return {"reconstruction_ll": rec, "kl": kl, "pm_kl": pm_kl}

    def _start_state(self, xs, r: int, B: int) -> torch.Tensor:
        width = self.config["width"]
        x_in = self.ws.get(f"decoder/x_start_{r}", (B, r, r, width))
        if r in self.bias_res:
            ops.broadcast_rows(self.store.p[f"decoder/x_bias_{r}"], x_in)
        else:
            ops.fill_zero(x_in)
        return x_in

    def impute(self, x: torch.Tensor, b: torch.Tensor, num_samples: int = 100, seed: int = 0,
               eps: Optional[Sequence[Sequence[torch.Tensor]]] = None) -> torch.Tensor:
        """reference vdvae.py:161-186: [B, num_samples, H, W, C] - the decoder driven by the masked posterior
        only, observed pixels kept.  eps[s][i]: explicit noise of sample s / block i (parity mode); otherwise
        device Philox keyed by (seed, s)."""
        if self.store is None:
            self.init(x.device)
        c = self.config
        B, H, W_, C = x.shape
        nm = c["num_mixtures"]
        xn = self.ws.get("vdvae/xn", tuple(x.shape))
        ops.scale_shift(x, 1.0 / 127.5, -1.0, xn)
        xob = self.ws.get("vdvae/x_o_b", (B, H, W_, 2))
        ops.mask_concat(xn, b, xob)
        macts = self.masked_encoder(xob)                    # identical for every sample: the scan body is deterministic in it
        out = torch.empty((B, num_samples, H, W_, C), device=x.device)
        shapes = self.eps_shapes(B)
        noise = [self.ws.get(f"vdvae/impute_eps_{i}", sh) for i, sh in enumerate(shapes)] if eps is None else None
        mean = self.ws.get("vdvae/impute_mean", (B, H, W_, C))
        for s in range(num_samples):
            xs: Dict[int, torch.Tensor] = {}
            for i, blk in enumerate(self.dec_blocks):
                r = blk.base
                x_in = xs[r] if r in xs else self._start_state(xs, r, B)
                if blk.mixin is not None:
                    ops.resize_nearest_add(xs[blk.mixin], x_in)
                if eps is None:
                    ops.normal_fill(noise[i], seed + 7919 * s, None, stream_id=i)
                    e = noise[i]
                else:
                    e = eps[s][i]
                xs[r] = blk.forward_partial(x_in, macts[r], e)
            px_z = self.ws.get("decoder/px_z", tuple(xs[H].shape))
            ops.affine_fwd(xs[H], self.store.p["decoder/gain"], self.store.p["decoder/bias"], px_z)
            params = self.ws.get("decoder/dmol_params", (B, H, W_, 3 * nm))
            ops.layer_forward(self.out_net.g, px_z, self.store.p[self.out_net.w], self.store.p[self.out_net.b], params,
                              wsplit=self.store.split_view(self.out_net.ws_f))
            ops.dmol_mean(params, mean, nm)
            out[:, s].copy_(mean)
        ops.impute_blend(x, b, out, 1.0, 0.0)               # jnp.where(b == 1, x, mean): no clipping
        return out

    def sample(self, num_samples: int, seed: int = 0, eps: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """reference vdvae.py:148-159: unconditional samples, [num_samples, H, W, C] (decoder mean of a prior draw).
        eps[i] [N,res,res,Z]: explicit prior noise (parity mode); otherwise device Philox keyed by seed."""
        if self.store is None:
            self.init(self._device)
        c = self.config
        N, nm = int(num_samples), c["num_mixtures"]
        H, W_, C = c["image_shape"]
        dev = self.store.device
        xs: Dict[int, torch.Tensor] = {}
        for i, blk in enumerate(self.dec_blocks):
            r = blk.base
            x_in = xs[r] if r in xs else self._start_state(xs, r, N)
            if blk.mixin is not None:
                ops.resize_nearest_add(xs[blk.mixin], x_in)
            if eps is None:
                e = self.ws.get(f"vdvae/sample_eps_{i}", (N, r, r, c["latent_dim"]))
                ops.normal_fill(e, seed, None, stream_id=i)
            else:
                e = eps[i]
            xs[r] = blk.forward_prior(x_in, e)
        px_z = self.ws.get("decoder/px_z", tuple(xs[H].shape))
        ops.affine_fwd(xs[H], self.store.p["decoder/gain"], self.store.p["decoder/bias"], px_z)
        params = self.ws.get("decoder/dmol_params", (N, H, W_, 3 * nm))
        ops.layer_forward(self.out_net.g, px_z, self.store.p[self.out_net.w], self.store.p[self.out_net.b], params,
                          wsplit=self.store.split_view(self.out_net.ws_f))
        out = torch.empty((N, H, W_, C), device=dev)
        ops.dmol_mean(params, out, nm)
        return out

    def is_log_probs(self, x: torch.Tensor, b: torch.Tensor, num_samples: int = 100, seed: int = 0,
                     eps: Optional[Sequence[Sequence[torch.Tensor]]] = None,
                     eps_masked: Optional[Sequence[Sequence[torch.Tensor]]] = None):
        """reference vdvae.py:96-146: importance-sampled (log p(x), log p(x_u | x_o)), each [B].  One decoder pass per
        sample over the stacked states (see PosteriorMatchingDecoderBlock.forward_lls); eps[s][i] / eps_masked[s][i] are
        the explicit draws of sample s, block i (parity mode), otherwise device Philox keyed by (seed, s)."""
        if self.store is None:
            self.init(x.device)
        c = self.config
        B, H, W_, C = x.shape
        S, nm, Z = int(num_samples), c["num_mixtures"], c["latent_dim"]
        xn = self.ws.get("vdvae/xn", tuple(x.shape))
        ops.scale_shift(x, 1.0 / 127.5, -1.0, xn)
        xob = self.ws.get("vdvae/x_o_b", (B, H, W_, 2))
        ops.mask_concat(xn, b, xob)
        acts = self.encoder(xn)
        macts = self.masked_encoder(xob)
        names = ("pz", "qzx", "masked_pz", "masked_qzx")
        stats = {n: self.ws.get(f"vdvae/is_{n}", (B,)) for n in names}
        comb = self.ws.get("vdvae/is_comb", (2, S, B))       # [0]: pz - qzx, [1]: masked_pz - masked_qzx, per sample
        lls = self.ws.get("vdvae/is_lls", (2, S, B))         # [0]: log p(x | z), [1]: observed-pixel log p(x_o | z)
        pix = self.ws.get("vdvae/is_pix_ll", (B * H * W_,))
        noise = [[self.ws.get(f"vdvae/is_eps_{k}_{i}", (B, blk.base, blk.base, Z)) for i, blk in enumerate(self.dec_blocks)]
                 for k in range(2)] if eps is None else None
        for s in range(S):
            for n in names:
                ops.fill_zero(stats[n])
            xs: Dict[int, torch.Tensor] = {}
            for i, blk in enumerate(self.dec_blocks):
                r = blk.base
                x_in = xs[r] if r in xs else self._start_state(xs, r, 2 * B)
                if blk.mixin is not None:
                    ops.resize_nearest_add(xs[blk.mixin], x_in)
                if eps is None:
                    ops.normal_fill(noise[0][i], seed + 7919 * s, None, stream_id=2 * i)
                    ops.normal_fill(noise[1][i], seed + 7919 * s, None, stream_id=2 * i + 1)
                    e, em = noise[0][i], noise[1][i]
                else:
                    e, em = eps[s][i], eps_masked[s][i]
                xs[r] = blk.forward_lls(x_in, acts[r], macts[r], e, em, stats)
            px_z = self.ws.get("decoder/px_z", tuple(xs[H].shape))
            ops.affine_fwd(xs[H], self.store.p["decoder/gain"], self.store.p["decoder/bias"], px_z)
            params = self.ws.get("decoder/dmol_params", (2 * B, H, W_, 3 * nm))
            ops.layer_forward(self.out_net.g, px_z, self.store.p[self.out_net.w], self.store.p[self.out_net.b], params,
                              wsplit=self.store.split_view(self.out_net.ws_f))
            ops.dmol_ll_fwd(params[:B], x, lls[0, s], nm, H * W_)
            ops.dmol_ll_fwd(params[B:], x, pix, nm, 1)                               # log_prob(x, independent=False)
            ops.segment_wsum(pix, b, lls[1, s])                                      # sum over the observed pixels
            # px = pxz_ll + pz - qzx ; pxo = pxoz_ll + masked_pz - masked_qzx   (:134-140)
            ops.scale_shift(stats["qzx"], -1.0, 0.0, comb[0, s])
            ops.axpy1(stats["pz"], comb[0, s])
            ops.scale_shift(stats["masked_qzx"], -1.0, 0.0, comb[1, s])
            ops.axpy1(stats["masked_pz"], comb[1, s])
        px, pxo = self.ws.get("vdvae/is_log_px", (B,)), self.ws.get("vdvae/is_log_pxo", (B,))
        ops.logmeanexp3(lls[0], comb[0], None, px, S, sample_major=True)            # reduce_logmeanexp over the samples
        ops.logmeanexp3(lls[1], comb[1], None, pxo, S, sample_major=True)
        neg = self.ws.get("vdvae/is_neg", (B, 1))
        ops.scale_shift(pxo.view(B, 1), -1.0, 0.0, neg)
        out = self.ws.get("vdvae/is_pxu_xo", (B, 1))
        ops.add_cols(px.view(B, 1), neg, 0, out)
        return px, out.view(B)

    def reconstruction(self) -> torch.Tensor:
        """decoder_dist.mean() of the last call (reference :93)"""
        B, H, W_, _ = self._x.shape
        out = torch.empty((B, H, W_, 1), device=self._x.device)
        ops.dmol_mean(self._params, out, self.config["num_mixtures"])
        return out

    def backward(self, grad_scale: float = 1.0) -> None:
        """Gradient of loss = -mean(rec_ll - kl) + mean(pm_kl) (train_pm_vdvae.py:112-118) times grad_scale,
        accumulated into the flat gradient buffer (zero it first)."""
        c = self.config
        B, H, W_, _ = self._x.shape
        width, nm = c["width"], c["num_mixtures"]
        g = grad_scale / B
        import os

        batched = self.store.use_bf16 and not os.environ.get("PM_NO_WGRAD_BATCH") and not os.environ.get("PM_NO_VDVAE_FUSED")
        if batched:
            if getattr(self, "_wgrad_batch", None) is None:
                self._wgrad_batch = ops.WgradBatch()
            self._wgrad_batch.reducer = self.store.reducer     # data-parallel: every grouped launch reports its weight ranges
            self.ws.wgrad_batch = self._wgrad_batch
        try:
            self._backward_body(g, B, H, W_, width, nm)
        finally:
            if batched:                 # never leave the shared workspace in deferred mode (evaluation paths share it)
                self._wgrad_batch.discard()
                self.ws.wgrad_batch = None

    def _backward_body(self, g, B, H, W_, width, nm) -> None:
        import os

        dparams = self.ws.get("decoder/d_dmol_params", tuple(self._params.shape))
        ops.dmol_ll_bwd(self._params, self._x, -g, dparams, nm, H * W_)
        self.wgrad(self.out_net.g, self._px_z, dparams, self.store.g[self.out_net.w], self.store.g[self.out_net.b])
        if self.store.reducer is not None and self.ws.wgrad_batch is None:     # (batched: the flush reports it)
            self.ws.join_all_aux()
            self.store.grads_ready(["decoder/out_net"])
        dpx = self.ws.get("decoder/d_px_z", tuple(self._px_z.shape))
        ops.layer_dgrad(self.out_net.g, dparams, self.store.p[self.out_net.w], dpx, wsplit=self.store.split_view(self.out_net.ws_d))
        dxs: Dict[int, torch.Tensor] = {}
        for r in self.resolutions:
            t = self.ws.get(f"decoder/dxs_{r}", (B, r, r, width))
            ops.fill_zero(t)
            dxs[r] = t
        ops.affine_bwd(self._top, self.store.p["decoder/gain"], dpx, dxs[H], self.store.g["decoder/gain"],
                       self.store.g["decoder/bias"])
        dacts = {r: self.ws.get(f"vdvae/dacts_{r}", tuple(t.shape)) for r, t in self._acts.items()}
        dmacts = {r: self.ws.get(f"vdvae/dmacts_{r}", tuple(t.shape)) for r, t in self._macts.items()}
        for t in list(dacts.values()) + list(dmacts.values()):
            ops.fill_zero(t)
        streams = self._branch_streams(self._x.device)
        main = torch.cuda.current_stream(self._x.device)
        # Weight gradients are grouped per (resolution, layer shape), so the groups of a resolution are complete when the
        # backward chain leaves it: they start then on their own stream, beside the rest of the chain (dependent small
        # launches that leave most of the chip idle) - same number of launches as one flush at the very end.
        flush_stream = None
        if self.ws.wgrad_batch is not None and not os.environ.get("PM_VDVAE_ONE_FLUSH"):
            if getattr(self, "_flush_stream", None) is None:
                self._flush_stream = torch.cuda.Stream(device=self._x.device)
            flush_stream = self._flush_stream

        def flush_side():
            ops.wait_stream(flush_stream, main)
            if ks is not None:
                ops.wait_stream(flush_stream, ks)       # the masked-posterior Blocks' operands were produced there
            if streams is not None:
                ops.wait_stream(flush_stream, streams[0])
                ops.wait_stream(flush_stream, streams[1])
            with torch.cuda.stream(flush_stream):
                self.ws.wgrad_batch.flush()

        ks = self._kl_stream(self._x.device)
        if ks is not None:
            ops.wait_stream(ks, main)
            with torch.cuda.stream(ks):
                for blk in reversed(self.dec_blocks):
                    if blk._mp_split and self.ws.wgrad_batch is not None:
                        blk.masked_backward(g, dmacts[blk.base])
                    else:
                        blk.pm_kl_backward(g)
        for i in reversed(range(len(self.dec_blocks))):
            blk = self.dec_blocks[i]
            r = blk.base
            if flush_stream is not None and i + 1 < len(self.dec_blocks) and self.dec_blocks[i + 1].base != r:
                flush_side()                            # the finer resolution is finished
            dxin = blk.backward(dxs[r], dacts[r], dmacts[r], g, g, streams=streams)
            if self.store.reducer is not None and self.ws.wgrad_batch is None:   # data-parallel: block i's weight gradients are final once the
                self.ws.join_all_aux()                  # main chain and the masked-posterior chain (streams[1]) got here
                self.store.grads_ready([f"decoder/block_{i}"], streams=None if streams is None else (main, streams[1]))
            if blk.mixin is not None:
                ops.resize_nearest_add_bwd(dxin, dxs[blk.mixin])
            if self._first_use[r] == i:
                if r in self.bias_res:
                    ops.groups_sum(dxin, self.store.g[f"decoder/x_bias_{r}"], B, accumulate=True)
            else:
                # x_in was the previous block of this resolution's output, and this block its only consumer (mix-in
                # sources are always the LAST state of a coarser resolution): hand dxin over as that block's dx3
                dxs[r] = dxin
        if ks is not None:
            ops.wait_stream(main, ks)                   # the masked-encoder activation gradients accumulated on the side stream
        if flush_stream is not None:
            flush_side()                                # the coarsest decoder resolution; the encoders' follow at the end
        if getattr(self, "_paired", False):
            if streams is not None:
                ops.wait_stream(main, streams[0])
                ops.wait_stream(main, streams[1])
            run_encoders_backward(self.encoder, self.masked_encoder, dacts, dmacts)
        elif streams is not None:
            s1, s2 = streams
            ops.wait_stream(s1, s2)                       # dmacts were accumulated on s2
            ops.wait_stream(s1, main)
            with torch.cuda.stream(s1):
                self.masked_encoder.backward(dmacts)
            self.encoder.backward(dacts)
            ops.wait_stream(main, s1)
            ops.wait_stream(main, s2)
        else:
            self.encoder.backward(dacts)
            self.masked_encoder.backward(dmacts)
        if self.ws.wgrad_batch is not None:
            # every Block has left its operands in HBM: their weight gradients, one launch per (resolution, layer shape)
            self.ws.wgrad_batch.flush()
        if flush_stream is not None:
            ops.wait_stream(main, flush_stream)
        self.ws.join_all_aux()

    def zero_grad(self) -> None:
        self.store.zero_grad()

    def params_dict(self):
        return self.store.to_dict("p")

    def grads_dict(self):
        return self.store.to_dict("g")

    def load_params(self, values) -> None:
        self.store.load_dict(values)


def vdvae_imputation_psnr(imputations: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """eval_pm_vdvae_imputation.py:123-128: PSNR [B] of the mean imputation, pixel values scaled by 1/255."""
    psnr = torch.empty(x.shape[0], device=x.device)
    ops.imputation_psnr(imputations, x, psnr, 1.0 / 255.0)
    return psnr
