"""Parameter store, feature handle and workspace shared by the host-side model classes.

The reference keeps parameters in haiku pytrees owned by bax.TrainState; here they live in ONE
flat float32 device buffer (weights first, 1-D leaves last, so that the optimizer's
`add_decayed_weights(mask = ndim != 1)` of train_pm_vae.py:76-79 is a prefix of the buffer), with
matching flat buffers for gradients and the two Adam moments.  288 GB of HBM per GPU make it
pointless to free or re-use activations: every layer keeps its own buffers, allocated once per
batch size.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import ops
from ..ops import ACT_NONE


def truncated_normal(rng, shape) -> np.ndarray:
    """N(0, 1) truncated to [-2, 2] by inverse-CDF sampling: the values scipy.stats.truncnorm.rvs(-2, 2, size=shape,
    random_state=rng) returns (same uniform draws, equal to 2e-15), 1000x faster (scipy's ppf took 41 s for the PixelCNN's
    35 M parameters)."""
    from scipy.special import ndtr, ndtri

    lo, hi = ndtr(-2.0), ndtr(2.0)
    return ndtri(lo + rng.uniform(size=shape) * (hi - lo))


@dataclass
class Feat:
    """A network output handed to its consumer.

    t        : the stored tensor
    in_act   : activation the consumer must still apply when it loads `t` (ResidualMLP keeps the
               pre-activation h and lets the next dense layer apply relu on load)
    grad_act : activation whose derivative, evaluated on `t`, turns a gradient w.r.t. the
               activated value into a gradient w.r.t. the producer's pre-activation
    """

    t: torch.Tensor
    in_act: int = ACT_NONE
    grad_act: int = ACT_NONE


class _GradViews(dict):
    """name -> view of the flat gradient buffer.  Handing a view out marks the buffer as (possibly) written: the train
    steps skip their zero-fill launch only while `store.g_clean` says that the optimizer kernel has zeroed the buffer
    (pm_adam_cfg.zero_grad) and no host code has asked for a gradient view since."""

    def __init__(self, store):
        super().__init__()
        self._store = store

    def __getitem__(self, name):
        self._store.g_clean = False
        return dict.__getitem__(self, name)


class ParamStore:
    def __init__(self):
        self.specs: "OrderedDict[str, Tuple[Tuple[int, ...], int]]" = OrderedDict()
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = _GradViews(self)
        self.g_clean = False           # True: flat_g is known to be all zeros (see _GradViews)
        self.flat_p = self._flat_g = self.flat_m = self.flat_v = None
        self.partials = None           # partials.PartialSums: weight-gradient launches leave per-split partial sums
        self.n_decay = 0
        self.device = None
        self._split_requests = []      # (param name, LayerGeom, mode, groups, group kw)
        self._split_views = []
        self.use_bf16 = True           # bf16x3 matrix-core path for layers that qualify
        self.reducer = None            # parallel.GradReducer when the step runs data-parallel (set by the train step)

    # The flat gradient buffer.  Weight-gradient launches store per-split partial sums into arenas (self.partials) and the
    # train steps add them with reduce_partials() in front of the optimizer; host code that reads the buffer any other way
    # (tests, tools) gets the same contract through this property: whatever is pending is added first (device-synchronous).
    @property
    def flat_g(self):
        if self.partials is not None and self.partials.pending:
            self.partials.flush_sync()
        return self._flat_g

    @flat_g.setter
    def flat_g(self, value):
        self._flat_g = value

    def reduce_partials(self, lo: int = 0, hi=None) -> None:
        """adds the pending partial sums of gradient elements [lo, hi) into the flat buffer on the CURRENT stream (which the
        caller has ordered behind the weight-gradient launches)"""
        if self.partials is not None and self.partials.pending:
            self.partials.reduce(lo, hi)

    def zero_grad(self) -> None:
        """zero gradients: pending partial sums are dropped, the flat buffer is zero-filled"""
        if self.partials is not None:
            self.partials.pending.clear()
        ops.fill_zero(self._flat_g)

    def grads_ready(self, prefixes, streams=None) -> None:
        """A model's backward reports that every parameter under `prefixes` has its final gradient (once the work
        enqueued so far on `streams` / the current stream has run): the data-parallel reducer may all-reduce that
        range while the rest of the backward pass runs.  No-op on one GPU."""
        if self.reducer is not None:
            self.reducer.ready(prefixes, streams)

    def add(self, name: str, shape, fan_in: int = 0) -> None:
        """fan_in > 0: haiku TruncatedNormal(stddev = 1/sqrt(fan_in)); 0: zeros; -1: TruncatedNormal(stddev 1)
        (hk.Embed); -2: RandomNormal(stddev 1) (PixelCNN conditional projections, pixel_cnn.py:567); -3: ones
        (VDVAE gain, vdvae.py:807-809)."""
        assert name not in self.specs, name
        assert self.flat_p is None, "parameters are frozen once allocated"
        self.specs[name] = (tuple(int(s) for s in shape), int(fan_in))

    @property
    def num_params(self) -> int:
        return sum(int(np.prod(s)) for s, _ in self.specs.values())

    def allocate(self, device, seed: int = 1) -> None:
        """Allocates the flat buffers and draws haiku-default initial values (SURVEY A1/A2/A4:
        N(0,1) truncated to [-2,2] times 1/sqrt(fan_in); biases and log_scale zero)."""
        self.device = device
        rng = np.random.default_rng(seed)
        host = {}
        for name, (shape, fan_in) in self.specs.items():      # creation order fixes the RNG stream
            if fan_in > 0:
                host[name] = (truncated_normal(rng, shape) / math.sqrt(fan_in)).astype(np.float32)
            elif fan_in == -1:
                host[name] = truncated_normal(rng, shape).astype(np.float32)
            elif fan_in == -2:
                host[name] = rng.normal(size=shape).astype(np.float32)
            elif fan_in == -3:
                host[name] = np.ones(shape, np.float32)
            else:
                host[name] = np.zeros(shape, np.float32)
        decayed = [n for n, (s, _) in self.specs.items() if len(s) != 1]
        plain = [n for n, (s, _) in self.specs.items() if len(s) == 1]
        # every tensor starts on a 16-byte boundary (its size rounded up to 4 floats; the pad stays zero in all four buffers:
        # zero gradient -> zero update): a 30-wide bias in the middle of the buffer used to leave everything behind it 8-byte
        # aligned, which sent those layers to the dword epilogues / scalar loads of every kernel that checks alignment
        pad4 = (lambda n: n) if os.environ.get("PM_NO_PARAM_PAD") else (lambda n: (n + 3) // 4 * 4)   # (A/B switch)
        total = sum(pad4(int(np.prod(self.specs[n][0]))) for n in decayed + plain)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=device)
        self._flat_g = torch.zeros_like(self.flat_p)
        if torch.device(device).type == "cuda":
            from ..partials import PartialSums

            self.partials = PartialSums(self._flat_g)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        off = 0
        self.offsets = {}
        self.n_decay = 0
        decayed_set = set(decayed)
        for name in decayed + plain:
            shape = self.specs[name][0]
            n = int(np.prod(shape))
            self.offsets[name] = (off, n)
            self.p[name] = self.flat_p[off:off + n].view(shape)
            self.g[name] = self._flat_g[off:off + n].view(shape)
            self.p[name].copy_(torch.from_numpy(host[name]))
            off += pad4(n)
            if name in decayed_set:
                self.n_decay = off                            # weights first: the decayed prefix ends behind the last of them
        assert off == total
        self._build_split_plan()
        self.split_all()

    # ---- pre-split bf16 weight copies for the bf16x3 matrix-core path (pm_split_weights) ----------
    def request_split(self, param_name: str, geom, mode: str, **group_kw) -> int:
        """Registers a K-contiguous hi/lo bf16 copy of `param_name` for `mode` ('fwd' | 'dgrad') of
        `geom`; returns a handle for split_view().  Must be called before allocate()."""
        self._split_requests.append((param_name, geom, mode, group_kw))
        return len(self._split_requests) - 1

    def split_view(self, handle: int):
        if not self.use_bf16 or not self._split_views:
            return None
        return self._split_views[handle]

    def _build_split_plan(self) -> None:
        import ctypes as C

        from .._lib import SplitJob

        jobs, views, off, blk = [], [], 0, 0
        for pname, geom, mode, kw in self._split_requests:
            d = geom._desc(1, mode, **{k: v for k, v in kw.items() if k not in ("B", "dense_k")})
            groups = d.groups
            if d.C % 8 != 0:
                views.append(None)
                continue
            npad = (d.N + 31) // 32 * 32
            dense = bool(kw.get("dense_k"))
            if dense:                                                   # K = (tap, c) without per-tap padding (pm_split_job.dense_k)
                plane = ((d.KH * d.KW * d.C + 31) // 32 * 32) * npad
            else:
                plane = d.KH * d.KW * ((d.C + 31) // 32 * 32) * npad      # channels zero-padded to whole 32-chunks per tap
            start = off
            for gi in range(groups):
                j = SplitJob()
                j.src_off = self.offsets[pname][0] + gi * d.w_gs
                j.dst_off, j.plane = off, plane
                j.taps, j.C, j.N, j.npad = d.KH * d.KW, d.C, d.N, npad
                j.wts, j.wcs, j.wns = d.wts, d.wcs, d.wns
                j.kw, j.kws = d.KW, d.kws
                j.dense_k = int(dense)
                j.first_block, j.num_blocks = blk, plane // 1024        # one workgroup per 32 x 32 tile
                blk += j.num_blocks
                off += 2 * plane
                jobs.append(j)
            views.append((start, off))
        self._split_total_blocks = blk
        self._split_njobs = len(jobs)
        if not jobs:
            self._split_views = [None] * len(views)
            return
        self.split_buf = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self._split_views = [None if v is None else self.split_buf[v[0]:v[1]] for v in views]
        raw = bytes(bytearray(b"".join(bytes(j) for j in jobs)))
        self._split_jobs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)

    def split_all(self) -> None:
        """Refreshes every pre-split copy from the current parameters: ONE launch (after each update)."""
        if self.use_bf16 and getattr(self, "_split_njobs", 0):
            ops.split_weights(self.flat_p, self.split_buf, self._split_jobs_dev, self._split_njobs,
                              self._split_total_blocks)

    def load_dict(self, values: Dict[str, "np.ndarray | torch.Tensor"]) -> None:
        for name, val in values.items():
            t = torch.as_tensor(np.asarray(val.detach().cpu() if isinstance(val, torch.Tensor) else val),
                                dtype=torch.float32)
            self.p[name].copy_(t.reshape(self.p[name].shape))
        self.split_all()

    def load_matching(self, values, require_all: bool, what: str = "parameters") -> set:
        """load_dict of the entries of `values` that name parameters of this store; returns the names it used.
        require_all: a checkpoint that does not cover the whole store (another prefix, a missing leaf) raises KeyError with
        the missing names instead of leaving those parameters at their random initial values."""
        used = {k for k in values if k in self.specs}
        missing = [n for n in self.specs if n not in used]
        if require_all and missing:
            raise KeyError(f"checkpoint lacks {len(missing)} of the {len(self.specs)} {what}: {missing[:8]}"
                           f"{' ...' if len(missing) > 8 else ''}")
        self.load_dict({k: values[k] for k in used})
        return used

    def to_dict(self, which: str = "p") -> Dict[str, torch.Tensor]:
        flat = {"p": self.flat_p, "g": self.flat_g, "m": self.flat_m, "v": self.flat_v}[which]
        return OrderedDict((n, flat[o:o + c].view(self.specs[n][0]).detach().clone()) for n, (o, c) in
                           ((n, self.offsets[n]) for n in self.specs))


class Workspace:
    """Named device buffers allocated on first use; shapes are fixed per batch size, so a
    captured HIP graph always sees the same addresses."""

    def __init__(self, device):
        self.device = device
        self._bufs: Dict[Tuple[str, Tuple[int, ...]], torch.Tensor] = {}
        self._aux_streams: Dict[int, "torch.cuda.Stream"] = {}
        self._aux_pool: list = []
        self._aux_next = -1
        self.wgrad_batch = None      # ops.WgradBatch while a backward pass defers its weight gradients to ONE launch per geometry
        self.overlap_wgrad = False   # measured on MI355X: 4 streams (3.23 ms/step) lose to 2 (2.93 ms/step)
        self.wgrad_stream = None     # set by a model while a chain whose weight gradients should run elsewhere is issued

    def aux_stream(self) -> "torch.cuda.Stream":
        """A companion stream of the current stream for weight-gradient kernels.  The weight and
        bias gradients of a layer only feed the optimizer, so they run beside the data-gradient
        chain: the f32 MFMA of the weight-gradient kernels executes on the VALU pipeline while the
        bf16x3 data-gradient kernels occupy the matrix cores."""
        cur = torch.cuda.current_stream(self.device)
        if self.wgrad_stream is not None:       # weight gradients of the current chain are lent to another chain's stream
            return self.wgrad_stream
        if not self.overlap_wgrad:
            return cur
        n = int(self.overlap_wgrad) if not isinstance(self.overlap_wgrad, bool) else 1
        if n > 1:                               # a pool shared by every chain, handed out round-robin (VDVAE: the weight
            if len(self._aux_pool) < n:         # gradients of several Blocks in flight beside the data-gradient chains)
                self._aux_pool += [torch.cuda.Stream(device=self.device) for _ in range(n - len(self._aux_pool))]
            self._aux_next = (self._aux_next + 1) % n
            return self._aux_pool[self._aux_next]
        aux = self._aux_streams.get(cur.cuda_stream)
        if aux is None:
            aux = torch.cuda.Stream(device=self.device)
            self._aux_streams[cur.cuda_stream] = aux
        return aux

    def join_aux(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        aux = self._aux_streams.get(cur.cuda_stream)
        if aux is not None and self.overlap_wgrad:
            ops.wait_stream(cur, aux)

    def join_all_aux(self) -> None:
        """the current stream waits for every companion stream that carried weight gradients (models whose backward
        runs on several streams)"""
        cur = torch.cuda.current_stream(self.device)
        if self.overlap_wgrad:
            for aux in list(self._aux_streams.values()) + self._aux_pool:
                ops.wait_stream(cur, aux)

    def get(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        key = (name, tuple(int(s) for s in shape))
        buf = self._bufs.get(key)
        if buf is None:
            buf = torch.zeros(key[1], dtype=dtype, device=self.device)
            self._bufs[key] = buf
        return buf


class Module:
    """Base of the host-side mirrors of the reference's hk.Module classes."""

    def __init__(self, name: Optional[str] = None):
        self.name = name or type(self).__name__
        self.store: Optional[ParamStore] = None
        self.ws: Optional[Workspace] = None

    def attach(self, store: ParamStore, prefix: str) -> None:
        self.store, self.prefix = store, prefix

    def P(self, leaf: str) -> torch.Tensor:
        return self.store.p[f"{self.prefix}/{leaf}"]

    def G(self, leaf: str) -> torch.Tensor:
        return self.store.g[f"{self.prefix}/{leaf}"]

    def buf(self, name: str, shape) -> torch.Tensor:
        return self.ws.get(f"{self.prefix}/{name}", shape)

    def wgrad(self, *args, **kw) -> None:
        """ops.layer_wgrad on the companion stream of the current stream (ordered after everything
        enqueued so far on the current stream)."""
        kw.setdefault("bf16", self.store.use_bf16)
        batch = self.ws.wgrad_batch
        if batch is not None and len(args) == 5 and set(kw) <= {"bf16", "in_act"}:
            g, x, dy, dw, db = args
            if ops.WgradBatch.eligible(g, g._desc(x.shape[0], "wgrad")):     # joins the end-of-backward launch of its geometry
                batch.add(g, x, dy, dw, db, kw["bf16"], kw.get("in_act", ACT_NONE))
                return
        cur = torch.cuda.current_stream(self.ws.device)
        aux = self.ws.aux_stream()
        if aux is cur or aux == cur:
            ops.layer_wgrad(*args, **kw)
            return
        ops.wait_stream(aux, cur)
        with torch.cuda.stream(aux):
            ops.layer_wgrad(*args, **kw)
