"""Partial sums instead of f32 atomics for parameter gradients (include/pmhip.h: pm_wgrad_part, pm_reduce_partials).

jax.grad hands the reference ONE deterministic array per parameter (networks.py:30-36,62-68,116-129 under
train_pm_vae.py:58-72).  Here a weight-gradient launch splits its reduction over m-splits / persistent workgroups; each
split STORES its sums into its own slot of an arena owned by this object and `reduce()` adds the slots into the flat
gradient buffer in a fixed order (one launch, or fused into the optimizer's read of g): no f32 atomic reaches memory, the
gradients are run-to-run identical, and the flat buffer keeps the contract every consumer relies on (the optimizer kernels, the
data-parallel reducer, the tests): after reduce() it holds the gradient.

One `PartialSums` per ParamStore.  ops.gather_wgrad / ops.WgradBatch / ops.colsum find the store a gradient view belongs to by
address (`owner_of`), ask the library how many slots the launch will write (the launch plan is the library's), get the arena
from `arena()` and mark it pending; `reduce()` sums the pending arenas.  A recorded ops.LaunchPlan replays the same calls with
the same arenas and job tables.  PM_NO_PARTIALS=1: every launch keeps its atomics (A/B runs).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import ReduceJob

_OWNERS: "weakref.WeakSet[PartialSums]" = weakref.WeakSet()
ENABLED = not os.environ.get("PM_NO_PARTIALS")


def owner_of(t: Optional[torch.Tensor]) -> Optional["PartialSums"]:
    """the PartialSums whose flat gradient buffer contains `t` (None: a free-standing tensor, or partial sums are off)"""
    if t is None or not ENABLED:
        return None
    p = t.data_ptr()
    for o in _OWNERS:
        if o.base <= p < o.end:
            return o
    return None


class PartialSums:
    def __init__(self, flat_g: torch.Tensor):
        self.flat = flat_g
        self.base = flat_g.data_ptr()
        self.end = self.base + 4 * flat_g.numel()
        # (g_off, count) -> [arena tensor, nslots, stride, src offset inside the arena]; one entry per run of gradient elements
        self.entries: Dict[Tuple[int, int], list] = {}
        self.pending: Dict[Tuple[int, int], None] = {}       # insertion-ordered set of entries written since the last reduce
        self._tables: Dict[tuple, Tuple[torch.Tensor, int]] = {}
        self.arena_bytes = 0
        if flat_g.is_cuda:
            _OWNERS.add(self)

    def offset(self, t: torch.Tensor) -> int:
        off = (t.data_ptr() - self.base) // 4
        assert 0 <= off and off + t.numel() <= self.flat.numel(), "gradient view outside the flat buffer"
        return off

    # -- arenas -----------------------------------------------------------------------------------------------------------
    def arena(self, g_off: int, count: int, nslots: int, stride: Optional[int] = None, shared: Optional[tuple] = None):
        """Arena of the gradient run [g_off, g_off + count): `nslots` slots `stride` floats apart (default: count rounded up to
        4).  shared = (arena tensor, float offset): the run lives inside an arena another run allocated (the groups of a table
        launch share one slot-major arena).  Returns (arena tensor, float offset of the run's slot 0); marks the run pending."""
        from . import ops

        key = (int(g_off), int(count))
        stride = int(stride) if stride is not None else (count + 3) // 4 * 4
        if key in self.pending:
            # a second launch into the same run before its slots were added (a backward pass repeated without reading the
            # gradients in between): the first launch's sums must reach the flat buffer before they are overwritten
            if ops._recording is not None:
                raise RuntimeError(f"partial sums: two launches into gradient run {key} inside one recorded step")
            self.flush_sync()
        e = self.entries.get(key)
        if e is None:
            for (o, c) in self.entries:
                if o < g_off + count and g_off < o + c:
                    raise ValueError(f"partial sums: gradient run {key} overlaps the registered run {(o, c)}")
        if shared is not None:
            buf, src_off = shared
            if e is None or e[0] is not buf or e[1] != nslots or e[2] != stride or e[3] != src_off:
                self.entries[key] = [buf, int(nslots), stride, int(src_off)]
                self._tables.clear()
        elif e is None or e[1] != nslots or e[2] != stride or e[3] != 0:
            if e is not None and e[3] == 0:
                self.arena_bytes -= e[0].numel() * 4
            buf = torch.zeros(int(nslots) * stride, dtype=torch.float32, device=self.flat.device)
            self.arena_bytes += buf.numel() * 4
            self.entries[key] = [buf, int(nslots), stride, 0]
            self._tables.clear()
        self.pending[key] = None
        e = self.entries[key]
        return e[0], e[3]

    def new_shared(self, total_floats: int) -> torch.Tensor:
        buf = torch.zeros(int(total_floats), dtype=torch.float32, device=self.flat.device)
        self.arena_bytes += buf.numel() * 4
        return buf

    # -- reduction --------------------------------------------------------------------------------------------------------
    def _table(self, keys: tuple) -> Tuple[torch.Tensor, int]:
        cached = self._tables.get(keys)
        if cached is not None:
            return cached
        jobs: List[ReduceJob] = []
        for key in keys:
            buf, nslots, stride, src_off = self.entries[key]
            g_off, count = key
            per = 256 if nslots >= 8 else 1024
            base = buf.data_ptr() + 4 * src_off
            # whole 16-byte vectors when the run is aligned (the pad elements are zero in the arena and belong to nobody in
            # the flat buffer): the kernel then takes its vector path for the ragged tail too - the same summation order as
            # pm_adam_step_jobs, so the fused and the unfused optimizer give the same bits
            if g_off % 4 == 0 and stride % 4 == 0 and src_off % 4 == 0 and self.flat.numel() % 4 == 0:
                count = (count + 3) // 4 * 4
            for o in range(0, count, per):
                j = ReduceJob()
                j.src = base + 4 * o
                j.stride, j.g_off = stride, g_off + o
                j.count, j.nslots = min(per, count - o), nslots
                jobs.append(j)
        raw = b"".join(bytes(j) for j in jobs)
        dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.flat.device)
        self._tables[keys] = (dev, len(jobs))
        return self._tables[keys]

    def reduce(self, lo: int = 0, hi: Optional[int] = None, only: Optional[tuple] = None) -> None:
        """flat_g[i] += sum over slots, for every pending run that touches [lo, hi) (default: all; only: these runs), on the
        current stream.  The caller orders the stream behind the launches that wrote the arenas."""
        from . import ops

        hi = self.flat.numel() if hi is None else hi
        keys = tuple(sorted(k for k in (only if only is not None else self.pending)
                            if k in self.pending and k[0] < hi and lo < k[0] + k[1]))
        if not keys:
            return
        for k in keys:
            del self.pending[k]
        table, njobs = self._table(keys)
        nbytes = sum((self.entries[k][1] + 2) * k[1] * 4.0 for k in keys)
        ops.reduce_partials(table, njobs, self.flat, nbytes)

    def adam_table(self, ranges=None):
        """Job table of pm_adam_step_jobs: every pending run, plus runs without partial sums for the rest of the flat buffer.
        Returns (device table, njobs, arena bytes read) and takes the runs off the pending set; None when a run or the buffer
        is not made of whole 16-byte vectors (PM_NO_PARAM_PAD builds): the caller reduces and runs the plain optimizer.
        ranges: sorted, disjoint (lo, hi) element ranges (multiples of 4) - the table covers exactly their union and only the
        pending runs inside them are taken (an optimizer update of the parameters whose gradients are final, while the rest
        of the backward pass still runs); a pending run that straddles a range boundary -> None."""
        n = self.flat.numel()
        ranges = tuple((int(a), int(b)) for a, b in ranges) if ranges is not None else ((0, n),)
        if any(a % 4 or (b % 4 and b != n) or a >= b for a, b in ranges):
            return None
        inside = lambda k: any(a <= k[0] and k[0] + (k[1] + 3) // 4 * 4 <= b for a, b in ranges)     # noqa: E731
        touches = lambda k: any(k[0] < b and a < k[0] + k[1] for a, b in ranges)                   # noqa: E731
        keys = tuple(sorted(k for k in self.pending if touches(k)))
        if any(not inside(k) for k in keys):
            return None
        if n % 4 or any(k[0] % 4 or self.entries[k][2] % 4 or self.entries[k][3] % 4 for k in keys):
            return None
        cached = self._tables.get(("adam", ranges) + keys)
        if cached is None:
            jobs: List[ReduceJob] = []

            def plain(lo, hi):
                for o in range(lo, hi, 1024):
                    j = ReduceJob()
                    j.src, j.stride, j.g_off, j.count, j.nslots = None, 0, o, min(1024, hi - o), 0
                    jobs.append(j)

            nbytes = 0.0
            for lo, hi in ranges:
                pos = lo
                for key in keys:
                    g_off, count = key
                    if not (lo <= g_off < hi):
                        continue
                    buf, nslots, stride, src_off = self.entries[key]
                    plain(pos, g_off)
                    per = 256 if nslots >= 8 else 1024
                    base = buf.data_ptr() + 4 * src_off
                    cnt4 = (count + 3) // 4 * 4              # whole vectors: the pad elements are zero in every buffer
                    for o in range(0, cnt4, per):
                        j = ReduceJob()
                        j.src, j.stride, j.g_off = base + 4 * o, stride, g_off + o
                        j.count, j.nslots = min(per, cnt4 - o), nslots
                        jobs.append(j)
                    pos = g_off + cnt4
                    nbytes += nslots * count * 4.0
                plain(pos, hi)
            raw = b"".join(bytes(j) for j in jobs)
            dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.flat.device)
            cached = (dev, len(jobs), nbytes)
            self._tables[("adam", ranges) + keys] = cached
        for k in keys:
            del self.pending[k]
        return cached

    def flush_sync(self) -> None:
        """host code is about to read or overwrite the flat gradient buffer (tests, checkpoints): everything in flight ends,
        pending partial sums are added"""
        if self.pending:
            torch.cuda.synchronize(self.flat.device)
            self.reduce()
            torch.cuda.synchronize(self.flat.device)


def make_part(w_buf, w_off, w_stride, nslots, b_buf=None, b_off=0, b_stride=0, bg_buf=None, bg_off=0, bg_stride=0):
    from ._lib import WgradPart

    p = WgradPart()
    p.w = w_buf.data_ptr() + 4 * w_off
    p.b = (b_buf.data_ptr() + 4 * b_off) if b_buf is not None else None
    p.bg = (bg_buf.data_ptr() + 4 * bg_off) if bg_buf is not None else None
    p.w_stride, p.b_stride, p.bg_stride, p.nslots = int(w_stride), int(b_stride), int(bg_stride), int(nslots)
    return p
