"""Data parallelism for the PM-VAE step: one process per GPU, gradients summed with ONE
all-reduce of the flat gradient buffer per step (RCCL over xGMI on MI355X, `nccl` backend; `gloo`
on CPU for tests).  This is what bax.Trainer(num_devices=N) does with pmap + pmean in the
reference (train_pm_vdvae.py:146-154): batch sizes in the configs are PER DEVICE
(README.md:139-141), so scaling is weak and the loss is the mean of per-rank means.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_rows(global_batch: int, rank: int, world: int) -> slice:
    """Rank r owns rows [r*B/N, (r+1)*B/N) of a global batch."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the flat gradient buffer (the optimizer divides by world size)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_mean_scalars(values: torch.Tensor) -> torch.Tensor:
    """Mean over ranks of a few logging scalars (loss / aux), only at logging steps."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
        values /= dist.get_world_size()
    return values


class GradReducer:
    """Reverse-order gradient buckets, all-reduced while the rest of the backward pass still runs (SURVEY.md 8(e);
    what pmap's pmean does for the reference at train_pm_vdvae.py:146-154, one bucket at a time instead of after the fact).

    The flat gradient buffer is [ndim != 1 leaves in creation order | 1-D leaves].  A model's backward calls
    `ready(prefixes)` (or `ready_ranges` with element ranges of the flat buffer: ops.WgradBatch reports the weight
    gradients of every grouped launch that way) on the stream that carried the weight-gradient kernels as soon as those
    parameters have their final gradient.  Ranges join the pending set; once `bucket_bytes` are pending, every merged
    (contiguous) pending range of at least `min_issue_bytes` is all-reduced on the communication stream behind an event
    of the producing stream(s); smaller fragments (one layer of a Block whose neighbours are still to come) wait until
    their neighbours arrive or until finish().  `finish()` reduces what is left - in particular the whole 1-D suffix
    (biases: a few KB) in one call - and makes the current stream wait for every collective.  Sum only: the optimizer
    kernels divide by the world size (grad_scale), and the VDVAE's global-norm clip / non-finite skip run on the reduced
    buffer after finish(), so every rank takes the same decision.  Every rank runs the same host code, so every rank
    issues the same collectives in the same order.

    Every operation goes through ops.record_event / ops.wait_event / ops.host_call, so a recorded launch plan replays
    the same overlap.  While a HIP graph is being captured ready() does nothing (a collective issued during capture would
    run once at capture time and never again): a captured step reduces everything in finish(), between its graphs."""

    def __init__(self, store, bucket_bytes: int = 16 << 20, overlap: bool = True, async_issue: Optional[bool] = None,
                 min_issue_bytes: Optional[int] = None):
        """async_issue: collectives are enqueued without blocking the host (default for nccl = RCCL: the collective is a
        kernel on RCCL's stream, ordered behind the bucket's event).  gloo stages device tensors through pinned host
        memory on a two-thread pool; several collectives of one step in flight there were measured 60x slower than one
        (2 ranks sharing one MI355X: 402 vs 6.9 ms per step), so for gloo (CPU tests, the one-GPU rehearsal) each bucket
        is reduced synchronously at its ready point: same buckets, same order, same results, no overlap."""
        self.store, self.bucket_bytes, self.overlap = store, int(bucket_bytes), overlap
        self.min_issue_bytes = int(min_issue_bytes) if min_issue_bytes is not None else self.bucket_bytes // 4
        if async_issue is None:
            async_issue = dist.is_initialized() and dist.get_backend() == "nccl"
        self.async_issue = async_issue
        self.flat = store.flat_g           # (the tensor itself: ranges of it are all-reduced in place)
        self._cuda = torch.device(store.device).type == "cuda"     # host tensors (gloo, CPU tests): no streams to order
        self.comm = torch.cuda.Stream(device=store.device) if self._cuda else None
        self._done: List[Tuple[int, int]] = []          # element ranges already handed to a collective this step
        self._pending: List[Tuple[int, int]] = []
        self._pending_events: List["torch.cuda.Event"] = []
        self._works: List = []
        self.calls_last_step = 0
        self.calls_before_finish_last_step = 0          # collectives issued while the backward pass was still running
        self._calls = 0
        self._calls_early = 0

    # -- ranges ---------------------------------------------------------------------------------------------------
    def _weight_range(self, prefixes: Sequence[str]) -> Optional[Tuple[int, int]]:
        offs = [(o, o + n) for name, (o, n) in self.store.offsets.items()
                if o < self.store.n_decay and any(name == p or name.startswith(p + "/") for p in prefixes)]
        if not offs:
            return None
        lo, hi = min(a for a, _ in offs), max(b for _, b in offs)
        if sum(b - a for a, b in offs) != hi - lo:
            raise ValueError(f"parameters under {list(prefixes)} are not contiguous in the flat buffer")
        return lo, hi

    @staticmethod
    def _merge(ranges):
        out = []
        for a, b in sorted(ranges):
            if out and a <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], b))
            else:
                out.append((a, b))
        return out

    @staticmethod
    def _overlaps(r, ranges) -> bool:
        return any(a < r[1] and r[0] < b for a, b in ranges)

    # -- host-side pieces that a launch plan replays --------------------------------------------------------------------
    def _issue(self, a: int, b: int, early: bool = False) -> None:
        self._calls += 1
        self._calls_early += int(early)
        if not self.async_issue:
            if self._cuda:
                self.comm.synchronize()         # the bucket's producers (events the comm stream waits for) have run
            dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM)
            return
        if self._cuda:
            with torch.cuda.stream(self.comm):
                self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def _wait_all(self) -> None:
        for w in self._works:
            w.wait()                      # the current stream waits for the collective (nccl); gloo blocks the host
        self._works.clear()
        self.calls_last_step, self._calls = self._calls, 0
        self.calls_before_finish_last_step, self._calls_early = self._calls_early, 0

    def _flush(self, final: bool) -> None:
        from . import ops

        merged = self._merge(self._pending)
        go = [r for r in merged if final or (r[1] - r[0]) * 4 >= self.min_issue_bytes]
        if not go:
            return
        # the communication stream is in order: once it has waited for an event, every later collective is behind it too
        for ev in self._pending_events:
            ops.wait_event(self.comm, ev)
        self._pending_events = []
        for a, b in go:
            # the weight-gradient launches left per-split partial sums (partials.PartialSums): this range's are added into
            # the flat buffer on the communication stream, behind its producers and in front of its collective
            if self._cuda and getattr(self.store, "partials", None) is not None and self.store.partials.pending:
                with torch.cuda.stream(self.comm):
                    self.store.reduce_partials(a, b)
            ops.host_call(self._issue, a, b, not final)
            self._done.append((a, b))
        self._pending = [r for r in merged if r not in go]

    # -- model-facing API ------------------------------------------------------------------------------------------
    def ready(self, prefixes: Sequence[str], streams: Optional[Sequence["torch.cuda.Stream"]] = None) -> None:
        """every parameter under `prefixes` has its final gradient once the work enqueued so far on `streams` (default:
        the current stream) has run"""
        if not self.overlap:
            return
        r = self._weight_range(prefixes)
        if r is not None:
            self.ready_ranges([r], streams)

    def ready_ranges(self, ranges: Sequence[Tuple[int, int]], streams: Optional[Sequence["torch.cuda.Stream"]] = None) -> None:
        """the elements [a, b) of the flat gradient buffer, for every (a, b) of `ranges`, are final once the work enqueued
        so far on `streams` (default: the current stream) has run"""
        if not self.overlap or not ranges:
            return
        if self._cuda and torch.cuda.is_current_stream_capturing():
            return
        from . import ops

        for r in ranges:
            if self._overlaps(r, self._done) or self._overlaps(r, self._pending):
                raise ValueError(f"gradient range {r} was reported ready twice in one step")
            self._pending.append((int(r[0]), int(r[1])))
        self._pending = self._merge(self._pending)
        for s in ((streams or [torch.cuda.current_stream(self.store.device)]) if self._cuda else ()):
            ev = torch.cuda.Event()
            ops.record_event(ev, s)
            self._pending_events.append(ev)
        if sum(b - a for a, b in self._pending) * 4 >= self.bucket_bytes:
            self._flush(final=False)

    def finish(self) -> None:
        """reduces every range not yet handed over (the 1-D suffix included) and joins the collectives; call on the stream
        the optimizer runs on, after the backward pass has been joined onto it"""
        from . import ops

        if self._cuda:
            ev = torch.cuda.Event()
            ops.record_event(ev, torch.cuda.current_stream(self.store.device))
            self._pending_events.append(ev)
        covered = self._merge(self._done)
        pos, total = 0, self.flat.numel()
        rest = []
        for a, b in covered + [(total, total)]:
            if a > pos:
                rest.append((pos, a))
            pos = max(pos, b)
        self._pending = rest              # everything not reduced yet, pending fragments included, in as few calls as possible
        self._flush(final=True)
        self._done, self._pending, self._pending_events = [], [], []
        ops.host_call(self._wait_all)
