"""The per-step training loop of the PM-VAE as one launch sequence (optionally one HIP graph).

This is what bax.Trainer's jitted train_step is for train_pm_vae.py (reference :85-102 with the
loss_fn of :58-72 and the optimizer of :74-83): sample eps -> forward -> loss -> backward ->
[gradient all-reduce across ranks] -> Adam -> step += 1.  Every arithmetic op is a libpmhip.so
kernel; the step counter, beta and learning-rate schedules live on the device so the sequence
can be captured once and replayed.
"""
from __future__ import annotations

import os

from typing import Any, Dict, Mapping, Optional

import torch

from . import ops
from ._lib import LossCfg
from .models.vae import PosteriorMatchingVAE
from .optim import Chain


class _PlannedStep:
    """Host side of a train step: the first call runs the model classes eagerly (allocating every buffer), the
    second call runs them again while recording an ops.LaunchPlan, every later call replays the plan (same
    launches, same streams and cross-stream waits, ~2 us of host time per launch).  `use_plan = False` keeps the
    eager path; a KernelTimer (bench.py's profiling pass) also forces it."""

    use_plan = True

    # ---- double-buffered step inputs -------------------------------------------------------------------------------------
    # A recorded launch plan reads its inputs at fixed addresses, so a new batch has to be COPIED into them; on the step's own
    # stream those copies (two 6 us kernels plus their launch gaps and a cross-stream wait: ~40 us of a 1.33 ms PM-VAE step,
    # tools/pace_probe.py) sit between two steps with nothing else to run.  With two input sets and one recorded plan per set,
    # set_batch() fills the set the NEXT step reads on a feed stream while the current step is still running.
    # MEASURED (same box, three pairs, profiles/r04_ab_tail_head.txt): 1.349 / 1.354 / 1.370 ms with the feed stream against
    # 1.335 / 1.335 / 1.340 ms with the copies on the step's own stream - a third stream's copies disturb the two queues' balance
    # more than their 40 us are worth, like every third stream tried on this step: OFF unless PM_FEED_OVERLAP=1.
    def _init_inputs(self, dev, shapes, double: bool) -> None:
        n = 2 if (double and os.environ.get("PM_FEED_OVERLAP")) else 1
        self._in = [{k: torch.zeros(v, device=dev) for k, v in shapes.items()} for _ in range(n)]
        self._cur = 0
        self._plans, self._warms = [None] * n, [False] * n
        self._feed = torch.cuda.Stream(device=dev) if n > 1 else None
        self._ready = [None] * n                 # event behind the copies into set i (None: nothing fed since its last step)
        self._free = [None] * n                  # event behind the last step that read set i

    def _input_property(name):          # noqa: N805 - steps without input sets (single static buffers) keep plain attributes
        def get(self):
            sets = self.__dict__.get("_in")
            return sets[self._cur][name] if sets is not None and name in sets[0] else self.__dict__["_" + name]

        def put(self, value):
            self.__dict__["_" + name] = value

        return property(get, put)

    x, b, eps = _input_property("x"), _input_property("b"), _input_property("eps")

    def _feed_inputs(self, **srcs) -> None:
        """copies the given tensors into the input set the next step() will read"""
        dev = self._in[0]["x"].device
        if self._feed is None:
            self.stream.wait_stream(torch.cuda.current_stream(dev))          # producers of the batch
            with torch.cuda.stream(self.stream):
                for k, t in srcs.items():
                    if t is not None:
                        self._in[0][k].copy_(t.reshape(self._in[0][k].shape), non_blocking=True)
            return
        self._cur ^= 1
        cur = self._cur
        self._feed.wait_stream(torch.cuda.current_stream(dev))
        if self._free[cur] is not None:
            self._feed.wait_event(self._free[cur])                          # the step that last read this set has finished
        with torch.cuda.stream(self._feed):
            for k, t in srcs.items():
                if t is not None:
                    self._in[cur][k].copy_(t.reshape(self._in[cur][k].shape), non_blocking=True)
        if self._ready[cur] is None:
            self._ready[cur] = torch.cuda.Event()
        self._ready[cur].record(self._feed)

    def _inputs_acquire(self) -> None:
        """step(): the step's stream waits for the copies into the current set"""
        if self._feed is not None and self._ready[self._cur] is not None:
            self.stream.wait_event(self._ready[self._cur])

    def _inputs_release(self) -> None:
        if self._feed is not None:
            if self._free[self._cur] is None:
                self._free[self._cur] = torch.cuda.Event()
            self._free[self._cur].record(self.stream)

    @property
    def _plan(self):
        return self._plans[self._cur] if getattr(self, "_plans", None) else getattr(self, "_plan1", None)

    @_plan.setter
    def _plan(self, v):
        if getattr(self, "_plans", None):
            self._plans[self._cur] = v
        else:
            self._plan1 = v

    @property
    def _warm(self):
        return self._warms[self._cur] if getattr(self, "_warms", None) else getattr(self, "_warm1", False)

    @_warm.setter
    def _warm(self, v):
        if getattr(self, "_warms", None):
            self._warms[self._cur] = v
        else:
            self._warm1 = v

    def _planned(self, sequence) -> None:
        if ops._timer is not None:      # per-kernel timing (ops.KernelTimer): always the eager path, also when a plan exists
            sequence()
            return
        if self._plan is not None:
            store = self._grad_store()
            if store is None or store.g_clean or not self.adam_cfg.zero_grad:
                self._plan.replay()
                return
            sequence()                  # host code wrote gradients since the last step: this step zero-fills first (the
            return                      # recorded plan relies on the optimizer kernel having zeroed the buffer)
        if self.use_plan and self._warm and ops._timer is None:
            ops.begin_recording()
            try:
                sequence()
            finally:
                plan = ops.end_recording()
            self._plan = plan
        else:
            sequence()
            self._warm = True

    def invalidate_plan(self) -> None:
        if getattr(self, "_plans", None):
            self._plans, self._warms = [None] * len(self._plans), [False] * len(self._warms)
        else:
            self._plan, self._warm = None, False

    def _grad_store(self):
        return getattr(self, "store", None) or getattr(getattr(self, "model", None), "store", None)

    # The optimizer kernel zeroes the gradient buffer it has just consumed (pm_adam_cfg.zero_grad): the zero-fill launch in
    # front of the next backward pass (its weight-gradient kernels accumulate with atomics) is only needed when something
    # else ran a backward pass in between.  `adam_cfg.zero_grad = 0` restores the separate launch (tests that read the
    # gradient buffer after step()).
    def _zero_grad(self, store) -> None:
        if not store.g_clean:
            store.zero_grad()           # also drops partial sums a host-side backward pass left pending
        store.g_clean = False

    plain_adam = True       # False: the step's optimizer reads the REDUCED buffer (global-norm clip: VDVAETrainStep)

    def _reduce_partials(self, store) -> None:
        """The weight-gradient launches of the backward pass left per-split partial sums (partials.PartialSums): add them
        into the flat gradient buffer on the current stream - the stream the optimizer runs on, already ordered behind every
        weight-gradient launch.  With an overlapping data-parallel reducer each bucket's runs are added on the communication
        stream right in front of its all-reduce instead (GradReducer._flush)."""
        r = store.reducer
        if r is None and self._fuse_adam() and store.partials is not None:
            return                      # one GPU, plain Adam: _adam_step reads the partial sums itself (pm_adam_step_jobs)
        if r is None or not r.overlap:
            store.reduce_partials()

    fused_adam = not os.environ.get("PM_NO_FUSED_ADAM")       # A/B switch

    def _fuse_adam(self) -> bool:
        # zero_grad = 0 (tests that read the gradient buffer after step()): the buffer must hold the gradient -> reduce first
        return bool(self.fused_adam and self.plain_adam and self.adam_cfg is not None and self.adam_cfg.zero_grad)

    def _adam_step(self, s, step_dev) -> None:
        """optax.scale_by_adam -> add_decayed_weights -> schedule -> apply_updates on the flat buffers.  With partial sums
        pending (one GPU) the optimizer kernel adds them as it reads g; otherwise g already holds the gradient.  Ranges an
        early update of this step already covered (_early_adam) are left out."""
        ps = s.partials
        early = getattr(self, "_early_done", None) or []
        self._early_done = []
        if early:                                   # the complement of the ranges updated during the backward pass
            rest, pos = [], 0
            for a, b in sorted(early):
                if a > pos:
                    rest.append((pos, a))
                pos = b
            if pos < s.flat_p.numel():
                rest.append((pos, s.flat_p.numel()))
            tab = ps.adam_table(rest)
            if tab is None:
                raise RuntimeError("early optimizer ranges left a gradient run straddling their boundary")
            table, njobs, nbytes = tab
            ops.adam_step_jobs(table, njobs, s.flat_p, s._flat_g, s.flat_m, s.flat_v, s.n_decay, step_dev, self.adam_cfg, nbytes)
            return
        if ps is not None and ps.pending and s.reducer is None and self._fuse_adam():
            tab = ps.adam_table()
            if tab is not None:
                table, njobs, nbytes = tab
                ops.adam_step_jobs(table, njobs, s.flat_p, s._flat_g, s.flat_m, s.flat_v, s.n_decay, step_dev, self.adam_cfg,
                                   nbytes)
                return
            s.reduce_partials()
        ops.adam_step(s.flat_p, s.flat_g, s.flat_m, s.flat_v, s.n_decay, step_dev, self.adam_cfg)

    early_adam = not os.environ.get("PM_NO_EARLY_ADAM")          # A/B switch

    def _early_adam(self, s, prefixes, step_dev) -> bool:
        """A model's backward pass reports (on the stream those launches ran on) that the WEIGHTS under `prefixes` have their
        final gradients and are not read again this step: their optimizer update - HBM-bound, 8 passes over the range - runs
        now, on the current stream, beside the rest of the backward pass instead of after it.  One GPU, plain Adam, fused
        partial sums only; the update at the end of the step covers the complement.  Same arithmetic per element, so the
        step's results do not change."""
        ps = s.partials
        if not (self.early_adam and ps is not None and s.reducer is None and self._fuse_adam()) or self.adam_cfg is None:
            return False
        offs = [(o, o + n) for name, (o, n) in s.offsets.items()
                if o < s.n_decay and any(name == p or name.startswith(p + "/") for p in prefixes)]
        if not offs:
            return False
        lo, hi = min(a for a, _ in offs), max(b for _, b in offs)
        hi = (hi + 3) // 4 * 4
        if sum((b - a + 3) // 4 * 4 for a, b in offs) != hi - lo or lo % 4:
            return False                            # not one contiguous run of the flat buffer
        tab = ps.adam_table([(lo, hi)])
        if tab is None:
            return False
        table, njobs, nbytes = tab
        ops.adam_step_jobs(table, njobs, s.flat_p, s._flat_g, s.flat_m, s.flat_v, s.n_decay, step_dev, self.adam_cfg, nbytes)
        self._early_done = (getattr(self, "_early_done", None) or []) + [(lo, hi)]
        return True

    def _fresh_grads_before_graph(self, store) -> None:
        """A captured step was recorded while the gradient buffer was known to be zero (no zero-fill inside the graph): if host
        code has written gradients since (a direct model.backward(), a test poking store.g), zero it before the replay."""
        if self.adam_cfg.zero_grad and not store.g_clean:
            store.zero_grad()
            store.g_clean = True

    def _grads_consumed(self, store) -> None:
        store.g_clean = bool(self.adam_cfg.zero_grad)


def _step_stream(dev) -> "torch.cuda.Stream":
    """the stream a train step's main (critical) chain runs on: high priority unless PM_MAIN_PRIO=0"""
    return torch.cuda.Stream(device=dev, priority=int(os.environ.get("PM_MAIN_PRIO", "-1")))


def _make_reducer(store, world_size: int, overlap: bool):
    """data-parallel gradient reduction (parallel.GradReducer): buckets of a quarter of the buffer, 1-16 MB"""
    import torch.distributed as dist

    # PM_FORCE_DP=1 with an initialised process group: the reducer also runs on one rank (a 1-rank RCCL communicator on a single
    # GPU rehearses the nccl stream semantics of the N > 1 path: asynchronous buckets on the communication stream, collectives
    # inside a replayed launch plan)
    forced = os.environ.get("PM_FORCE_DP") == "1" and dist.is_available() and dist.is_initialized()
    if world_size <= 1 and not forced:
        store.reducer = None
        return None
    from .parallel import GradReducer

    total = store.flat_g.numel() * 4
    mb = os.environ.get("PM_BUCKET_MB")                       # A/B knob for measurements
    # a quarter of the buffer, 1 - 32 MB: PM-VAE (8.4 MB) one bucket behind the encoder backward; PM-VQVAE 140 MB / CelebA 273 MB
    # 5 - 9 collectives of 32 MB - each far above the size at which a ring over xGMI (7 links x ~153 GB/s per GPU, per-link
    # bound) stops being latency-bound, few enough that their fixed cost (tens of us each) stays under 1 % of the 8 - 10 ms step
    bucket = int(float(mb) * (1 << 20)) if mb else max(1 << 20, min(32 << 20, total // 4))
    store.reducer = GradReducer(store, bucket_bytes=bucket, overlap=overlap)
    return store.reducer


def loss_cfg_from_config(config: Mapping[str, Any], batch_size: int) -> LossCfg:
    """get_beta_schedule + the loss weights of loss_fn (train_pm_vae.py:28-43,62-70)."""
    c = LossCfg()
    beta = config.get("beta", {}) or {}
    c.beta_kind, c.low, c.high, c.period_or_steps, c.delay_or_begin = 0, 0.0, 1.0, 1, 0
    if "schedule" in beta:
        if beta["schedule"] == "monotonic":
            c.beta_kind = 1
            c.low, c.high = beta["low_value"], beta["high_value"]
            c.period_or_steps, c.delay_or_begin = beta["transition_steps"], beta.get("transition_begin", 0)
        elif beta["schedule"] == "cyclic":
            c.beta_kind = 2
            c.low, c.high = beta["low_value"], beta["high_value"]
            c.period_or_steps, c.delay_or_begin = beta["period"], beta.get("delay", 0)
        else:
            raise KeyError(beta["schedule"])
    c.matching_coef = config.get("matching_coef", 1.0)
    c.grad_scale = 1.0 / batch_size
    return c


class PMVAETrainStep(_PlannedStep):
    """Fused train step over static device buffers (x, b, eps are copied/generated in place)."""

    def __init__(self, model: PosteriorMatchingVAE, config: Mapping[str, Any], optimizer: Chain, batch_size: int,
                 x_shape, seed: int = 0, world_size: int = 1, rank: int = 0, use_graph: bool = False,
                 external_eps: bool = False, use_plan: bool = True, overlap_allreduce: bool = True):
        """use_graph=False (default): eager launches, the ELBO and posterior-matching chains overlap on
        two HIP streams.  use_graph=True: one HIP graph replay per step; ROCm 7.2 serialises the
        branches of a captured graph, so this form runs the two chains back to back (measured:
        65 k vs 80 k img/s) but costs no host time per kernel."""
        if model.store is None:
            model.init(x_shape)
        dev = model.store.device
        self.model, self.opt, self.B = model, optimizer, batch_size
        self.world_size, self.rank, self.seed = world_size, rank, seed
        self.loss_cfg = loss_cfg_from_config(config, batch_size)
        self.adam_cfg = optimizer.adam_cfg(grad_scale=1.0 / world_size)
        self.adam_cfg.zero_grad = 1
        # a captured step reduces between its two graphs: nothing may be issued from inside the capture (GradReducer.ready
        # is a no-op while capturing as well)
        self.reducer = _make_reducer(model.store, world_size, overlap_allreduce and not use_graph)
        x_shape = tuple(x_shape)
        b_shape = x_shape[:-1] + (1,) if len(x_shape) == 3 else x_shape
        self._init_inputs(dev, {"x": (batch_size,) + x_shape, "b": (batch_size,) + b_shape,
                                "eps": (batch_size, model.latent_dim)}, double=model.concurrent and not use_graph)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.metrics = torch.zeros(8, device=dev)       # loss, rec, kl, mll, beta
        self.g_rec = torch.zeros(batch_size, device=dev)
        self.g_kl = torch.zeros(batch_size, device=dev)
        self.g_mll = torch.zeros(batch_size, device=dev)
        self.external_eps = external_eps
        self.use_graph = use_graph
        self.use_plan = use_plan and not use_graph
        # hk.dropout(hk.next_rng_key(), ...) inside ResidualMLP (networks.py:125): device Philox keyed by (seed, rank),
        # the training step counter and one stream id per (network, block)
        from .models.networks import ResidualMLP

        for i, net in enumerate((model.encoder_net, model.decoder_net, model.partial_encoder_net)):
            if isinstance(net, ResidualMLP):
                net.dropout_seed, net.dropout_step_dev = seed + 7919 * rank, self.step_dev
                net.dropout_stream_base = 2000 + 64 * i
        if use_graph:
            model.concurrent = False
        # HIP graph capture is not allowed on the NULL stream: the step owns a side stream
        # The chain on this stream (encoder -> z -> decoder -> their backward passes) bounds the step; the side stream's
        # chain (partial encoder, AR-GMM, lent weight gradients) has slack.  High HIP stream priority for the critical chain:
        # when both queues have a kernel ready the dispatcher serves this one first (+2.8 % measured; PM_MAIN_PRIO=0 for A/B)
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        self._graph_fb: Optional[ops.Graph] = None
        self._graph_opt: Optional[ops.Graph] = None

    # -- pieces -----------------------------------------------------------------------------
    def _forward_backward(self) -> None:
        m = self.model
        if not self.external_eps:
            ops.normal_fill(self.eps, self.seed, self.step_dev, stream_id=self.rank)
        self._zero_grad(m.store)
        # d loss / d matching_ll = -matching_coef / B does not depend on the forward pass: with it on the device up front, the
        # posterior-matching branch runs its backward pass right behind its forward pass on the side stream (beside the
        # decoder) instead of waiting for the loss (PM_NO_EARLY_PM=1: the round-2 order, for A/B runs)
        early = m.concurrent and not self.use_graph and not os.environ.get("PM_NO_EARLY_PM")
        # round 4: ALL three upstream gradients are constants of the step (-1 / B, beta(step) / B, -matching_coef / B): with
        # them on the device up front the Bernoulli head writes d loss / d logits from its forward launch and the loss
        # kernel (metrics only then) leaves the dependent chain - it runs on the side stream (PM_NO_EARLY_LOSS=1: A/B)
        early_all = early and not os.environ.get("PM_NO_EARLY_LOSS")
        if early_all:
            ops.pmvae_loss_grads(self.B, self.loss_cfg, self.step_dev, self.g_rec, self.g_kl, self.g_mll)
        elif early:
            ops.pmvae_loss_grads(self.B, self.loss_cfg, self.step_dev, None, None, self.g_mll)
        self._wait_split()              # the bf16 weight copies of the previous update (side stream, see _update)
        out = m(self.x, self.b, is_training=True, eps=self.eps, early_g_mll=self.g_mll if early else None,
                early_g_rec=self.g_rec if early_all else None)
        if early_all:
            main, side = torch.cuda.current_stream(self.x.device), m._side_stream(self.x.device)
            ops.wait_stream(side, main)                 # reconstruction_ll and kl are ready behind the main stream's last launch
            with torch.cuda.stream(side):
                ops.pmvae_loss(out["reconstruction_ll"], out["kl"], out["matching_ll"], self.loss_cfg, self.step_dev,
                               self.metrics, None, None, None)
        else:
            ops.pmvae_loss(out["reconstruction_ll"], out["kl"], out["matching_ll"], self.loss_cfg, self.step_dev,
                           self.metrics, self.g_rec, self.g_kl, None if early else self.g_mll)
        m.backward(self.g_rec, self.g_kl, self.g_mll)
        self._reduce_partials(m.store)

    def _wait_split(self) -> None:
        ev = getattr(self, "_split_done", None)
        if ev is not None and self._split_on_side:
            ops.wait_event(torch.cuda.current_stream(self.x.device), ev)

    def _update(self) -> None:
        s = self.model.store
        m = self.model
        self._split_on_side = bool(m.concurrent and not self.use_graph and not os.environ.get("PM_SPLIT_MAIN"))
        if self._split_on_side and os.environ.get("PM_ADAM_SIDE"):
            # Two-stream steps: the optimizer (the one kernel that runs ALONE on the chip: 57 us for 266 MB) and the refresh
            # of the pre-split bf16 weight copies go to the SIDE stream; the main stream advances the step counter (the
            # optimizer reads a snapshot of it), takes the next batch's copies and runs the next step's head kernels (noise,
            # upstream gradients) beside them, and waits for `_split_done` in front of the first layer (_wait_split).
            # MEASURED: 1.349 / 1.340 / 1.342 ms against 1.346 / 1.345 / 1.339 ms with the optimizer on the main stream - no
            # effect, so the simpler order stays the default (PM_ADAM_SIDE=1 enables this one; PM_SPLIT_MAIN=1: the weight split
            # on the main stream too)
            main, side = torch.cuda.current_stream(self.x.device), m._side_stream(self.x.device)
            if getattr(self, "_split_done", None) is None:
                self._split_done = torch.cuda.Event()
                self._opt_step = torch.zeros(1, dtype=torch.int32, device=self.x.device)
            ops.counter_snapshot_increment(self.step_dev, self._opt_step)
            ops.wait_stream(side, main)
            with torch.cuda.stream(side):
                self._adam_step(s, self._opt_step)
                s.split_all()
                ops.record_event(self._split_done, side)
            self._grads_consumed(s)
            return
        self._adam_step(s, self.step_dev)
        self._grads_consumed(s)
        # refresh the pre-split bf16 weight copies (one launch): on the side stream beside the counter increment, the next
        # batch's copies and the next step's head kernels
        if self._split_on_side:
            main, side = torch.cuda.current_stream(self.x.device), m._side_stream(self.x.device)
            if getattr(self, "_split_done", None) is None:
                self._split_done = torch.cuda.Event()
            ops.wait_stream(side, main)
            with torch.cuda.stream(side):
                s.split_all()
                ops.record_event(self._split_done, side)
        else:
            s.split_all()
        ops.counter_increment(self.step_dev)

    def _allreduce(self) -> None:
        self.reducer.finish()           # buckets issued during the backward pass + the rest; Adam divides by world_size

    def _eager_sequence(self) -> None:
        self._forward_backward()
        if self.reducer is not None:
            self._allreduce()
        self._update()

    # -- one optimizer step on whatever is in self.x / self.b (/ self.eps) ---------------------
    def step(self) -> None:
        self._inputs_acquire()
        with torch.cuda.stream(self.stream):
            self._step()
        self._inputs_release()

    def _step(self) -> None:
        if not self.use_graph:
            self._planned(self._eager_sequence)
            return
        if self._graph_fb is None:
            # run once eagerly so that every workspace buffer exists before capture
            self._forward_backward()
            if self.world_size > 1:
                self._allreduce()
            self._update()
            self.stream.synchronize()
            self._graph_fb, self._graph_opt = ops.Graph(), ops.Graph()
            if self.world_size > 1:
                with self._graph_fb:
                    self._forward_backward()
                with self._graph_opt:
                    self._update()
            else:
                with self._graph_fb:
                    self._forward_backward()
                    self._update()
            return
        self._fresh_grads_before_graph(self.model.store)
        self._graph_fb.launch()
        if self.world_size > 1:
            self._allreduce()
            self._graph_opt.launch()

    def set_batch(self, x: torch.Tensor, b: torch.Tensor, eps: Optional[torch.Tensor] = None) -> None:
        self._feed_inputs(x=x, b=b, eps=eps)

    def synchronize(self) -> None:
        self.stream.synchronize()
        side = getattr(self.model, "_side", None)       # the weight copies / the loss metrics may still be on the side stream
        if side is not None:
            side.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.synchronize()
        v = self.metrics.cpu().tolist()
        return {"loss": v[0], "reconstruction_ll": v[1], "kl": v[2], "matching_ll": v[3], "beta": v[4]}

    def evaluate(self, x, b, eps) -> Dict[str, float]:
        """loss_fn with is_training=False (validation, bax semantics): forward + loss only."""
        m = self.model
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            self._wait_split()
            out = m(x, b, is_training=False, eps=eps)
            cfg = LossCfg.from_buffer_copy(self.loss_cfg)
            cfg.grad_scale = 1.0 / x.shape[0]
            metrics = torch.zeros(8, device=x.device)
            ops.pmvae_loss(out["reconstruction_ll"], out["kl"], out["matching_ll"], cfg, self.step_dev, metrics, None,
                           None, None)
        self.stream.synchronize()
        v = metrics.cpu().tolist()
        return {"loss": v[0], "reconstruction_ll": v[1], "kl": v[2], "matching_ll": v[3], "beta": v[4]}


class VQVAETrainStep(_PlannedStep):
    """train_vqvae.py:67-111 as one launch sequence: VQVAE forward (EMA codebook update inside, as in
    haiku) -> loss -> backward -> [gradient all-reduce] -> Adam -> step += 1."""

    def __init__(self, model, optimizer: Chain, batch_size: int, x_shape, world_size: int = 1, rank: int = 0,
                 use_graph: bool = False, use_plan: bool = True, overlap_allreduce: bool = True):
        if model.store is None:
            model.init(x_shape)
        dev = model.store.device
        self.model, self.opt, self.B = model, optimizer, batch_size
        self.world_size, self.rank = world_size, rank
        self.adam_cfg = optimizer.adam_cfg(grad_scale=1.0 / world_size)
        self.adam_cfg.zero_grad = 1
        self.reducer = _make_reducer(model.store, world_size, overlap_allreduce)
        self.x = torch.zeros((batch_size,) + tuple(x_shape), device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.use_graph = use_graph and world_size == 1
        self.use_plan = use_plan and not self.use_graph
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        self._graph: Optional[ops.Graph] = None

    def _sequence(self) -> None:
        m, s = self.model, self.model.store
        m(self.x, is_training=True)
        self._zero_grad(s)
        m.backward()
        self._reduce_partials(s)
        if self.reducer is not None:
            self.reducer.finish()
        self._adam_step(s, self.step_dev)
        self._grads_consumed(s)
        s.split_all()
        ops.counter_increment(self.step_dev)

    def step(self) -> None:
        with torch.cuda.stream(self.stream):
            if not self.use_graph:
                self._planned(self._sequence)
            elif self._graph is None:
                self._sequence()                    # eager once: every workspace buffer exists before capture
                self.stream.synchronize()
                self._graph = ops.Graph()
                with self._graph:
                    self._sequence()
            else:
                self._fresh_grads_before_graph(self.model.store)
                self._graph.launch()

    def set_batch(self, x: torch.Tensor) -> None:
        self.stream.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(self.stream):
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.stream.synchronize()
        v = self.model.metrics.cpu().tolist()
        return {"loss": v[0], "reconstruction_loss": v[1], "vq_loss": v[2], "perplexity": v[3]}

    def evaluate(self, x: torch.Tensor) -> Dict[str, float]:
        """loss_fn with is_training=False: no EMA update, no gradients."""
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            self.model(x, is_training=False)
        return self.read_metrics()


class PMVQVAETrainStep(_PlannedStep):
    """train_pm_vqvae.py:81-123 as one launch sequence: frozen VQ-VAE encode (is_training=False) ->
    code indices; partial encoder on [x*b | b] -> conditional vector; loss = -mean PixelCNN.log_prob;
    backward through the PixelCNN and the partial encoder only (trainable_predicate: every module
    not under "vqvae/"), Adam with the exponential-decay schedule, step += 1."""

    def __init__(self, vqvae, partial_encoder, pixel_cnn, optimizer: Chain, batch_size: int, x_shape, seed: int = 0,
                 world_size: int = 1, rank: int = 0, external_dropout: bool = False, overlap_allreduce: bool = True):
        from .models.core import ParamStore, Workspace

        if vqvae.store is None:
            vqvae.init(x_shape)
        dev = vqvae.store.device
        self.vqvae, self.penc, self.pcnn = vqvae, partial_encoder, pixel_cnn
        self.B, self.world_size, self.rank, self.seed = batch_size, world_size, rank, seed
        if getattr(pixel_cnn, "store", None) is None:          # not built yet (models.vqvae.build_partial_posterior)
            store, ws = ParamStore(), Workspace(dev)
            partial_encoder.ws = pixel_cnn.ws = ws
            xb_shape = tuple(x_shape[:-1]) + (x_shape[-1] + 1,)     # [x*b | b], b [B,H,W,1] (train_pm_vqvae.py:87-89)
            (cond_dim,) = partial_encoder.build(store, "partial_encoder", xb_shape)
            pixel_cnn.build(store, "pixel_cnn", cond_dim)
            store.allocate(dev, seed)
        self.store, self.ws = pixel_cnn.store, pixel_cnn.ws
        # optimizer None: forward / evaluation only (trainer.PMVQVAELoss called as a function)
        self.adam_cfg = optimizer.adam_cfg(grad_scale=1.0 / world_size) if optimizer is not None else None
        if self.adam_cfg is not None:
            self.adam_cfg.zero_grad = 1
        self.reducer = _make_reducer(self.store, world_size, overlap_allreduce) if optimizer is not None else None
        x_shape = tuple(x_shape)
        self.x = torch.zeros((batch_size,) + x_shape, device=dev)
        self.b = torch.zeros((batch_size,) + x_shape[:-1] + (1,), device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.metrics = torch.zeros(8, device=dev)
        self.g_ll = torch.zeros(batch_size, device=dev)
        self.dropout_masks = None          # parity tests set explicit masks (external_dropout)
        self.external_dropout = external_dropout
        self.use_plan = not external_dropout   # explicit masks are new tensors every step: nothing static to replay
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))

    @property
    def num_trainable_params(self) -> int:
        return self.store.num_params

    def forward(self, is_training: bool) -> torch.Tensor:
        # frozen VQ-VAE (no EMA update, no gradient): only its code indices are read, and they do not depend on the partial
        # encoder - the two run side by side (the frozen encoder on the PixelCNN's second stream)
        xob = self.ws.get("x_o_b", tuple(self.x.shape[:-1]) + (self.x.shape[-1] + self.b.shape[-1],))
        side = None if os.environ.get("PM_PX_SERIAL_HEAD") else self.pcnn.side_stream(self.x.device)
        if side is None:
            idx = self.vqvae.encoding_indices(self.x)
        else:
            main = torch.cuda.current_stream(self.x.device)
            ops.wait_stream(side, main)
            with torch.cuda.stream(side):
                idx = self.vqvae.encoding_indices(self.x)
        ops.mask_concat(self.x, self.b, xob)
        cond = self.penc(xob, is_training=is_training)
        if side is not None:
            ops.wait_stream(main, side)
        masks = self.dropout_masks if (self.external_dropout and is_training) else None
        ll = self.pcnn.log_prob(idx, training=is_training, conditional_input=cond, dropout_masks=masks,
                                seed=self.seed + self.rank, step_dev=self.step_dev)
        ops.neg_mean_loss(ll, 1.0 / self.B, self.metrics, self.g_ll if is_training else None)
        self._idx = idx
        return ll

    def _sequence(self) -> None:
        self._forward_backward()
        self._update()

    def _forward_backward(self) -> None:
        """forward + loss + the whole backward pass as the train step runs it (grouped weight gradients, two chains): the
        flat gradient buffer holds d loss / d trainable parameters afterwards"""
        s = self.store
        self.forward(True)
        self._zero_grad(s)
        # the 4 x num_resnet gated blocks repeat a handful of layer shapes: their weight gradients (and the partial encoder's)
        # are collected and launched at the end, one table-driven grouped launch per geometry (ops.WgradBatch)
        batched = s.use_bf16 and not os.environ.get("PM_NO_WGRAD_BATCH")
        if batched:
            if getattr(self, "_wgrad_batch", None) is None:
                self._wgrad_batch = ops.WgradBatch()
            self._wgrad_batch.reducer = self.reducer
            self.ws.wgrad_batch = self._wgrad_batch
        try:
            # the PixelCNN's grouped weight gradients (3 ms of chip-filling launches at the mnist size) start on the main stream
            # as soon as its data gradients are done; the latency-bound rest of the pass (embedding scatter, conditional
            # projection, the partial encoder's chain) runs beside them on the second stream
            tail = batched and not self.reducer and not os.environ.get("PM_PX_SERIAL_TAIL")
            # PM_PX_EARLY_ADAM=1 (one GPU): the optimizer update of the up pass's weights runs on a third stream beside the down
            # pass's backward chains (_early_adam).  MEASURED, same box, two pairs: pm_vqvae_celeb_a 7.39 vs 7.10 ms,
            # pm_vqvae_mnist 9.61 vs 9.44 ms WITHOUT it - a third stream of HBM-bound work slows the two latency-bound chains more
            # than the shorter tail returns (the same verdict as every third-stream experiment of rounds 3 and 4): off by default
            self.pcnn.early_update = ((lambda prefixes: self._early_adam(s, prefixes, self.step_dev))
                                      if (batched and not self.reducer and self.adam_cfg is not None and self.early_adam
                                          and os.environ.get("PM_PX_EARLY_ADAM", "0") == "1") else None)
            dcond = self.pcnn.backward(self.g_ll, overlap_tail=tail)
            ts = self.pcnn.tail_stream if tail else None
            if ts is not None:
                self.ws.wgrad_stream = ts               # a weight gradient that cannot join a grouped launch stays on `ts`
                try:
                    with torch.cuda.stream(ts):
                        self.penc.backward(dcond)
                finally:
                    self.ws.wgrad_stream = None
                ops.wait_stream(torch.cuda.current_stream(self.x.device), ts)
            else:
                self.penc.backward(dcond)
            if batched:
                self.ws.wgrad_batch.flush()
        finally:
            if batched:                     # never leave the shared workspace in deferred mode (evaluation paths share it)
                self._wgrad_batch.discard()
                self.ws.wgrad_batch = None
        self.ws.join_aux()

    def _update(self) -> None:
        s = self.store
        if self.adam_cfg is None:
            raise RuntimeError("this PMVQVAETrainStep was built without an optimizer (evaluation only)")
        self._reduce_partials(s)
        if self.reducer is not None:
            self.reducer.finish()
        self._adam_step(s, self.step_dev)
        self._grads_consumed(s)
        s.split_all()
        ops.counter_increment(self.step_dev)

    def step(self) -> None:
        with torch.cuda.stream(self.stream):
            self._planned(self._sequence)

    def set_batch(self, x: torch.Tensor, b: torch.Tensor) -> None:
        self.stream.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(self.stream):
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
            self.b.copy_(b.reshape(self.b.shape), non_blocking=True)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.stream.synchronize()
        return {"loss": self.metrics[0].item()}

    def evaluate(self, x: torch.Tensor, b: torch.Tensor) -> Dict[str, float]:
        self.set_batch(x, b)
        with torch.cuda.stream(self.stream):
            self.forward(False)
        return self.read_metrics()


class VDVAETrainStep(_PlannedStep):
    """train_pm_vdvae.py:109-154 as one launch sequence: eps -> PosteriorMatchingVDVAE forward ->
    loss = -mean(rec_ll - kl) + mean(pm_kl) -> backward -> [gradient all-reduce] -> global-norm clip +
    Adam (+ parameter EMA, non-finite steps skipped) -> step += 1."""

    plain_adam = False      # sumsq / clip / non-finite skip read the reduced gradient: the partial sums are added first


    def __init__(self, model, lr, batch_size: int, gradient_clip: float = 200.0, ema_rate: Optional[float] = 0.999,
                 weight_decay: float = 0.0, adam: Optional[Mapping[str, float]] = None, seed: int = 0, world_size: int = 1,
                 rank: int = 0, external_eps: bool = False, skip_nonfinite_updates: bool = True,
                 overlap_allreduce: bool = True):
        from ._lib import AdamCfg

        if model.store is None:
            model.init()
        s = model.store
        dev = s.device
        self.model, self.B, self.world_size, self.rank, self.seed = model, batch_size, world_size, rank, seed
        adam = dict(adam or {})
        c = AdamCfg()
        c.b1, c.b2, c.eps = adam.get("b1", 0.9), adam.get("b2", 0.999), adam.get("eps", 1e-8)
        from .optim import LinearSchedule

        if isinstance(lr, LinearSchedule):       # warm-up: optax.linear_schedule(0, config.lr, warm_up) (train_pm_vdvae.py:128-130)
            c.weight_decay, c.lr_kind, c.lr_init, c.lr_end = weight_decay, 1, lr.init_value, lr.end_value
            c.lr_decay_rate, c.lr_transition_steps = 1.0, lr.transition_steps
        else:
            c.weight_decay, c.lr_init, c.lr_decay_rate, c.lr_transition_steps = weight_decay, float(lr), 1.0, 1.0
            c.lr_kind, c.lr_end = 0, 0.0
        c.grad_scale = 1.0 / world_size
        c.zero_grad = 1
        self.adam_cfg, self.clip, self.skip = c, float(gradient_clip or 0.0), skip_nonfinite_updates
        self.reducer = _make_reducer(s, world_size, overlap_allreduce)
        self.ema_rate = ema_rate
        self.ema = s.flat_p.clone() if ema_rate is not None else None       # bax: ema_params start at the parameters
        H, W_, C = model.config["image_shape"]
        self.x = torch.zeros((batch_size, H, W_, C), device=dev)
        self.b = torch.zeros((batch_size, H, W_, 1), device=dev)
        shapes = model.eps_shapes(batch_size)
        sizes = [int(torch.Size(sh).numel()) for sh in shapes]
        self.eps_flat = torch.zeros(sum(sizes), device=dev)
        self.eps, off = [], 0
        for sh, n in zip(shapes, sizes):
            self.eps.append(self.eps_flat[off:off + n].view(sh))
            off += n
        self.external_eps = external_eps
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # training step (RNG counter)
        self.opt_count = torch.zeros(1, dtype=torch.int32, device=dev)     # optax count: not advanced by skipped steps
        self.gnorm_sq = torch.zeros(1, device=dev)
        self._gnorm_scratch = torch.zeros(1026, device=dev)      # pm_sumsq_det: partial sums + ticket
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        # Weight gradients on companion streams: measured on this chain of ~1 900 tiny launches they do NOT pay (B = 8 / 16:
        # 375 -> 364 / 632 -> 623 img/s; with 8 hardware queues the cross-queue waits triple the step): off unless asked for.
        # PM_VDVAE_WGRAD_STREAMS=n: n > 1 = a round-robin pool of n streams for the Blocks' weight gradients
        n_ws = int(os.environ.get("PM_VDVAE_WGRAD_STREAMS", "0"))
        model.ws.overlap_wgrad = n_ws if n_ws > 1 else (n_ws == 1)

    def _sequence(self) -> None:
        m, s = self.model, self.model.store
        if not self.external_eps:
            ops.normal_fill(self.eps_flat, self.seed, self.step_dev, stream_id=self.rank)
        m(self.x, self.b, self.eps)
        self._zero_grad(s)
        m.backward()
        self._reduce_partials(s)
        if self.reducer is not None:
            self.reducer.finish()       # the clip / non-finite decision below sees the REDUCED gradient on every rank
        ops.sumsq_det(s.flat_g, self.gnorm_sq, self._gnorm_scratch)     # fixed order: the clip factor has the same bits every run
        ops.adam_step_clip_ema(s.flat_p, s.flat_g, s.flat_m, s.flat_v, self.ema, s.n_decay, self.opt_count, self.gnorm_sq,
                               self.adam_cfg, self.clip, self.ema_rate if self.ema_rate is not None else 0.0, self.skip)
        self._grads_consumed(s)
        s.split_all()
        ops.counter_increment(self.step_dev)

    def step(self) -> None:
        with torch.cuda.stream(self.stream):
            self._planned(self._sequence)

    def set_batch(self, x: torch.Tensor, b: torch.Tensor, eps: Optional[Sequence[torch.Tensor]] = None) -> None:
        self.stream.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(self.stream):
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
            self.b.copy_(b.reshape(self.b.shape), non_blocking=True)
            if eps is not None:
                for dst, src in zip(self.eps, eps):
                    dst.copy_(src, non_blocking=True)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.stream.synchronize()
        v = self.model.metrics.cpu().tolist()
        return {"loss": v[0], "reconstruction_ll": v[1], "kl": v[2], "pm_kl": v[3], "bpd": v[4],
                "grad_norm": float(self.gnorm_sq.sqrt().item()) / self.world_size}

    def ema_params(self) -> Dict[str, torch.Tensor]:
        s = self.model.store
        return {n: self.ema[o:o + c].view(s.specs[n][0]).detach().clone() for n, (o, c) in s.offsets.items()}

    def swap_in_ema(self) -> None:
        """exchanges the parameters with their EMA (Trainer(use_ema_for_eval=True)); call again to swap back"""
        s = self.model.store
        if self.ema is None:
            return
        with torch.cuda.stream(self.stream):
            tmp = s.flat_p.clone()
            s.flat_p.copy_(self.ema)
            self.ema.copy_(tmp)
            s.split_all()

    def evaluate(self, x: torch.Tensor, b: torch.Tensor) -> Dict[str, float]:
        """loss_fn on a validation batch with the EMA parameters (use_ema_for_eval), fresh posterior noise"""
        self.swap_in_ema()
        self.set_batch(x, b)
        with torch.cuda.stream(self.stream):
            ops.normal_fill(self.eps_flat, self.seed + 104729, self.step_dev, stream_id=1000 + self.rank)
            self.model(self.x, self.b, self.eps)
        out = self.read_metrics()
        self.swap_in_ema()
        out.pop("grad_norm", None)
        return out


class VADETrainStep(_PlannedStep):
    """train_vade.py as launch sequences: `mode="pretrain"` is pretrain_loss_fn (:45-49: loss = -mean decoder(encoder(x).mean())
    .log_prob(x)) under optax.adam(pretrain_lr); `mode="elbo"` is loss_fn (:51-55: loss = -mean VADE.elbo(x)) under
    optax.chain(scale_by_adam(**adam), scale_by_schedule(exponential_decay), scale(-1)).  eps -> forward -> loss -> backward ->
    [gradient all-reduce] -> Adam -> step += 1; one stream (the model is one chain)."""

    def __init__(self, model, optimizer: Chain, batch_size: int, x_shape, mode: str = "elbo", seed: int = 0,
                 world_size: int = 1, rank: int = 0, external_eps: bool = False, use_plan: bool = True):
        if mode not in ("pretrain", "elbo"):
            raise ValueError(mode)
        if model.store is None:
            model.init(x_shape)
        dev = model.store.device
        self.model, self.mode, self.B, self.seed, self.rank, self.world_size = model, mode, batch_size, seed, rank, world_size
        self.adam_cfg = optimizer.adam_cfg(grad_scale=1.0 / world_size)
        self.adam_cfg.zero_grad = 1
        # a new Trainer = a fresh optimizer state (train_vade.py builds one for pre-training and another for the ELBO phase on the
        # same model): the store's Adam moments restart from zero
        ops.fill_zero(model.store.flat_m)
        ops.fill_zero(model.store.flat_v)
        self.reducer = _make_reducer(model.store, world_size, False)
        self.x = torch.zeros((batch_size,) + tuple(x_shape), device=dev)
        self.eps = torch.zeros((batch_size, model.latent_dim), device=dev)
        self.external_eps = external_eps
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.metrics = torch.zeros(8, device=dev)
        self.g = torch.zeros(batch_size, device=dev)
        self.use_plan = use_plan
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))

    def _sequence(self) -> None:
        m, s = self.model, self.model.store
        if self.mode == "elbo":
            if not self.external_eps:
                ops.normal_fill(self.eps, self.seed, self.step_dev, stream_id=self.rank)
            value = m.elbo(self.x, self.eps, is_training=True)
        else:
            value = m.reconstruction_ll_at_mean(self.x, is_training=True)
        ops.neg_mean_loss(value, 1.0 / self.B, self.metrics, self.g)        # loss = -mean(value); g = d loss / d value
        self._zero_grad(s)
        if self.mode == "elbo":
            m.backward_elbo(self.g)
        else:
            m.backward_reconstruction_at_mean(self.g)
        self._reduce_partials(s)
        if self.reducer is not None:
            self.reducer.finish()
        self._adam_step(s, self.step_dev)
        self._grads_consumed(s)
        s.split_all()
        ops.counter_increment(self.step_dev)

    def step(self) -> None:
        with torch.cuda.stream(self.stream):
            self._planned(self._sequence)

    def set_batch(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None) -> None:
        self.stream.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(self.stream):
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
            if eps is not None:
                self.eps.copy_(eps, non_blocking=True)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.stream.synchronize()
        return {"loss": self.metrics[0].item()}

    def evaluate(self, x: torch.Tensor) -> Dict[str, float]:
        """loss_fn with is_training=False on a validation batch (fresh posterior noise)"""
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            m = self.model
            x = x.reshape((x.shape[0],) + tuple(self.x.shape[1:])).contiguous()
            value = m.elbo(x, None, seed=self.seed + 104729) if self.mode == "elbo" else m.reconstruction_ll_at_mean(x)
            metrics = torch.zeros(8, device=x.device)
            ops.neg_mean_loss(value, 1.0 / x.shape[0], metrics, None)
        self.stream.synchronize()
        return {"loss": metrics[0].item()}


class PMVADETrainStep(_PlannedStep):
    """train_pm_vade.py:40-83 as one launch sequence: z ~ q(z | x) of the frozen VaDE encoder, loss = -mean log q(z | x_o) of the
    partial encoder; backward through the partial encoder and its distribution only (trainable_predicate: "partial_" in
    module_name - they live on model.partial_store, the only buffer the optimizer touches); Adam with the exponential-decay
    schedule; step += 1."""

    def __init__(self, model, optimizer: Chain, batch_size: int, x_shape, seed: int = 0, world_size: int = 1, rank: int = 0,
                 external_eps: bool = False, use_plan: bool = True):
        if model.store is None:
            model.init(x_shape)
        dev = model.store.device
        self.model, self.store = model, model.partial_store
        self.B, self.seed, self.rank, self.world_size = batch_size, seed, rank, world_size
        self.adam_cfg = optimizer.adam_cfg(grad_scale=1.0 / world_size)
        self.adam_cfg.zero_grad = 1
        self.reducer = _make_reducer(self.store, world_size, False)
        x_shape = tuple(x_shape)
        b_shape = x_shape[:-1] + (1,) if len(x_shape) == 3 else x_shape
        self.x = torch.zeros((batch_size,) + x_shape, device=dev)
        self.b = torch.zeros((batch_size,) + b_shape, device=dev)
        self.eps = torch.zeros((batch_size, model.latent_dim), device=dev)
        self.external_eps = external_eps
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.metrics = torch.zeros(8, device=dev)
        self.g = torch.zeros(batch_size, device=dev)
        self.use_plan = use_plan
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))

    def _sequence(self) -> None:
        m, s = self.model, self.store
        if not self.external_eps:
            ops.normal_fill(self.eps, self.seed, self.step_dev, stream_id=self.rank)
        ll = m.posterior_matching_ll(self.x, self.b, self.eps, is_training=True)
        ops.neg_mean_loss(ll, 1.0 / self.B, self.metrics, self.g)
        self._zero_grad(s)
        m.backward_posterior_matching_ll(self.g)
        self._reduce_partials(s)
        if self.reducer is not None:
            self.reducer.finish()
        self._adam_step(s, self.step_dev)
        self._grads_consumed(s)
        s.split_all()
        ops.counter_increment(self.step_dev)

    def step(self) -> None:
        with torch.cuda.stream(self.stream):
            self._planned(self._sequence)

    def set_batch(self, x: torch.Tensor, b: torch.Tensor, eps: Optional[torch.Tensor] = None) -> None:
        self.stream.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(self.stream):
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
            self.b.copy_(b.reshape(self.b.shape), non_blocking=True)
            if eps is not None:
                self.eps.copy_(eps, non_blocking=True)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.stream.synchronize()
        return {"loss": self.metrics[0].item()}

    def evaluate(self, x: torch.Tensor, b: torch.Tensor) -> Dict[str, float]:
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            ll = self.model.posterior_matching_ll(x.contiguous(), b.contiguous(), None, seed=self.seed + 104729)
            metrics = torch.zeros(8, device=x.device)
            ops.neg_mean_loss(ll, 1.0 / x.shape[0], metrics, None)
        self.stream.synchronize()
        return {"loss": metrics[0].item()}


class LookaheadTrainStep(_PlannedStep):
    """train_lookahead_posterior.py:46-96 as one launch sequence: the frozen PM-VAE produces the one-step-ahead latent samples
    (LookaheadPosterior.model_one_step_z), loss = -mean lookahead_lls; backward through the lookahead encoder only
    (trainable_predicate: "lookahead" in module_name - its parameters live on model.store, the only buffer the optimizer
    touches); Adam with the exponential-decay schedule; step += 1.  The subsampled feature indices are drawn on the host every
    step (jax.random.choice without replacement, lookahead.py:151-156), so the step is issued eagerly (no launch plan)."""

    use_plan = False

    def __init__(self, model, optimizer: Chain, batch_size: int, x_shape, seed: int = 0, world_size: int = 1, rank: int = 0,
                 external_noise: bool = False):
        import numpy as np

        if model.store is None:
            model.init(x_shape)
        dev = model.store.device
        self.model, self.store = model, model.store
        self.B, self.seed, self.rank, self.world_size = batch_size, seed, rank, world_size
        self.adam_cfg = optimizer.adam_cfg(grad_scale=1.0 / world_size)
        self.adam_cfg.zero_grad = 1
        self.reducer = _make_reducer(self.store, world_size, False)
        x_shape = tuple(x_shape)
        self.x = torch.zeros((batch_size,) + x_shape, device=dev)
        self.b = torch.zeros((batch_size,) + x_shape[:-1] + (1,), device=dev)
        self.external_noise = external_noise
        self.noise, self.inds = None, None          # external_noise: set through set_batch
        self._rng = np.random.default_rng(seed + 7 * rank)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.metrics = torch.zeros(8, device=dev)
        self.g = torch.zeros(batch_size, device=dev)
        self.stream = _step_stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))

    def _sequence(self) -> None:
        m, s = self.model, self.store
        if self.external_noise:
            noise, inds = self.noise, self.inds
        else:
            noise, inds = None, m.draw_indices(self._rng)
        ll = m(self.x, self.b, is_training=True, noise=noise, inds=inds, seed=self.seed + self.rank)
        ops.neg_mean_loss(ll, 1.0 / self.B, self.metrics, self.g)
        self._zero_grad(s)
        m.backward(self.g)
        self._reduce_partials(s)
        if self.reducer is not None:
            self.reducer.finish()
        self._adam_step(s, self.step_dev)
        self._grads_consumed(s)
        s.split_all()
        ops.counter_increment(self.step_dev)

    def step(self) -> None:
        with torch.cuda.stream(self.stream):
            self._planned(self._sequence)

    def set_batch(self, x: torch.Tensor, b: torch.Tensor, noise=None, inds=None) -> None:
        self.stream.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(self.stream):
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
            self.b.copy_(b.reshape(self.b.shape), non_blocking=True)
        if noise is not None:
            self.noise, self.inds = noise, inds

    def synchronize(self) -> None:
        self.stream.synchronize()

    def read_metrics(self) -> Dict[str, float]:
        self.stream.synchronize()
        return {"loss": self.metrics[0].item()}

    def evaluate(self, x: torch.Tensor, b: torch.Tensor) -> Dict[str, float]:
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            ll = self.model(x.contiguous(), b.contiguous(), is_training=False, seed=self.seed + 104729)
            metrics = torch.zeros(8, device=x.device)
            ops.neg_mean_loss(ll, 1.0 / x.shape[0], metrics, None)
        self.stream.synchronize()
        return {"loss": metrics[0].item()}
