"""A trainer with the constructor / fit() surface the reference uses from bax
(train_pm_vae.py:85-102; bax itself is third-party and not in the reference tree).

loss_fn must be a `PMVAELoss` (train_pm_vae.py builds it) or a `VQVAELoss` (train_vqvae.py): its
forward, loss, backward and the optimizer run as the fused HIP step of engine.PMVAETrainStep /
engine.VQVAETrainStep.  One process drives one GPU;
`num_devices` must equal the torch.distributed world size (launch N processes with
`python -m torch.distributed.run --nproc-per-node N train_pm_vae.py ...`).
"""
from __future__ import annotations

import pickle
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Mapping, Optional

import torch

from . import ops
from .engine import LookaheadTrainStep, PMVADETrainStep, PMVAETrainStep, PMVQVAETrainStep, VADETrainStep, VDVAETrainStep, VQVAETrainStep
from .models.vae import PosteriorMatchingVAE
from .models.vqvae import VQVAE
from .optim import Chain
from .parallel import allreduce_mean_scalars, init_distributed
from .utils import Callback


def _step_counter(step, device) -> torch.Tensor:
    return torch.tensor([int(step)], dtype=torch.int32, device=device)


def _scalars(metrics: torch.Tensor, names) -> Dict[str, torch.Tensor]:
    return {n: metrics[i] for i, n in enumerate(names)}


class PMVAELoss:
    """loss_fn(step, is_training, batch) of train_pm_vae.py:58-72.  Handed to Trainer it is lowered to the fused train
    step (engine.PMVAETrainStep) instead of being traced; CALLED, it evaluates the same function through the same
    kernels - forward + loss only - and returns (loss, aux) like the reference: 0-dim device tensors, aux = the batch means
    of reconstruction_ll / kl / matching_ll and beta(step).  The posterior noise (hk.next_rng_key in the reference) is a
    device Philox draw keyed by (`seed`, step) unless `eps` is given."""

    def __init__(self, config: Mapping[str, Any], model: PosteriorMatchingVAE, data_key: str = "image", seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, model, data_key, seed

    def __call__(self, step, is_training, batch, eps: Optional[torch.Tensor] = None):
        from .engine import loss_cfg_from_config
        from .models.networks import ResidualMLP

        m = self.model
        x = batch[self.data_key]
        if m.store is None:
            m.init(tuple(x.shape[1:]), seed=self.seed)
        dev = m.store.device
        x, b = x.to(dev).float().contiguous(), batch["mask"].to(dev).float().contiguous()
        B = x.shape[0]
        step_dev = _step_counter(step, dev)
        if eps is None:
            eps = torch.empty((B, m.latent_dim), device=dev)
            ops.normal_fill(eps, self.seed, step_dev, stream_id=0)
        for i, net in enumerate((m.encoder_net, m.decoder_net, m.partial_encoder_net)):
            if isinstance(net, ResidualMLP) and getattr(net, "dropout_step_dev", None) is None:
                net.dropout_seed, net.dropout_step_dev, net.dropout_stream_base = self.seed, step_dev, 2000 + 64 * i
        out = m(x, b, is_training=bool(is_training), eps=eps)
        metrics = torch.zeros(8, device=dev)
        ops.pmvae_loss(out["reconstruction_ll"], out["kl"], out["matching_ll"], loss_cfg_from_config(self.config, B), step_dev,
                       metrics, None, None, None)
        aux = _scalars(metrics, ("loss", "reconstruction_ll", "kl", "matching_ll", "beta"))
        return aux.pop("loss"), aux


class VQVAELoss:
    """loss_fn(step, is_training, batch) of train_vqvae.py:67-75: returns (out["loss"], aux) with aux perplexity /
    reconstruction_loss / vq_loss.  Lowered to engine.VQVAETrainStep by Trainer; callable for validation / inspection
    (is_training=True updates the EMA codebook state, as the haiku module does in the reference)."""

    def __init__(self, config: Mapping[str, Any], model: VQVAE, data_key: str = "image", seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, model, data_key, seed

    def __call__(self, step, is_training, batch):
        m = self.model
        x = batch[self.data_key]
        if m.store is None:
            m.init(tuple(x.shape[1:]), seed=self.seed)
        m(x.to(m.store.device).float().contiguous(), is_training=bool(is_training))
        aux = _scalars(m.metrics.clone(), ("loss", "reconstruction_loss", "vq_loss", "perplexity"))
        return aux.pop("loss"), aux


class PMVQVAELoss:
    """loss_fn of train_pm_vqvae.py:81-99: frozen VQ-VAE -> code indices, partial encoder -> conditional vector,
    loss = -mean PixelCNN.log_prob, aux = {}.  `vqvae` must carry the stage-1 parameters and state; `partial_encoder` /
    `pixel_cnn` come from models.vqvae.build_partial_posterior.  Lowered to engine.PMVQVAETrainStep by Trainer; callable
    (forward + loss through the same kernels; is_training=True draws the PixelCNN's dropout masks from (seed, step))."""

    def __init__(self, config: Mapping[str, Any], vqvae: VQVAE, partial_encoder, pixel_cnn, data_key: str = "image",
                 seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, vqvae, data_key, seed
        self.partial_encoder, self.pixel_cnn = partial_encoder, pixel_cnn
        self._eval = {}

    def __call__(self, step, is_training, batch):
        x = batch[self.data_key]
        B, xs = x.shape[0], tuple(x.shape[1:])
        ev = self._eval.get((B, xs))
        if ev is None:
            ev = PMVQVAETrainStep(self.model, self.partial_encoder, self.pixel_cnn, None, B, xs, seed=self.seed)
            self._eval[(B, xs)] = ev
        dev = ev.x.device
        ev.set_batch(x.to(dev).float(), batch["mask"].to(dev).float())
        with torch.cuda.stream(ev.stream):
            ev.step_dev.fill_(int(step))
            ev.forward(bool(is_training))
        ev.synchronize()
        return ev.metrics[0].clone(), {}


class VDVAELoss:
    """loss_fn of train_pm_vdvae.py:109-120: loss = -mean(rec_ll - kl) + mean(pm_kl), aux = the batch means of
    reconstruction_ll / kl / pm_kl and bpd.  Lowered to engine.VDVAETrainStep by Trainer; callable (the model has no
    train / eval difference: `is_training` is accepted and ignored, as in the reference; posterior noise = device Philox
    keyed by (`seed`, step) unless `eps` is given)."""

    def __init__(self, config: Mapping[str, Any], model, data_key: str = "image", seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, model, data_key, seed

    def __call__(self, step, is_training, batch, eps=None):
        m = self.model
        if m.store is None:
            m.init(seed=self.seed)
        dev = m.store.device
        x, b = batch[self.data_key].to(dev).float().contiguous(), batch["mask"].to(dev).float().contiguous()
        if eps is None:
            shapes = m.eps_shapes(x.shape[0])
            flat = torch.empty(sum(int(torch.Size(sh).numel()) for sh in shapes), device=dev)
            ops.normal_fill(flat, self.seed, _step_counter(step, dev), stream_id=0)
            eps, off = [], 0
            for sh in shapes:
                n = int(torch.Size(sh).numel())
                eps.append(flat[off:off + n].view(sh))
                off += n
        m(x, b, eps)
        aux = _scalars(m.metrics.clone(), ("loss", "reconstruction_ll", "kl", "pm_kl", "bpd"))
        return aux.pop("loss"), aux


class VADEPretrainLoss:
    """pretrain_loss_fn of train_vade.py:45-49: loss = -mean decoder(encoder(x).mean()).log_prob(x), aux = {}.  Lowered to
    engine.VADETrainStep(mode="pretrain") by Trainer; callable like the reference's function."""

    mode = "pretrain"

    def __init__(self, config: Mapping[str, Any], model, data_key: str = "image", seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, model, data_key, seed

    def _value(self, step, is_training, x):
        return self.model.reconstruction_ll_at_mean(x, is_training=bool(is_training))

    def __call__(self, step, is_training, batch):
        m = self.model
        x = batch[self.data_key]
        if m.store is None:
            m.init(tuple(x.shape[1:]), seed=self.seed)
        x = x.to(m.store.device).float().contiguous()
        metrics = torch.zeros(8, device=m.store.device)
        ops.neg_mean_loss(self._value(step, is_training, x), 1.0 / x.shape[0], metrics, None)
        return metrics[0], {}


class VADELoss(VADEPretrainLoss):
    """loss_fn of train_vade.py:51-55: loss = -mean VADE.elbo(x), aux = {} (posterior noise: device Philox keyed by (seed, step))"""

    mode = "elbo"

    def _value(self, step, is_training, x):
        m = self.model
        eps = torch.empty((x.shape[0], m.latent_dim), device=x.device)
        ops.normal_fill(eps, self.seed, _step_counter(step, x.device), stream_id=0)
        return m.elbo(x, eps, is_training=bool(is_training))


class PMVADELoss:
    """loss_fn of train_pm_vade.py:40-43: loss = -mean PosteriorMatchingVADE.posterior_matching_ll(x, mask), aux = {}.  Lowered
    to engine.PMVADETrainStep by Trainer (only the "partial_" modules train, :59-60); callable."""

    def __init__(self, config: Mapping[str, Any], model, data_key: str = "image", seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, model, data_key, seed

    def __call__(self, step, is_training, batch):
        m = self.model
        x = batch[self.data_key]
        if m.store is None:
            m.init(tuple(x.shape[1:]), seed=self.seed)
        dev = m.store.device
        x, b = x.to(dev).float().contiguous(), batch["mask"].to(dev).float().contiguous()
        eps = torch.empty((x.shape[0], m.latent_dim), device=dev)
        ops.normal_fill(eps, self.seed, _step_counter(step, dev), stream_id=0)
        metrics = torch.zeros(8, device=dev)
        ops.neg_mean_loss(m.posterior_matching_ll(x, b, eps, is_training=bool(is_training)), 1.0 / x.shape[0], metrics, None)
        return metrics[0], {}


class LookaheadLoss:
    """loss_fn of train_lookahead_posterior.py:46-52: loss = -mean LookaheadPosterior(...)(x, mask), aux = {}.  Lowered to
    engine.LookaheadTrainStep by Trainer (only the "lookahead" modules train, :61-62); callable."""

    def __init__(self, config: Mapping[str, Any], model, data_key: str = "image", seed: int = 0):
        self.config, self.model, self.data_key, self.seed = config, model, data_key, seed

    def __call__(self, step, is_training, batch):
        m = self.model
        x = batch[self.data_key]
        if m.store is None:
            m.init(tuple(x.shape[1:]), seed=self.seed)
        dev = m.store.device
        x, b = x.to(dev).float().contiguous(), batch["mask"].to(dev).float().contiguous()
        metrics = torch.zeros(8, device=dev)
        s = int(step.item()) if isinstance(step, torch.Tensor) else int(step)
        ops.neg_mean_loss(m(x, b, is_training=bool(is_training), seed=self.seed + s), 1.0 / x.shape[0], metrics, None)
        return metrics[0], {}


@dataclass
class TrainState:
    step: int = 0
    params: Dict[str, torch.Tensor] = field(default_factory=dict)
    state: Dict[str, Any] = field(default_factory=dict)       # no mutable module state in the PM-VAE
    opt_state: Dict[str, Any] = field(default_factory=dict)
    ema_params: Optional[Dict[str, torch.Tensor]] = None


class CheckpointCallback(Callback):
    """bax.callbacks.CheckpointCallback(path): pickles the TrainState at validation time
    (train_pm_vae.py:91)."""

    def __init__(self, path: str):
        self._path = path

    def on_validation_end(self, train_state: TrainState, step: int, logs) -> None:
        with open(self._path, "wb") as fp:
            pickle.dump(train_state, fp)


class LearningRateLoggerCallback(Callback):
    def __init__(self, schedule):
        self._schedule = schedule

    def on_validation_end(self, train_state, step, logs) -> None:
        logs["learning_rate"] = self._schedule(step)


class Trainer:
    def __init__(self, loss_fn: PMVAELoss, optimizer: Chain, num_devices: int = 1, seed: int = 0,
                 trainable_predicate=None, skip_nonfinite_updates: bool = False, ema_rate: Optional[float] = None,
                 use_ema_for_eval: bool = False, use_graph: bool = False):
        if isinstance(loss_fn, PMVQVAELoss):
            # train_pm_vqvae.py:122-123: the only predicate the reference uses freezes every module under "vqvae/";
            # that is exactly what PMVQVAETrainStep does (the VQ-VAE lives on its own, never-updated store)
            if trainable_predicate is not None and (trainable_predicate("vqvae/encoder", "w", None)
                                                    or not trainable_predicate("pixel_cnn", "w", None)):
                raise NotImplementedError("only the reference's predicate (freeze 'vqvae/...') is lowered")
            trainable_predicate = None
        self.ema_rate, self.skip_nonfinite = ema_rate, skip_nonfinite_updates
        if isinstance(loss_fn, VDVAELoss):     # train_pm_vdvae.py:146-154: the flags the fused VDVAE step implements
            skip_nonfinite_updates, ema_rate, use_ema_for_eval = False, None, False
        if isinstance(loss_fn, PMVADELoss):
            # train_pm_vade.py:59-60: the only predicate the reference uses trains the modules whose name contains "partial_";
            # that is what PMVADETrainStep does (they live on their own store, the VaDE's is never updated)
            if trainable_predicate is not None and (trainable_predicate("vade", "mu", None)
                                                    or not trainable_predicate("partial_encoder_net", "w", None)):
                raise NotImplementedError("only the reference's predicate ('partial_' in module_name) is lowered")
            trainable_predicate = None
        if isinstance(loss_fn, LookaheadLoss):
            # train_lookahead_posterior.py:61-62: "lookahead" in module_name - what LookaheadTrainStep does (own store)
            if trainable_predicate is not None and (trainable_predicate("partial_encoder_net", "w", None)
                                                    or not trainable_predicate("lookahead_encoder_net", "w", None)):
                raise NotImplementedError("only the reference's predicate ('lookahead' in module_name) is lowered")
            trainable_predicate = None
        if not isinstance(loss_fn, (PMVAELoss, VQVAELoss, PMVQVAELoss, VDVAELoss, VADEPretrainLoss, PMVADELoss, LookaheadLoss)):
            raise NotImplementedError("Trainer lowers PMVAELoss / VQVAELoss (the loss_fn of train_pm_vae.py / "
                                      "train_vqvae.py) to the fused HIP step; arbitrary Python loss functions have "
                                      "no HIP path")
        if trainable_predicate is not None or skip_nonfinite_updates or ema_rate is not None or use_ema_for_eval:
            raise NotImplementedError("trainable_predicate / skip_nonfinite_updates / EMA are used by the VQ-VAE and "
                                      "VDVAE scripts only (SURVEY.md 8a-16,20): not on the PM-VAE path")
        self.loss_fn, self.optimizer, self.seed, self.use_graph = loss_fn, optimizer, seed, use_graph
        self.rank, self.local_rank, self.world = init_distributed()
        if num_devices != self.world:
            raise ValueError(f"num_devices={num_devices} but {self.world} process(es) are running: this engine is one "
                             "process per GPU, start it with torch.distributed.run --nproc-per-node num_devices")

    def _state(self, ts) -> TrainState:
        if isinstance(ts, VDVAETrainStep):
            store = ts.model.store
            ema = {k: v.cpu() for k, v in ts.ema_params().items()} if ts.ema is not None else None
            return TrainState(step=int(ts.step_dev.item()), params={k: v.cpu() for k, v in store.to_dict("p").items()},
                              opt_state={"mu": store.flat_m.cpu(), "nu": store.flat_v.cpu(),
                                         "count": int(ts.opt_count.item())}, ema_params=ema)
        if isinstance(ts, LookaheadTrainStep):   # frozen PM-VAE + trainable lookahead encoder in one tree
            return TrainState(step=int(ts.step_dev.item()), params={k: v.cpu() for k, v in ts.model.params_dict().items()},
                              opt_state={"mu": ts.store.flat_m.cpu(), "nu": ts.store.flat_v.cpu()})
        if isinstance(ts, PMVADETrainStep):      # frozen VaDE + trainable partial encoder in one tree, as in the reference
            params = {k: v.cpu() for k, v in ts.model.store.to_dict("p").items()}
            params.update({k: v.cpu() for k, v in ts.store.to_dict("p").items()})
            return TrainState(step=int(ts.step_dev.item()), params=params,
                              opt_state={"mu": ts.store.flat_m.cpu(), "nu": ts.store.flat_v.cpu()})
        if isinstance(ts, PMVQVAETrainStep):
            # the reference's TrainState holds frozen and trainable parameters in one tree (vqvae/ prefix, :123)
            params = {f"vqvae/{k}": v.cpu() for k, v in ts.vqvae.params_dict().items()}
            params.update({k: v.cpu() for k, v in ts.store.to_dict("p").items()})
            state = {f"vqvae/{k}": v.cpu() for k, v in ts.vqvae.state_dict().items()}
            return TrainState(step=int(ts.step_dev.item()), params=params, state=state,
                              opt_state={"mu": ts.store.flat_m.cpu(), "nu": ts.store.flat_v.cpu()})
        store = ts.model.store
        state = {k: v.cpu() for k, v in ts.model.state_dict().items()} if isinstance(ts.model, VQVAE) else {}
        return TrainState(step=int(ts.step_dev.item()), params={k: v.cpu() for k, v in store.to_dict("p").items()},
                          state=state, opt_state={"mu": store.flat_m.cpu(), "nu": store.flat_v.cpu()})

    def fit(self, train_dataset: Iterable[Dict[str, torch.Tensor]], steps: int, val_dataset=None,
            validation_freq: Optional[int] = None, callbacks: Optional[List[Callback]] = None,
            initial_params: Optional[Dict[str, Any]] = None, initial_state=None, log_fn=print) -> TrainState:
        lf = self.loss_fn
        model, key = lf.model, lf.data_key
        it = iter(train_dataset)
        first = next(it)
        x0 = first[key]
        B, x_shape = x0.shape[0], tuple(x0.shape[1:])
        is_vade, is_pmvade = isinstance(lf, VADEPretrainLoss), isinstance(lf, PMVADELoss)
        if model.store is None:                                                              # same init on all ranks
            if isinstance(lf, VDVAELoss):
                model.init(device=torch.device("cuda", self.local_rank), seed=self.seed)
            else:
                model.init(x_shape, device=torch.device("cuda", self.local_rank), seed=self.seed)
        def strip(d):     # stage-1 checkpoints may carry the reference's "vqvae/" module prefix (train_pm_vqvae.py:123)
            return {(k[len("vqvae/"):] if k.startswith("vqvae/") else k): v for k, v in d.items()}

        if initial_params is not None:
            model.load_params(strip(initial_params))
        if initial_state is not None and isinstance(model, VQVAE):
            model.load_state(strip(initial_state))
        dev = model.store.device
        self._broadcast_initial_state(model, lf)
        is_vq = isinstance(lf, VQVAELoss)
        is_pmvq = isinstance(lf, PMVQVAELoss)
        if is_vade:
            ts = VADETrainStep(model, self.optimizer, B, x_shape, mode=lf.mode, seed=self.seed, world_size=self.world,
                               rank=self.rank)
        elif is_pmvade:
            ts = PMVADETrainStep(model, self.optimizer, B, x_shape, seed=self.seed, world_size=self.world, rank=self.rank)
        elif isinstance(lf, LookaheadLoss):
            ts = LookaheadTrainStep(model, self.optimizer, B, x_shape, seed=self.seed, world_size=self.world, rank=self.rank)
        elif isinstance(lf, VDVAELoss):
            opt = self.optimizer
            from .optim import LinearSchedule

            lr = opt.schedule if isinstance(opt.schedule, LinearSchedule) else opt.schedule.init_value
            ts = VDVAETrainStep(model, lr, B, gradient_clip=opt.clip.max_norm if opt.clip else 0.0,
                                ema_rate=self.ema_rate, weight_decay=opt.decay.weight_decay,
                                adam={"b1": opt.adam.b1, "b2": opt.adam.b2, "eps": opt.adam.eps}, seed=self.seed,
                                world_size=self.world, rank=self.rank, skip_nonfinite_updates=self.skip_nonfinite)
        elif is_pmvq:
            ts = PMVQVAETrainStep(model, lf.partial_encoder, lf.pixel_cnn, self.optimizer, B, x_shape, seed=self.seed,
                                  world_size=self.world, rank=self.rank)
        elif is_vq:
            if self.world > 1:
                model.vq.cross_replica_axis = "i"          # psum of the EMA statistics across ranks
            ts = VQVAETrainStep(model, self.optimizer, B, x_shape, world_size=self.world, rank=self.rank,
                                use_graph=self.use_graph)
        else:
            ts = PMVAETrainStep(model, lf.config, self.optimizer, B, x_shape, seed=self.seed, world_size=self.world,
                                rank=self.rank, use_graph=self.use_graph)
        callbacks = callbacks or []
        batch = first
        from .utils import steady_state_gc

        with steady_state_gc() as gcs:
            self._fit_loop(ts, steps, batch, it, key, dev, is_vq or is_vade, val_dataset, validation_freq, callbacks, log_fn, gcs)
        ts.synchronize()
        return self._state(ts)

    def _broadcast_initial_state(self, model, lf) -> None:
        """Replicas must START identical: only gradients are all-reduced afterwards.  The seeded init is the same on every rank,
        but `initial_params` may be rank-local (train_vade.py fits its GMM on each rank's own data shard; the reference runs that
        phase on one device, train_vade.py:88-121), so rank 0's parameters (and haiku state) go to everyone before the first
        step (bax replicates the train state onto its devices the same way)."""
        if self.world <= 1:
            return
        import torch.distributed as dist

        stores = [model.store] + [getattr(m, "store", None) for m in
                                  (getattr(lf, "partial_encoder", None), getattr(lf, "pixel_cnn", None))]
        stores += [getattr(model, n, None) for n in ("partial_store",)]
        seen = set()
        for st in stores:
            if st is None or id(st) in seen or getattr(st, "flat_p", None) is None:
                continue
            seen.add(id(st))
            dist.broadcast(st.flat_p, src=0)
            st.split_all()
        vq = getattr(model, "vq", None)
        if vq is not None and getattr(vq, "state", None):
            for k in sorted(vq.state):
                dist.broadcast(vq.state[k], src=0)

    def _fit_loop(self, ts, steps, batch, it, key, dev, unmasked, val_dataset, validation_freq, callbacks, log_fn, gcs) -> None:
        for step in range(steps):
            if unmasked:
                ts.set_batch(batch[key].to(dev, non_blocking=True))
            else:
                ts.set_batch(batch[key].to(dev, non_blocking=True), batch["mask"].to(dev, non_blocking=True))
            # datasets that refill static device buffers (uint8 gather, device masks) do so on the current stream: it must
            # not run ahead of the copies set_batch queued on the step's stream (it may overlap the step itself)
            torch.cuda.current_stream(dev).wait_stream(ts.stream)
            ts.step()
            if validation_freq and ((step + 1) % validation_freq == 0 or step + 1 == steps):
                logs = dict(ts.read_metrics())
                logs = {f"train_{k}": v for k, v in logs.items()}
                if val_dataset is not None:
                    logs.update(self._validate(ts, val_dataset, key, dev, callbacks))
                state = self._state(ts)
                if self.rank == 0:
                    for cb in callbacks:
                        cb.on_validation_end(state, step + 1, logs)
                    if log_fn:
                        log_fn(f"step {step + 1}: " + ", ".join(f"{k}={v:.5g}" for k, v in logs.items()
                                                                 if getattr(v, "ndim", 0) == 0))
                gcs.collect()                       # the safe point for a cyclic collection
            elif (step + 1) % 1000 == 0:
                gcs.collect()                       # runs without validation points (train_vade.py's 70 k pretraining steps)
            batch = next(it)

    def _validate(self, ts, val_dataset, key: str, dev, callbacks=()) -> Dict[str, float]:
        """loss_fn with is_training=False averaged over the validation batches (bax semantics); every callback sees each
        validation batch first (bax: on_validation_step - the clustering-accuracy callback of train_vade.py collects there)."""
        sums: Dict[str, float] = {}
        n = 0
        batches = getattr(val_dataset, "batches", None) or list(val_dataset)
        for i, vb in enumerate(batches):
            if self.rank == 0:                      # on_validation_end only runs there (other ranks would collect forever)
                for cb in callbacks or ():
                    cb.on_validation_step(None, None, vb)
            if isinstance(ts, (VQVAETrainStep, VADETrainStep)):
                out = ts.evaluate(vb[key].to(dev))
            elif isinstance(ts, (PMVQVAETrainStep, VDVAETrainStep, PMVADETrainStep, LookaheadTrainStep)):
                out = ts.evaluate(vb[key].to(dev), vb["mask"].to(dev))
            else:
                x, b = vb[key].to(dev), vb["mask"].to(dev)
                eps = torch.empty((x.shape[0], ts.model.latent_dim), device=dev)
                with torch.cuda.stream(ts.stream):
                    ops.normal_fill(eps, self.seed + 7919, ts.step_dev, stream_id=1000 + i)
                out = ts.evaluate(x, b, eps)
            for k, v in out.items():
                sums[k] = sums.get(k, 0.0) + v
            n += 1
        vals = torch.tensor([sums[k] / n for k in sorted(sums)], dtype=torch.float64)
        if self.world > 1:
            vals = allreduce_mean_scalars(vals.to(dev)).cpu()
        return {f"val_{k}": float(v) for k, v in zip(sorted(sums), vals)}
