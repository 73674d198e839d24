"""The optax pieces the reference's train_pm_vae.py:74-83 composes, as plain spec objects.

Only the exact chain the reference builds is executable - it is lowered to ONE fused HIP kernel
(pm_adam_step, csrc/pm_optim.hip).  Any other composition raises instead of silently running a
different optimizer.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Optional


@dataclass
class ExponentialDecay:
    """optax.exponential_decay(init_value, transition_steps, decay_rate), non-staircase."""

    init_value: float
    transition_steps: float
    decay_rate: float

    def __call__(self, count: int) -> float:
        return self.init_value * self.decay_rate ** (count / self.transition_steps)


def exponential_decay(init_value, transition_steps, decay_rate, staircase=False, **unsupported) -> ExponentialDecay:
    if staircase:                                   # configs/vade_mnist.py passes staircase=False explicitly
        unsupported["staircase"] = staircase
    if unsupported:
        raise NotImplementedError(f"exponential_decay options {sorted(unsupported)} have no HIP path")
    return ExponentialDecay(float(init_value), float(transition_steps), float(decay_rate))


@dataclass
class ScaleByAdam:
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    eps_root: float = 0.0


def scale_by_adam(b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0) -> ScaleByAdam:
    if eps_root != 0.0:
        raise NotImplementedError("eps_root != 0")
    return ScaleByAdam(b1, b2, eps, eps_root)


@dataclass
class AddDecayedWeights:
    weight_decay: float = 0.0
    mask: Any = None  # the reference's mask is `ndim != 1`; that is what the kernel implements


def add_decayed_weights(weight_decay=0.0, mask=None) -> AddDecayedWeights:
    return AddDecayedWeights(float(weight_decay), mask)


@dataclass
class ScaleBySchedule:
    schedule: Callable[[int], float]


def scale_by_schedule(schedule) -> ScaleBySchedule:
    return ScaleBySchedule(schedule)


@dataclass
class Scale:
    factor: float


def scale(factor) -> Scale:
    return Scale(float(factor))


@dataclass
class ClipByGlobalNorm:
    """optax.clip_by_global_norm(max_norm)"""

    max_norm: float


def clip_by_global_norm(max_norm) -> ClipByGlobalNorm:
    return ClipByGlobalNorm(float(max_norm))


def constant_schedule(value) -> ExponentialDecay:
    """`schedule = lambda _: config.lr` (train_pm_vdvae.py:128-129)"""
    return ExponentialDecay(float(value), 1.0, 1.0)


@dataclass
class LinearSchedule:
    """optax.linear_schedule(init_value, end_value, transition_steps): the VDVAE warm-up (train_pm_vdvae.py:128-132)."""

    init_value: float
    end_value: float
    transition_steps: float

    def __call__(self, count: int) -> float:
        f = min(max(count / self.transition_steps, 0.0), 1.0)
        return (self.init_value - self.end_value) * (1.0 - f) + self.end_value


def linear_schedule(init_value, end_value, transition_steps, transition_begin=0) -> LinearSchedule:
    if transition_begin != 0:
        raise NotImplementedError("linear_schedule(transition_begin != 0) has no HIP path (the reference passes none)")
    if transition_steps <= 0:
        raise ValueError("transition_steps must be positive")
    return LinearSchedule(float(init_value), float(end_value), float(transition_steps))


@dataclass
class Chain:
    adam: ScaleByAdam
    decay: AddDecayedWeights
    schedule: Any                      # ExponentialDecay | LinearSchedule
    clip: Optional[ClipByGlobalNorm] = None

    def adam_cfg(self, grad_scale: float = 1.0):
        from ._lib import AdamCfg

        c = AdamCfg()
        c.b1, c.b2, c.eps = self.adam.b1, self.adam.b2, self.adam.eps
        c.weight_decay = self.decay.weight_decay
        if isinstance(self.schedule, LinearSchedule):
            c.lr_kind, c.lr_init, c.lr_end, c.lr_decay_rate = 1, self.schedule.init_value, self.schedule.end_value, 1.0
        else:
            c.lr_kind, c.lr_init, c.lr_decay_rate, c.lr_end = 0, self.schedule.init_value, self.schedule.decay_rate, 0.0
        c.lr_transition_steps = self.schedule.transition_steps
        c.grad_scale = grad_scale
        return c


def chain(*transforms) -> Chain:
    """optax.chain(scale_by_adam, [add_decayed_weights,] scale_by_schedule(exponential_decay), scale(-1)):
    the 4-transform form of train_pm_vae.py:74-83 or the 3-transform form of train_pm_vqvae.py:115-120."""
    if len(transforms) == 5 and isinstance(transforms[0], ClipByGlobalNorm):          # train_pm_vdvae.py:131-144
        c = chain(*transforms[1:])
        c.clip = transforms[0]
        return c
    if len(transforms) == 3 and isinstance(transforms[0], ScaleByAdam) and isinstance(transforms[1], ScaleBySchedule):
        transforms = (transforms[0], AddDecayedWeights(0.0), transforms[1], transforms[2])
    if (len(transforms) != 4 or not isinstance(transforms[0], ScaleByAdam)
            or not isinstance(transforms[1], AddDecayedWeights) or not isinstance(transforms[2], ScaleBySchedule)
            or not isinstance(transforms[2].schedule, (ExponentialDecay, LinearSchedule)) or not isinstance(transforms[3], Scale)
            or transforms[3].factor != -1.0):
        raise NotImplementedError(
            "only chain(scale_by_adam, add_decayed_weights, scale_by_schedule(exponential_decay), scale(-1.0)) "
            "(train_pm_vae.py:74-83) is lowered to the fused HIP optimizer")
    return Chain(transforms[0], transforms[1], transforms[2].schedule)


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0) -> Chain:
    """optax.adam(lr) = chain(scale_by_adam, scale(-lr)) (train_vqvae.py:82): the same fused kernel with
    a constant schedule and no weight decay."""
    if callable(learning_rate):
        if not isinstance(learning_rate, ExponentialDecay):
            raise NotImplementedError("optax.adam(schedule): only exponential_decay schedules are lowered")
        return Chain(scale_by_adam(b1, b2, eps, eps_root), AddDecayedWeights(0.0), learning_rate)
    return Chain(scale_by_adam(b1, b2, eps, eps_root), AddDecayedWeights(0.0),
                 ExponentialDecay(float(learning_rate), 1.0, 1.0))
