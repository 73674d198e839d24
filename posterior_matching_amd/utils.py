"""Host-side helpers with the names of the reference's posterior_matching/utils.py."""
from __future__ import annotations

import json
import os
from datetime import datetime
from typing import Any, Callable, Dict, Optional

from .data import load_datasets  # noqa: F401  (reference utils.py:36-121; synthetic / .npy data here)


def configure_environment() -> None:
    """reference utils.py:21-24 hides GPUs from TF and stops XLA pre-allocation; nothing to do here
    beyond keeping the dmabuf IPC mode RCCL needs on this pool."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def make_run_dir(path: str = "runs", prefix: Optional[str] = None) -> str:
    """reference utils.py:27-33: runs/<prefix>-YYYYmmdd-HHMMSS"""
    run_id = datetime.now().strftime("%Y%m%d-%H%M%S")
    if prefix is not None:
        run_id = prefix + "-" + run_id
    run_dir = os.path.join(path, run_id)
    os.makedirs(run_dir, exist_ok=True)
    return run_dir


def cyclical_annealing_schedule(low_value: float, high_value: float, period: int, delay: int = 0) -> Callable[[int], float]:
    """reference utils.py:124-136 (host copy for logging; the training step evaluates the same
    formula on the device, csrc/pm_optim.hip beta_from_step)."""

    def schedule(count: int) -> float:
        true_count = count
        count = min(max((count - delay) % period, 0), period // 2)
        frac = 1 - count / (period // 2)
        x = (low_value - high_value) * frac + high_value
        return x * float(true_count >= delay)

    return schedule


class Callback:
    def on_validation_step(self, train_state, key, batch) -> None:
        pass

    def on_validation_end(self, train_state, step: int, logs: Dict[str, Any]) -> None:
        pass


class JSONLCallback(Callback):
    """Scalar logger in place of the reference's TensorBoardCallback (utils.py:139-151;
    tensorflow is not available): one JSON object per validation into <path>/scalars.jsonl."""

    def __init__(self, path: str):
        os.makedirs(path, exist_ok=True)
        self._file = os.path.join(path, "scalars.jsonl")

    def on_validation_end(self, train_state, step: int, logs: Dict[str, Any]) -> None:
        import numpy as np

        scalars = {}
        for k, v in logs.items():
            if getattr(v, "ndim", 0) > 0:            # image summaries (train_vqvae.py's reconstructions)
                np.save(os.path.join(os.path.dirname(self._file), f"{k}_{step}.npy"), np.asarray(v))
            else:
                scalars[k] = float(v)
        with open(self._file, "a") as fp:
            fp.write(json.dumps({"step": step, **scalars}) + "\n")


TensorBoardCallback = JSONLCallback


class steady_state_gc:
    """Context manager for the steady-state part of a training loop: the cyclic garbage collector is paused (after one full
    collection; everything alive then is moved out of its reach with gc.freeze) - a generation-2 pass in the middle of a step
    stalls the host long enough for the launch queue to drain (a 1.33 ms PM-VAE step shows up as 1.5 - 1.7 ms once in a few
    dozen steps).  Reference counting still frees everything acyclic; `collect()` runs a collection at a safe point
    (Trainer.fit: at validation time).  PM_NO_GC_FREEZE=1: leave the collector alone (A/B)."""

    def __enter__(self):
        import gc
        import os

        self._on = not os.environ.get("PM_NO_GC_FREEZE") and gc.isenabled()
        if self._on:
            gc.collect()
            gc.freeze()
            gc.disable()
        return self

    def collect(self) -> None:
        if self._on:
            import gc

            gc.collect()

    def __exit__(self, *exc):
        if self._on:
            import gc

            gc.enable()
            gc.unfreeze()
        return False
