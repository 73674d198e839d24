"""Clustering accuracy and its validation callback (reference posterior_matching/clustering.py:14-72; host-side: a
confusion matrix over a few thousand labels and scipy's linear-sum assignment)."""
from __future__ import annotations

from typing import Any, Callable, Dict

import numpy as np

from .utils import Callback


def clustering_accuracy(y_true, y_pred) -> float:
    """The maximum accuracy over all assignments of clusters to class labels (clustering.py:19-37: confusion matrix ->
    linear sum assignment on max(cm) - cm)."""
    from scipy.optimize import linear_sum_assignment

    y_true, y_pred = np.asarray(y_true).astype(np.int64).ravel(), np.asarray(y_pred).astype(np.int64).ravel()
    labels = np.unique(np.concatenate([y_true, y_pred]))                  # sklearn.metrics.confusion_matrix's label set
    index = {int(v): i for i, v in enumerate(labels)}
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    np.add.at(cm, (np.fromiter((index[int(v)] for v in y_true), np.int64, len(y_true)),
                   np.fromiter((index[int(v)] for v in y_pred), np.int64, len(y_pred))), 1)
    rows, cols = linear_sum_assignment(-cm + cm.max())
    return float(cm[rows, cols].sum()) / float(cm.sum())


class ClusteringAccuracyCallback(Callback):
    """bax callback of clustering.py:40-72: collects pred_fn(batch) and batch["label"] at every validation step, logs
    `val_clustering_accuracy` at the end.  pred_fn(batch) -> integer cluster assignments [B] (device tensor or array)."""

    def __init__(self, pred_fn: Callable[[Dict[str, Any]], Any]):
        self._pred_fn = pred_fn
        self._preds, self._labels = [], []

    def on_validation_step(self, train_state, key, batch) -> None:
        if "label" not in batch:
            return
        preds = self._pred_fn(batch)
        self._preds.append(np.asarray(preds.detach().cpu() if hasattr(preds, "detach") else preds))
        lab = batch["label"]
        self._labels.append(np.asarray(lab.detach().cpu() if hasattr(lab, "detach") else lab))

    def on_validation_end(self, train_state, step: int, logs: Dict[str, Any]) -> None:
        if self._preds:
            logs["val_clustering_accuracy"] = clustering_accuracy(np.hstack(self._labels), np.hstack(self._preds))
        self._preds.clear()
        self._labels.clear()
