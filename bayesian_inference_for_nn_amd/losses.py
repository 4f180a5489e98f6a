"""Stand-ins for the two ``tf.keras.losses`` classes the reference passes to ``Dataset``
(as classes; ``Dataset.loss(reduction)`` instantiates them -- Pyesian/datasets/Dataset.py:152-159).
A real tf.keras loss class works too: the optimizers dispatch on ``__name__``."""

import numpy as np


class _Loss:
    kind = None

    def __init__(self, reduction="auto", **kwargs):
        self.reduction = reduction


class SparseCategoricalCrossentropy(_Loss):
    kind = "scce"

    def __call__(self, y_true, y_pred):
        p = np.asarray(y_pred, dtype=np.float64)
        y = np.asarray(y_true).reshape(-1).astype(np.int64)
        p = np.clip(p / p.sum(axis=-1, keepdims=True), 1e-7, 1 - 1e-7)
        return float(-np.log(p[np.arange(len(y)), y]).mean())


class MeanSquaredError(_Loss):
    kind = "mse"

    def __call__(self, y_true, y_pred):
        p = np.asarray(y_pred, dtype=np.float64)
        y = np.asarray(y_true, dtype=np.float64).reshape(p.shape)
        return float(((p - y) ** 2).mean(axis=-1).mean())


def loss_kind(loss_cls) -> str:
    """'scce' | 'mse' from a loss class or instance (ours or tf.keras')."""
    name = getattr(loss_cls, "__name__", type(loss_cls).__name__)
    if name == "SparseCategoricalCrossentropy":
        return "scce"
    if name == "MeanSquaredError":
        return "mse"
    raise ValueError(f"unsupported loss '{name}' (supported: SparseCategoricalCrossentropy, MeanSquaredError)")
