"""Seeded synthetic inputs of the BASELINE.json configurations (BASELINE.md section 3,
SURVEY.md 8d).  Pure NumPy; shared by bench.py, the examples and the tests."""

from __future__ import annotations

import numpy as np


def mnist_like(n: int = 48_000, seed: int = 1234):
    """X ~ U[0,1) (n, 784) float32, labels integers(0, 10) int32."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, 784), dtype=np.float32)
    y = rng.integers(0, 10, size=n).astype(np.int32)
    return x, y


def linreg(n: int = 600, seed: int = 7):
    """y = 2x + 2 on x in [1, 20) (simple_regression_example.py:11-12)."""
    rng = np.random.default_rng(seed)
    x = (1 + 19 * rng.random((n, 1))).astype(np.float32)
    return x, (2 * x + 2).astype(np.float32)


def moons(n: int = 2000, noise: float = 0.2, seed: int = 42):
    """Two interleaving half circles (the sklearn make_moons construction, HMC_classification.py:36)."""
    rng = np.random.default_rng(seed)
    n_out = n // 2
    n_in = n - n_out
    outer = np.stack([np.cos(np.linspace(0, np.pi, n_out)), np.sin(np.linspace(0, np.pi, n_out))], 1)
    inner = np.stack([1 - np.cos(np.linspace(0, np.pi, n_in)), 1 - np.sin(np.linspace(0, np.pi, n_in)) - 0.5], 1)
    x = np.concatenate([outer, inner]).astype(np.float64)
    y = np.concatenate([np.zeros(n_out, dtype=np.int32), np.ones(n_in, dtype=np.int32)])
    x += rng.normal(scale=noise, size=x.shape)
    perm = rng.permutation(n)
    return x[perm].astype(np.float32), y[perm]


def glorot_uniform(dims, seed: int = 99) -> np.ndarray:
    """Keras model_from_json default initialisation, flat order (kernel then bias per layer)."""
    rng = np.random.default_rng(seed)
    parts = []
    for i, o in zip(dims[:-1], dims[1:]):
        lim = np.sqrt(6.0 / (i + o))
        parts.append(rng.uniform(-lim, lim, size=(i, o)).astype(np.float32).reshape(-1))
        parts.append(np.zeros(o, dtype=np.float32))
    return np.concatenate(parts)


def batch_plan(n_rows: int, batch: int, n_steps: int, seed: int = 1236):
    """Row indices of `n_steps` consecutive batches of a shuffle-then-batch pipeline without
    drop_remainder (Optimizer.py:35-41): a fresh permutation per epoch (seed + epoch), the last
    batch of an epoch ragged.  Returns (idx int32 (n_steps, batch), sizes list)."""
    idx = np.zeros((n_steps, batch), dtype=np.int32)
    sizes, s, epoch = [], 0, 0
    while s < n_steps:
        perm = np.random.default_rng(seed + epoch).permutation(n_rows)
        for o in range(0, n_rows, batch):
            if s == n_steps:
                break
            chunk = perm[o:o + batch]
            idx[s, :len(chunk)] = chunk
            sizes.append(len(chunk))
            s += 1
        epoch += 1
    return idx, sizes


def sgld_lr_table(n_total: int, lr_upper: float, lr_lower: float, lr_gamma: float, start: int, count: int):
    """lr(step) = a (b + step)^-gamma with lr(0) = lr_upper, lr(n_total) = lr_lower (SGLD.py:112-118)."""
    l_g = np.power(lr_lower, 1.0 / lr_gamma)
    u_g = np.power(lr_upper, 1.0 / lr_gamma)
    b = -(n_total * l_g) / (l_g - u_g)
    a = lr_upper * np.power(b, lr_gamma)
    steps = np.arange(start, start + count, dtype=np.float64)
    return (a * np.power(b + steps, -lr_gamma)).astype(np.float32)
