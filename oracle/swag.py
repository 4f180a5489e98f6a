"""Oracle (test infrastructure): SWAG step and its low-rank-plus-diagonal posterior.

Follows ``Pyesian/optimizers/SWAG.py``:
  * ``:52-64``  forward, loss, ``var.assign_sub(lr * grad)`` (plain SGD update),
  * ``:66-92``  every ``frequency`` steps (``n % frequency == 0``), per layer:
                ``mean <- (mean*n + theta)/(n+1)``, ``sq_mean`` likewise -- ``n`` is the STEP count, not
                the number of moment updates, as written -- and the deviation matrix: a column
                ``theta - mean`` is appended until there are ``k`` columns; after that the first
                ``k-1`` columns are kept and the LAST column is replaced (``:84-89``, as written),
  * ``:93-94``  ``n += 1``; returns the batch loss,
  * ``:129-146`` ``result()``: ``MultivariateNormalDiagPlusLowRank(mean, sq_mean - mean**2,
                sqrt(scale/(k-1)) * dev)``.
and ``Pyesian/distributions/MultivariateNormalDiagPlusLowRank.py:31-41`` ``sample()``:
``mean + z1 + D z2 * sqrt(1/(2(k-1)))`` with ``z1 ~ N(0, scale=diag)`` (the variance is used as the
scale, as written), ``z2 ~ N(0, I_k)``, ``k`` = number of columns of D.
"""

from __future__ import annotations

import math

import numpy as np

from .mlp import MLPSpec, loss_and_grad


class SWAGState:
    def __init__(self, theta0: np.ndarray, k: int, dtype=np.float64):
        self.dtype, self.k = dtype, int(k)
        self.theta = np.asarray(theta0, dtype=dtype).copy()
        self.mean = np.zeros_like(self.theta)              # SWAG.py:113-127
        self.sq_mean = np.zeros_like(self.theta)
        self.dev = np.zeros((0, len(self.theta)), dtype=dtype)   # rows = the reference's columns
        self.n = 0


def swag_step(st: SWAGState, x, y, spec: MLPSpec, lr: float, frequency: int = 1):
    dt = st.dtype
    loss, g, _ = loss_and_grad(st.theta, x, y, spec, dt)
    st.theta = st.theta - dt(lr) * g                                    # SWAG.py:61-64
    if st.n % frequency == 0:                                           # SWAG.py:72
        n = dt(st.n)
        st.mean = (st.mean * n + st.theta) / (n + dt(1.0))              # SWAG.py:77-78
        st.sq_mean = (st.sq_mean * n + st.theta ** 2) / (n + dt(1.0))   # SWAG.py:81-82
        col = (st.theta - st.mean)[None, :]
        if st.dev.shape[0] == st.k:                                     # SWAG.py:85-89
            st.dev = np.concatenate([st.dev[:st.k - 1], col], axis=0)
        else:
            st.dev = np.concatenate([st.dev, col], axis=0)
    st.n += 1
    return loss


def result_distribution(st: SWAGState, scale: float):
    """(mean, diag, D) with D of shape (size, columns) -- SWAG.py:136-140."""
    return st.mean, st.sq_mean - st.mean ** 2, math.sqrt(scale / (st.k - 1)) * st.dev.T


def lowrank_sample(mean, diag, D, z1_unit, z2):
    """``MultivariateNormalDiagPlusLowRank.sample`` with injected N(0,1) draws z1_unit (size), z2 (k)."""
    k = D.shape[1]
    return mean + diag * z1_unit + (D @ z2) * math.sqrt(1 / (2 * (k - 1)))
