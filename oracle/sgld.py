"""Oracle (test infrastructure): Stochastic Gradient Langevin Dynamics step.

Follows ``Pyesian/optimizers/SGLD.py``:
  * ``:112-118`` ``_init_sgld_lr``: polynomial step-size schedule,
  * ``:54-57``   forward + loss,
  * ``:64-68``   per variable ``noise ~ N(0, std=lr(n))`` then
                 ``var += -lr(n) * (grad + noise)`` (so the injected noise has
                 standard deviation lr**2, exactly as written),
  * ``:70-87``   per layer running first and second moments
                 ``mean <- (mean*n + theta)/(n+1)``, ``sq_mean`` likewise,
  * ``:89-92``   the ``_dev`` matrix is dead state (never read by ``result()``)
                 and is NOT restated,
  * ``:94-95``   ``n += 1``; returns ``running_loss / n``,
  * ``:143-165`` ``result()``: ``Normal(loc=mean, scale=sq_mean - mean**2)``
                 (the variance is used as the scale, as written).
"""

from __future__ import annotations

import numpy as np

from .mlp import MLPSpec, loss_and_grad


def lr_schedule(nb_iterations: int, lr_upper: float, lr_lower: float, lr_gamma: float):
    """``SGLD._init_sgld_lr`` (SGLD.py:112-118): returns lr(step), float64."""
    n = nb_iterations
    l_g = np.power(lr_lower, 1.0 / lr_gamma)
    u_g = np.power(lr_upper, 1.0 / lr_gamma)
    b = -(n * l_g) / (l_g - u_g)
    a = lr_upper * np.power(b, lr_gamma)
    return lambda step: a * np.power((b + step), -lr_gamma)


class SGLDState:
    def __init__(self, theta0: np.ndarray, dtype=np.float64):
        self.dtype = dtype
        self.theta = np.asarray(theta0, dtype=dtype).copy()
        self.mean = np.zeros_like(self.theta)      # SGLD.py:97-110
        self.sq_mean = np.zeros_like(self.theta)
        self.n = 0
        self.running_loss = dtype(0)


def sgld_step(st: SGLDState, x, y, spec: MLPSpec, lr: float, unit_noise: np.ndarray):
    """One ``SGLD.step``.  ``unit_noise`` is N(0,1) of the flat parameter shape;
    the reference's ``tf.random.normal(stddev=lr)`` is ``lr * unit_noise``.
    Returns (batch loss, running_loss / n)."""
    dt = st.dtype
    lr = dt(lr)
    loss, g, _ = loss_and_grad(st.theta, x, y, spec, dt)
    st.running_loss = st.running_loss + loss                        # SGLD.py:58
    noise = lr * np.asarray(unit_noise, dtype=dt)                   # SGLD.py:67
    st.theta = st.theta + (-lr) * (g + noise)                       # SGLD.py:68
    n = dt(st.n)
    st.mean = (st.mean * n + st.theta) / (n + dt(1.0))              # SGLD.py:82-83
    st.sq_mean = (st.sq_mean * n + st.theta ** 2) / (n + dt(1.0))   # SGLD.py:86-87
    st.n += 1
    return loss, st.running_loss / st.n


def result_distribution(st: SGLDState):
    """(loc, scale) of the per-layer Normal built at SGLD.py:151-154."""
    return st.mean, st.sq_mean - st.mean ** 2
