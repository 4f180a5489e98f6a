"""Oracle (test infrastructure): Hamiltonian Monte Carlo proposal + Metropolis test.

Follows ``Pyesian/optimizers/HMC.py``:
  * ``:149-159`` ``_potential_energy``: ``U = -sum log N(w; mu_p, sigma_p)
                 + loss(y, model(X)) * N_train``; the prior scale is the raw
                 ``rho`` of ``GaussianPrior`` (no softplus).  With a negative
                 ``rho`` the log is NaN (tfp ``Normal.log_prob`` does not
                 validate its scale, Appendix A2) and every non-burn proposal is
                 rejected; the gradient stays finite because sigma enters squared.
  * ``:161-166`` ``_kinetic_energy``: ``K = sum p**2 / (2 m)``.
  * ``:168-171`` ``_sample_kinetic_energy``: ``p ~ N(0, std = m)``.
  * ``:128-141`` ``_step_p`` / ``_step_q``: ``p -= s * dU/dq``; ``q += s * p / m``.
  * ``:74-104``  ``step``: half kick, L x (drift, full kick), half kick (the
                 guard ``i != L`` is always true, so the total kick is (L+1) eps),
                 accept iff ``burning or u < exp(K0 + U0 - K1 - U1)``.
  * ``:106-126`` ``train``: 10 burn-in steps (forced accept, not recorded),
                 then the sampling steps; ``:176-187`` ``result``.
"""

from __future__ import annotations

import math

import numpy as np

from .mlp import MLPSpec, loss_and_grad

_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def normal_log_prob(x, mu, sigma):
    """tfp ``Normal(loc, scale).log_prob`` without scale validation."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return -0.5 * ((x - mu) / sigma) ** 2 - np.log(sigma) - _LOG_SQRT_2PI


def potential_energy(q, X, y, spec: MLPSpec, prior_mu, prior_sigma, n_train: int, dtype=np.float64):
    """(U, mean loss, dU/dq)  --  HMC.py:149-159 and the gradient taken at :128-136."""
    q = np.asarray(q, dtype=dtype)
    loss, g, _ = loss_and_grad(q, X, y, spec, dtype)
    mu = np.broadcast_to(np.asarray(prior_mu, dtype=dtype), q.shape)
    sg = np.broadcast_to(np.asarray(prior_sigma, dtype=dtype), q.shape)
    U = -normal_log_prob(q, mu, sg).sum() + loss * n_train
    dU = (q - mu) / (sg * sg) + g * n_train
    return U, loss, dU


def kinetic_energy(p, m):
    return (1.0 / (2.0 * m)) * np.sum(np.square(p))


def hmc_step(q, unit_p, X, y, spec: MLPSpec, prior_mu, prior_sigma, L: int, epsilon: float, m: float,
             u: float, burning: bool = False, n_train: int | None = None, dtype=np.float64):
    """One ``HMC.step``.  ``unit_p`` ~ N(0,1) (the reference's momentum is
    ``m * unit_p``); ``u`` is the host uniform of ``random.random()``.

    Returns dict(q, accepted, loss, U0, K0, U1, K1, log_ratio)."""
    q0 = np.asarray(q, dtype=dtype).copy()
    n_train = len(X) if n_train is None else n_train
    p = dtype(m) * np.asarray(unit_p, dtype=dtype)                        # HMC.py:171
    K0 = kinetic_energy(p, m)                                             # HMC.py:79
    U0, loss0, g = potential_energy(q0, X, y, spec, prior_mu, prior_sigma, n_train, dtype)
    q = q0.copy()
    p = p - (epsilon / 2) * g                                             # HMC.py:82
    for _ in range(int(L)):                                               # HMC.py:83-86
        q = q + (epsilon / m) * p
        _, _, g = potential_energy(q, X, y, spec, prior_mu, prior_sigma, n_train, dtype)
        p = p - epsilon * g
    _, _, g = potential_energy(q, X, y, spec, prior_mu, prior_sigma, n_train, dtype)
    p = p - (epsilon / 2) * g                                             # HMC.py:87
    K1 = kinetic_energy(p, m)
    U1, loss1, _ = potential_energy(q, X, y, spec, prior_mu, prior_sigma, n_train, dtype)
    log_ratio = K0 + U0 - K1 - U1                                         # HMC.py:91
    with np.errstate(over="ignore", invalid="ignore"):
        ratio = np.exp(log_ratio)
    accepted = bool(burning or (u < ratio))                               # NaN compares False
    return dict(q=q if accepted else q0, accepted=accepted, loss=loss1 if accepted else loss0,
                U0=U0, K0=K0, U1=U1, K1=K1, log_ratio=log_ratio, q_proposed=q, p_final=p)


class HMCChain:
    """Sample bookkeeping of ``HMC.step``/``train``/``result`` (HMC.py:75-77,92-104,176-187)."""

    def __init__(self, q0):
        self.q = np.asarray(q0).copy()
        self.samples, self.frequency = [], []

    def record(self, res, sampling=True):
        if sampling and not self.frequency:      # HMC.py:75-77 (records the *starting* q)
            self.frequency.append(1)
            self.samples.append(self.q.copy())
        self.q = res["q"]
        if sampling:
            if res["accepted"]:
                self.frequency.append(1)
                self.samples.append(self.q.copy())
            else:
                self.frequency[-1] += 1
