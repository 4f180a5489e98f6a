"""Oracle (test infrastructure): Stein variational gradient descent step.

Follows ``Pyesian/optimizers/SVGD.py``:
  * ``:143-157`` ``_init_particles``: (M, D) float64, row i = prior samples.
  * ``:183-202`` ``rbf_kernel``: ``K_ab = exp(-gamma * ||x_a - x_b||**2)``, gamma = 1,
                 on the float64 particle matrix.
  * ``:54-68``   ``_svgd_gradients``: ``grad_kernel = -1/2 d(sum K)/dX`` by autodiff,
                 which is ``2 gamma sum_j K_ij (x_i - x_j)`` for row i
                 (``literal_svgd_gradients`` below evaluates the (M,M,D) form and
                 a finite-difference derivative to confirm the identity);
                 ``phi = (K @ repeat(g_i, M) + grad_kernel) / M`` in float32,
                 row i kept (``:119``).
  * ``:100-129`` particle loop, **sequential** (Gauss-Seidel): particle i sees
                 the already-updated rows 0..i-1; the driving term is the *loss*
                 gradient (``:110-111``), the prior term ``lp``/``dlp`` is computed
                 and unused (``:108,112``); the update is a per-particle Keras
                 legacy Adam (``:120``); the updated float32 weights are stored
                 back as float64 (``:122-123``); a validation forward per
                 particle follows (``:126-129``).
  * ``:125,141`` returns ``sum_i loss_i / M``.
  * ``:165-181`` ``baseline__kernel`` (dead code in the reference): the median
                 heuristic, restated as ``median_kernel`` for the opt-in mode.
Keras legacy Adam (third-party, Appendix A3):
  ``m += (g-m)(1-b1); v += (g*g-v)(1-b2);
    theta -= lr*sqrt(1-b2**t)/(1-b1**t) * m/(sqrt(v)+eps)``, eps = 1e-7.
"""

from __future__ import annotations

import numpy as np

from .mlp import MLPSpec, forward, loss_and_grad, loss_value

BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-7


def rbf_row(particles64: np.ndarray, i: int, gamma: float = 1.0):
    """(K_i., rep_i): row i of the kernel and 2*gamma*sum_j K_ij (x_i - x_j), float64."""
    diff = particles64[i][None, :] - particles64            # (M, D)
    k = np.exp(-gamma * np.sum(diff * diff, axis=1))        # (M,)
    rep = 2.0 * gamma * (k[:, None] * diff).sum(axis=0)     # (D,)
    return k, rep


def literal_svgd_gradients(particles64: np.ndarray, g_i: np.ndarray, gamma: float = 1.0, fd_eps=1e-6):
    """The (M,M,D) formulation of SVGD.py:54-68 with the autodiff replaced by
    central finite differences of sum(K) -- small cases only."""
    M, D = particles64.shape

    def ksum(X):
        diff = X[:, None, :] - X[None, :, :]
        return np.exp(-gamma * np.sum(diff * diff, axis=-1))

    K = ksum(particles64)
    grad = np.zeros_like(particles64)
    for a in range(M):
        for d in range(D):
            Xp, Xm = particles64.copy(), particles64.copy()
            Xp[a, d] += fd_eps
            Xm[a, d] -= fd_eps
            grad[a, d] = (ksum(Xp).sum() - ksum(Xm).sum()) / (2 * fd_eps)
    grad_kernel = -grad / 2
    weighted = K @ np.repeat(g_i[None, :], M, axis=0)
    return (weighted + grad_kernel) / M


def adam_update(theta, g, m, v, t: int, lr: float, dtype):
    """Keras legacy Adam ``apply_gradients`` for one variable set; returns (theta, m, v)."""
    g = np.asarray(g, dtype=dtype)
    m = m + (g - m) * dtype(1 - BETA1)
    v = v + (g * g - v) * dtype(1 - BETA2)
    lr_t = dtype(lr) * np.sqrt(dtype(1) - dtype(BETA2) ** t) / (dtype(1) - dtype(BETA1) ** t)
    theta = theta - lr_t * m / (np.sqrt(v) + dtype(ADAM_EPS))
    return theta, m, v


class SVGDState:
    def __init__(self, particles0: np.ndarray, wdtype=np.float64):
        self.wdtype = wdtype
        self.particles = np.asarray(particles0, dtype=np.float64).copy()    # SVGD.py:144
        self.m = np.zeros(self.particles.shape, dtype=wdtype)
        self.v = np.zeros(self.particles.shape, dtype=wdtype)
        self.t = 0


def svgd_step(st: SVGDState, x, y, spec: MLPSpec, lr: float, gamma: float = 1.0,
              sweep: str = "gauss_seidel", x_val=None, y_val=None):
    """One ``SVGD.step``.  sweep = 'gauss_seidel' (the reference) or 'jacobi'
    (all particles updated from the same snapshot; the multi-GPU mode).
    gamma = None / 'median': the median-heuristic bandwidth (``median_kernel``).
    Returns dict(loss, val_loss, losses, phi): phi (M, D) is what each particle's Adam received."""
    M, _ = st.particles.shape
    dt = st.wdtype
    st.t += 1
    snapshot = st.particles.copy()
    total, total_val, losses, phis = 0.0, 0.0, [], []
    median = gamma is None or gamma == "median"
    if median:
        # the opt-in bandwidth: SVGD.baseline__kernel (SVGD.py:165-181) evaluated once on the snapshot, as in the
        # code it was taken from (all particles move from the same kernel matrix): Jacobi sweep only
        assert sweep == "jacobi", "the median-heuristic kernel is defined on a snapshot (Jacobi sweep)"
        K_med, dx_med, _ = median_kernel(snapshot)
    for i in range(M):
        theta_i = st.particles[i].astype(dt)                                # SVGD.py:101
        loss, g_i, _ = loss_and_grad(theta_i, x, y, spec, dt)               # SVGD.py:104-111
        src = st.particles if sweep == "gauss_seidel" else snapshot
        if median:
            k, rep = K_med[i], dx_med[i]                                     # SVGD.py:172-180, row i
        else:
            k, rep = rbf_row(src, i, gamma)                                  # SVGD.py:55-61
        k = k.astype(dt)
        phi = (k.sum() * g_i + rep.astype(dt)) / dt(M)                       # SVGD.py:64-68
        phis.append(phi)
        new_theta, st.m[i], st.v[i] = adam_update(theta_i, phi, st.m[i], st.v[i], st.t, lr, dt)
        st.particles[i] = new_theta.astype(np.float64)                       # SVGD.py:122-123
        total += loss / M                                                    # SVGD.py:125
        losses.append(loss)
        if x_val is not None:                                                # SVGD.py:126-129
            acts, logits = forward(new_theta, x_val, spec, dt)
            total_val += loss_value(acts[-1], logits, y_val, spec) / M
    return dict(loss=total, val_loss=total_val, losses=np.array(losses), phi=np.stack(phis))


def median_kernel(particles64: np.ndarray, h: float = -1):
    """``baseline__kernel`` (SVGD.py:165-181), dead code in the reference."""
    diff = particles64[:, None, :] - particles64[None, :, :]
    sq = np.sum(diff * diff, axis=-1)
    if h < 0:
        h = np.median(sq)
        h = np.sqrt(0.5 * h / np.log(particles64.shape[0] + 1))
    K = np.exp(-sq / h ** 2 / 2)
    dxkxy = -K @ particles64
    sumkxy = K.sum(axis=1)
    dxkxy = dxkxy + particles64 * sumkxy[:, None]
    return K, dxkxy / h ** 2, h
