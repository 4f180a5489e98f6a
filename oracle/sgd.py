"""Oracle (test infrastructure): plain mini-batch SGD step.

Follows ``Pyesian/optimizers/SGD.py:42-89``:
  * ``:56-58``  forward + loss on the batch,
  * ``:66-69``  ``var.assign_sub(lr * grad)`` per trainable variable,
  * ``:78-84``  ``_mean[layer] <- theta`` when ``n % frequency == 0`` (the running
                average is commented out in the reference: the "mean" is just the
                latest weights),
  * ``:85-87``  ``n += 1``; returns the epoch-running mean loss.
``result()`` (``SGD.py:131-146``) wraps each layer's ``_mean`` in a
``Deterministic`` distribution; the initial ``_mean`` is the starting weights
(``SGD.py:91-110``).
"""

from __future__ import annotations

import numpy as np

from .mlp import MLPSpec, loss_and_grad


class SGDState:
    def __init__(self, theta0: np.ndarray, dtype=np.float64):
        self.dtype = dtype
        self.theta = np.asarray(theta0, dtype=dtype).copy()
        self.mean = self.theta.copy()          # SGD.py:100-107 (init_val = weights)
        self.n = 0
        self.running_loss = dtype(0)
        self.seen_batches = 0


def sgd_step(st: SGDState, x, y, spec: MLPSpec, lr: float, frequency: int = 1, new_epoch: bool = False):
    """One ``SGD.step``; returns (batch loss, value the reference returns)."""
    st.seen_batches += 1                        # SGD.py:45
    if new_epoch:                               # SGD.py:48-54
        st.seen_batches = 1
        st.running_loss = st.dtype(0)
    loss, g, _ = loss_and_grad(st.theta, x, y, spec, st.dtype)
    st.running_loss = st.running_loss + loss    # SGD.py:60
    st.theta = st.theta - st.dtype(lr) * g      # SGD.py:66-69
    if st.n % frequency == 0:                   # SGD.py:78-84
        st.mean = st.theta.copy()
    st.n += 1
    return loss, st.running_loss / st.seen_batches
