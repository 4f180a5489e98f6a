"""Oracle (test infrastructure): eager torch-CPU restatement at the reference's
op granularity -- one autograd tape per step, per-variable updates, per-layer
moment updates (the structure of ``Pyesian/optimizers/SGLD.py:46-95``,
``BBB.py:128-211``, ``HMC.py:128-159``, ``SVGD.py:54-68``).

Two uses:
  1. an implementation of the same mathematics that is independent of
     ``oracle/mlp.py`` (autograd instead of hand-written reverse mode), used by
     tests/test_oracle_kat.py to cross-check every closed-form gradient;
  2. the ``cpu_baseline`` of ``bench.py`` (kind "port"): TensorFlow is not
     installed, so the reference's eager step is timed through this equivalent
     eager step on the host cores.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from .mlp import MLPSpec


def _act(z, name):
    if name == "linear":
        return z
    if name == "relu":
        return torch.relu(z)
    if name == "tanh":
        return torch.tanh(z)
    if name == "sigmoid":
        return torch.sigmoid(z)
    if name == "softmax":
        return torch.softmax(z, dim=-1)
    raise ValueError(name)


class EagerMLP:
    """A Keras-like container: ``variables`` is [kernel0, bias0, kernel1, ...]."""

    def __init__(self, spec: MLPSpec, theta: np.ndarray, dtype=torch.float32):
        self.spec, self.dtype = spec, dtype
        self.variables = []
        t = torch.as_tensor(np.asarray(theta), dtype=dtype)
        for (ko, bo), i, o in zip(spec.offsets(), spec.dims[:-1], spec.dims[1:]):
            self.variables.append(t[ko:ko + i * o].reshape(i, o).clone().requires_grad_(True))
            self.variables.append(t[bo:bo + o].clone().requires_grad_(True))

    def layers(self):
        return [(self.variables[2 * l], self.variables[2 * l + 1]) for l in range(self.spec.n_layers)]

    def __call__(self, x):
        h = x.reshape(len(x), -1)
        z = None
        for (w, b), a in zip(self.layers(), self.spec.acts):
            z = h @ w + b
            h = _act(z, a)
        return h, z

    def loss(self, x, y):
        out, z = self(x)
        if self.spec.loss == "scce":
            return torch.nn.functional.cross_entropy(z, y.reshape(-1).long(), reduction="mean")
        return ((out - y.reshape(out.shape).to(out.dtype)) ** 2).mean(dim=-1).mean()

    def flat(self) -> np.ndarray:
        return torch.cat([v.detach().reshape(-1) for v in self.variables]).numpy()


def flat_grad(spec: MLPSpec, theta, x, y, dtype=torch.float64):
    """(loss, flat gradient) via autograd."""
    m = EagerMLP(spec, theta, dtype)
    loss = m.loss(torch.as_tensor(x, dtype=dtype), torch.as_tensor(y))
    grads = torch.autograd.grad(loss, m.variables)
    return loss.item(), torch.cat([g.reshape(-1) for g in grads]).numpy()


class EagerSGLD:
    """``SGLD.step`` (SGLD.py:46-95) with torch eager ops, float32."""

    def __init__(self, spec: MLPSpec, theta0, lr_fn, generator=None):
        self.model = EagerMLP(spec, theta0, torch.float32)
        self.lr_fn = lr_fn
        self.n = 0
        self.running_loss = 0.0
        self.gen = generator
        self.mean, self.sq_mean = [], []
        for w, b in self.model.layers():
            size = w.numel() + b.numel()
            self.mean.append(torch.zeros(size, 1))
            self.sq_mean.append(torch.zeros(size, 1))

    def step(self, x, y, unit_noise=None):
        loss = self.model.loss(x, y)                                        # SGLD.py:54-57
        self.running_loss = self.running_loss + loss.detach()
        grads = torch.autograd.grad(loss, self.model.variables)             # SGLD.py:64
        lr = float(self.lr_fn(self.n))
        off = 0
        with torch.no_grad():
            for var, grad in zip(self.model.variables, grads):              # SGLD.py:65-68
                if unit_noise is None:
                    noise = torch.randn(grad.shape, generator=self.gen) * lr
                else:
                    noise = unit_noise[off:off + grad.numel()].reshape(grad.shape) * lr
                    off += grad.numel()
                var.add_(-lr * (grad + noise))
            for l, (w, b) in enumerate(self.model.layers()):                # SGLD.py:70-93
                theta = torch.cat([w.reshape(-1, 1), b.reshape(-1, 1)], 0)
                self.mean[l] = (self.mean[l] * self.n + theta) / (self.n + 1.0)
                self.sq_mean[l] = (self.sq_mean[l] * self.n + theta ** 2) / (self.n + 1.0)
        self.n += 1
        return self.running_loss / self.n


def bbb_autograd(spec: MLPSpec, mu, rho, eps, prior_mu, prior_rho, x, y, alpha, dtype=torch.float64):
    """Literal ``BBB._cost_function`` + the three tape.gradient calls
    (BBB.py:107-124,152-153,173).  Returns (cost, d/dmu, d/drho, d/dw) flat."""
    mu_t = torch.as_tensor(mu, dtype=dtype).clone().requires_grad_(True)
    rho_t = torch.as_tensor(rho, dtype=dtype).clone().requires_grad_(True)
    eps_t = torch.as_tensor(eps, dtype=dtype)
    pm = torch.as_tensor(np.broadcast_to(prior_mu, np.shape(mu)).copy(), dtype=dtype)
    pr = torch.as_tensor(np.broadcast_to(prior_rho, np.shape(mu)).copy(), dtype=dtype)
    w = (mu_t + torch.nn.functional.softplus(rho_t) * eps_t).detach().requires_grad_(True)   # assigned, not traced
    model = EagerMLP(spec, w.detach().numpy(), dtype)
    # rebuild the variables as views of w so d/dw is one flat gradient
    vars_, off = [], 0
    for i, o in zip(spec.dims[:-1], spec.dims[1:]):
        vars_.append(w[off:off + i * o].reshape(i, o)); off += i * o
        vars_.append(w[off:off + o]); off += o
    model.variables = vars_

    def logn(xv, m, r):
        s = torch.nn.functional.softplus(r)
        return (-0.5 * ((xv - m) / s) ** 2 - torch.log(s) - 0.5 * math.log(2 * math.pi)).sum()

    data = model.loss(torch.as_tensor(x, dtype=dtype), torch.as_tensor(y))
    cost = data + alpha * (logn(w, mu_t, rho_t) - logn(w, pm, pr))
    g_mu, g_rho, g_w = torch.autograd.grad(cost, [mu_t, rho_t, w], allow_unused=True)
    z = lambda g: np.zeros(len(mu)) if g is None else g.numpy()
    return cost.item(), z(g_mu), z(g_rho), z(g_w)


def hmc_potential_autograd(spec: MLPSpec, q, X, y, prior_mu, prior_sigma, n_train, dtype=torch.float64):
    """``HMC._potential_energy`` (HMC.py:149-159) and its gradient (``:128-136``)."""
    qt = torch.as_tensor(q, dtype=dtype).clone().requires_grad_(True)
    model = EagerMLP(spec, np.asarray(q), dtype)
    vars_, off = [], 0
    for i, o in zip(spec.dims[:-1], spec.dims[1:]):
        vars_.append(qt[off:off + i * o].reshape(i, o)); off += i * o
        vars_.append(qt[off:off + o]); off += o
    model.variables = vars_
    pm = torch.as_tensor(np.broadcast_to(prior_mu, np.shape(q)).copy(), dtype=dtype)
    ps = torch.as_tensor(np.broadcast_to(prior_sigma, np.shape(q)).copy(), dtype=dtype)
    logp = (-0.5 * ((qt - pm) / ps) ** 2 - torch.log(ps) - 0.5 * math.log(2 * math.pi)).sum()
    loss = model.loss(torch.as_tensor(X, dtype=dtype), torch.as_tensor(y))
    U = -logp + loss * n_train
    (g,) = torch.autograd.grad(U, [qt])
    return U.item(), loss.item(), g.numpy()


def svgd_phi_autograd(particles64: np.ndarray, i: int, g_i: np.ndarray, gamma: float = 1.0):
    """``SVGD._svgd_gradients`` (SVGD.py:54-68) with the (M,M,D) broadcast and
    autodiff, literally; returns row i of phi (float32 like the reference)."""
    X = torch.as_tensor(particles64, dtype=torch.float64).clone().requires_grad_(True)
    diff = X.unsqueeze(1) - X.unsqueeze(0)
    K = torch.exp(-gamma * (diff ** 2).sum(-1))
    (gk,) = torch.autograd.grad(K.sum(), [X])
    grad_kernel = (-gk / 2).float()
    Kf = K.detach().float()
    M = particles64.shape[0]
    G = torch.as_tensor(g_i, dtype=torch.float32).unsqueeze(0).repeat(M, 1)
    phi = (Kf @ G + grad_kernel) / M
    return phi[i].numpy()
