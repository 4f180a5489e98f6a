"""Oracle (test infrastructure): Bayes-by-Backprop step.

Follows ``Pyesian/optimizers/BBB.py``:
  * ``:250-275`` ``compile_extra_components``: prior mixing for scalar priors
                 (``mix_prior`` below), posterior initialisation ``mu <- prior
                 mean``, ``rho <- raw prior rho`` (``:277-296``).
  * ``:218-246`` ``_update_weights``: ``eps ~ N(0,1)``; ``w = mu + softplus(rho) * eps``.
  * ``:51-124``  cost = loss + alpha * ( sum log N(w; mu, softplus rho)
                                        - sum log N(w; mu_p, softplus rho_p) ).
  * ``:152-201`` gradients and update.  The tape watches mu and rho, the model
                 variables hold the *assigned* w, so
                   d cost/d mu  = alpha * (w - mu) / sigma**2
                   d cost/d rho = alpha * (-1/sigma + (w-mu)**2 / sigma**3) * sigmoid(rho)
                   d cost/d w   = d loss/d w + alpha * (-(w-mu)/sigma**2 + (w-mu_p)/sigma_p**2)
                 and  mu  <- mu  - lr * (d mu + d w)
                      rho <- rho - lr * (eps * sigmoid(rho) * d w + d rho).
  * ``:203-209`` when ``step % 10 != 0``: validation loss with the sampled w.
  * ``:300-323`` ``result()``: per layer ``Normal(mu, softplus(rho))``.
"""

from __future__ import annotations

import math

import numpy as np

from .mlp import MLPSpec, forward, loss_and_grad, loss_value

_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def softplus(x):
    return np.logaddexp(0.0, x)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def mix_prior(mean1, rho1, mean2=0.0, rho2=0.0, pi=1.0):
    """BBB.py:258-270 (raises ZeroDivisionError for rho1 == 0, as the reference)."""
    sign = rho1 / abs(rho1)
    return (mean1 * pi + mean2 * (1 - pi),
            sign * math.sqrt((rho1 * pi) ** 2 + (rho2 * (1 - pi)) ** 2))


def gaussian_log_prob_sum(w, mean, rho):
    """``_guassian_likelihood`` (BBB.py:51-64): scale = softplus(rho)."""
    s = softplus(rho)
    return np.sum(-0.5 * ((w - mean) / s) ** 2 - np.log(s) - _LOG_SQRT_2PI)


def sample_weights(mu, rho, eps):
    return mu + softplus(rho) * eps                                   # BBB.py:241-242


def cost_function(w, mu, rho, prior_mu, prior_rho, x, y, spec: MLPSpec, alpha, dtype=np.float64):
    """Literal ``_cost_function`` (BBB.py:107-124)."""
    acts, logits = forward(w, x, spec, dtype)
    data = loss_value(acts[-1], logits, y, spec)
    kl = gaussian_log_prob_sum(w, mu, rho) - gaussian_log_prob_sum(w, prior_mu, prior_rho)
    return data + alpha * kl


def bbb_step(mu, rho, eps, x, y, spec: MLPSpec, lr, alpha, prior_mu, prior_rho, dtype=np.float64):
    """One ``BBB.step`` without the validation pass.
    Returns dict(mu, rho, w, cost, loss, kl)."""
    mu = np.asarray(mu, dtype=dtype)
    rho = np.asarray(rho, dtype=dtype)
    eps = np.asarray(eps, dtype=dtype)
    pm = np.broadcast_to(np.asarray(prior_mu, dtype=dtype), mu.shape)
    pr = np.broadcast_to(np.asarray(prior_rho, dtype=dtype), mu.shape)
    sigma = softplus(rho)
    sigma_p = softplus(pr)
    w = mu + sigma * eps                                             # BBB.py:241-245
    loss, g_loss, _ = loss_and_grad(w, x, y, spec, dtype)
    kl = gaussian_log_prob_sum(w, mu, rho) - gaussian_log_prob_sum(w, pm, pr)
    cost = loss + alpha * kl                                         # BBB.py:121-124
    d = w - mu
    g_mu = alpha * d / sigma ** 2                                    # BBB.py:152
    g_rho = alpha * (-1.0 / sigma + d ** 2 / sigma ** 3) * sigmoid(rho)   # BBB.py:153
    g_w = g_loss + alpha * (-d / sigma ** 2 + (w - pm) / sigma_p ** 2)    # BBB.py:173
    new_mu = mu - lr * (g_mu + g_w)                                  # BBB.py:176-179
    sd_grad = eps / (1.0 + np.exp(-rho)) * g_w + g_rho               # BBB.py:185-187
    new_rho = rho - lr * sd_grad                                     # BBB.py:189-191
    return dict(mu=new_mu, rho=new_rho, w=w, cost=cost, loss=loss, kl=kl)


def validation_loss(w, x_val, y_val, spec: MLPSpec, dtype=np.float64):
    """BBB.py:203-209 (full validation split forwarded with the sampled w)."""
    acts, logits = forward(w, x_val, spec, dtype)
    return loss_value(acts[-1], logits, y_val, spec)


def result_distribution(mu, rho):
    """(loc, scale) per BBB.py:310-313."""
    return mu, softplus(rho)
