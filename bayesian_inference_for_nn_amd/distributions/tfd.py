"""Minimal stand-ins for the two tensorflow_probability distributions the hot path uses
(``tfp.distributions.Normal`` at GaussianPrior.py:42-43, SGLD.py:151-154, BBB.py:310-313;
``Deterministic`` at SGD.py:136).  Semantics per SURVEY.md Appendix A2: ``log_prob`` does
not validate the scale (negative scale -> NaN)."""

from __future__ import annotations

import math

import numpy as np

_rng = np.random.default_rng()
# device draws (BayesianModel.predict): Philox seed and the next unused draw counter
_dev_seed = int(np.random.default_rng().integers(1, 2 ** 62))
_dev_draw = 0


def seed(s):
    """Seeds the host generator used by ``sample()`` and the device draws of ``predict`` (the reference
    never seeds TF)."""
    global _rng, _dev_seed, _dev_draw
    _rng = np.random.default_rng(s)
    _dev_seed = int(np.random.default_rng(s).integers(1, 2 ** 62))
    _dev_draw = 0


def take_device_draws(n: int) -> int:
    """Reserves n draw counters of the device stream; returns the first."""
    global _dev_draw
    first = _dev_draw
    _dev_draw += int(n)
    return first


class Normal:
    def __init__(self, loc, scale):
        self.loc = np.asarray(loc, dtype=np.float32)
        self.scale = np.broadcast_to(np.asarray(scale, dtype=np.float32), self.loc.shape).copy()
        self.batch_shape = tuple(self.loc.shape)
        self.event_shape = ()

    def mean(self):
        return self.loc

    def stddev(self):
        return self.scale

    def sample(self):
        return (self.loc + self.scale * _rng.standard_normal(self.loc.shape, dtype=np.float32)).astype(np.float32)

    def sample_n_device(self, n: int, out, col0: int):
        """n draws into columns [col0, col0 + size) of the CUDA matrix `out`, generated on the device."""
        import torch
        from .. import engine
        if getattr(self, "_dev", None) is None:
            self._dev = (torch.as_tensor(self.loc).cuda(), torch.as_tensor(self.scale).cuda())
        engine.sample_normal_rows(out, col0, self._dev[0], self._dev[1], _dev_seed, take_device_draws(n))

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            return (-0.5 * ((x - self.loc) / self.scale) ** 2 - np.log(self.scale)
                    - np.float32(0.5 * math.log(2 * math.pi)))

    def batch_shape_tensor(self):
        return np.asarray(self.batch_shape)


class Deterministic:
    def __init__(self, loc):
        self.loc = np.asarray(loc, dtype=np.float32)
        self.batch_shape = tuple(self.loc.shape)
        self.event_shape = ()

    def mean(self):
        return self.loc

    def stddev(self):
        return np.zeros_like(self.loc)

    def sample(self):
        return self.loc.copy()
