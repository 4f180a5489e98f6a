"""SWAG's posterior: diagonal + low-rank Gaussian (mirrors
Pyesian/distributions/MultivariateNormalDiagPlusLowRank.py:11-41): ``sample = mean + z1 + D z2 *
sqrt(1/(2(k-1)))`` with ``z1 ~ N(0, scale=diag)`` (diag is used as the scale, as written) and
``z2 ~ N(0, I_k)``."""

import json
import os
from math import sqrt

import numpy as np

from . import tfd
from .Distribution import Distribution


class MultivariateNormalDiagPlusLowRank(Distribution):
    def __init__(self, mean, diag, D):
        mean = np.asarray(mean, dtype=np.float32)
        super().__init__(mean.shape[0])
        self._mean = mean
        self._D = np.asarray(D, dtype=np.float32).reshape(mean.shape[0], -1)
        self._diag = np.asarray(diag, dtype=np.float32)

    def sample_n(self, n: int):
        k = self._D.shape[1]
        z1 = tfd._rng.standard_normal((n, self._size), dtype=np.float32) * self._diag
        z2 = tfd._rng.standard_normal((n, k), dtype=np.float32)
        return (self._mean[None, :] + z1 + (z2 @ self._D.T) * np.float32(sqrt(1 / (2 * (k - 1))))).astype(np.float32)

    def sample(self):
        return self.sample_n(1)[0]

    def store(self, path: str):
        data = json.dumps({"mean": self._mean.tolist(), "D": self._D.tolist(), "diag": self._diag.tolist()})
        with open(os.path.join(path, "distribution.json"), "w") as f:
            f.write(data)

    @classmethod
    def load(cls, path: str) -> "Distribution":
        with open(os.path.join(path, "distribution.json"), "r") as f:
            d = json.load(f)
        return MultivariateNormalDiagPlusLowRank(np.asarray(d["mean"]), np.asarray(d["diag"]), np.asarray(d["D"]))
