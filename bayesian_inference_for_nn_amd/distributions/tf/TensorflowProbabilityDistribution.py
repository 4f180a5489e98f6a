"""Wrapper that turns a vector distribution into a result ``Distribution``
(mirrors Pyesian/distributions/tf/TensorflowProbabilityDistribution.py:9-58; the wrapped
object is one of distributions/tfd.py since TensorFlow-Probability is not a dependency)."""

import json
import os

import numpy as np

from .. import tfd
from ..Distribution import Distribution


class TensorflowProbabilityDistribution(Distribution):
    def __init__(self, tf_distribution):
        self._tf_distribution = tf_distribution
        if len(tf_distribution.batch_shape) + len(tf_distribution.event_shape) != 1:
            raise ValueError('The provided tensorflow distribution should be a vector')
        size = tf_distribution.event_shape[0] if len(tf_distribution.event_shape) >= 1 else tf_distribution.batch_shape[0]
        super().__init__(size)

    def sample(self):
        self._tf_distribution.sample()            # the reference draws twice and keeps the second (:55-58)
        return self._tf_distribution.sample()

    def sample_n(self, n: int):
        d = self._tf_distribution
        if isinstance(d, tfd.Deterministic):
            return np.repeat(d.loc[None, :], n, axis=0)
        return np.stack([self.sample() for _ in range(n)])

    def sample_n_device(self, n: int, out, col0: int) -> bool:
        """Device-side draws for Normal / Deterministic posteriors (True when done); other distributions
        are drawn on the host by the caller."""
        import torch
        d = self._tf_distribution
        if isinstance(d, tfd.Normal) and d.loc.ndim == 1:
            tfd.take_device_draws(n)                  # the reference draws twice and keeps the second (:55-58)
            d.sample_n_device(n, out, col0)
            return True
        if isinstance(d, tfd.Deterministic) and d.loc.ndim == 1:
            out[:, col0:col0 + d.loc.shape[0]] = torch.as_tensor(d.loc).to(out.device)
            return True
        return False

    def store(self, path: str):
        """distribution.json = {"type": <tfp class name>, "params": <its constructor parameters>}: the schema of the
        reference's BaseSerializer (distributions/tf/BaseSerializer.py:20-34), tfp's own bookkeeping parameters
        included, so that BaseSerializer.deserialize can rebuild the tfp object from a model stored here."""
        d = self._tf_distribution
        params = {"loc": np.asarray(d.loc).tolist()}
        if isinstance(d, tfd.Normal):
            params["scale"] = np.asarray(d.scale).tolist()
        elif isinstance(d, tfd.Deterministic):
            params.update(atol=None, rtol=None)
        params.update(validate_args=False, allow_nan_stats=True, name=type(d).__name__)
        with open(os.path.join(path, "distribution.json"), "w") as f:
            f.write(json.dumps({"type": type(d).__name__, "params": params}))

    @classmethod
    def load(cls, path: str) -> "Distribution":
        with open(os.path.join(path, "distribution.json"), "r") as f:
            data = json.load(f)
        params = data.get("params", data)            # (round 1 of this package wrote loc / scale at the top level)
        if data["type"] == "Normal":
            return cls(tfd.Normal(np.asarray(params["loc"], np.float32), np.asarray(params["scale"], np.float32)))
        if data["type"] == "Deterministic":
            return cls(tfd.Deterministic(np.asarray(params["loc"], np.float32)))
        raise ValueError("unknown distribution type " + str(data["type"]))
