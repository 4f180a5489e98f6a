"""Empirical weighted sample set -- the result distribution of HMC
(mirrors Pyesian/distributions/Sampled.py:8-60: cumulative frequencies, ``random.randint`` +
``bisect_left`` draw)."""

import bisect
import json
import os
import random

import numpy as np

from .Distribution import Distribution


class Sampled(Distribution):
    def __init__(self, samples, frequencies):
        if len(samples) == 0:
            raise ValueError("Can't have distribution Sampled with 0 samples")
        super().__init__(int(np.shape(samples[0])[0]))
        if len(samples) != len(frequencies):
            raise ValueError("Number of samples and list frequency do not have the same size")
        if len(np.shape(samples[0])) > 1:
            raise ValueError("Samples must have only one dimension")
        self._n_samples = len(samples)
        self._samples = [np.asarray(s, dtype=np.float32) for s in samples]
        self._frequencies = [int(f) for f in frequencies]
        self._acc_frequencies = []
        acc = 0
        for f in self._frequencies:
            acc += f
            if f == 0:
                raise ValueError("Samples frequencies can't sum up to zero")
            self._acc_frequencies.append(acc)

    def sample_index(self) -> int:
        w = random.randint(1, self._acc_frequencies[self._n_samples - 1])
        return bisect.bisect_left(self._acc_frequencies, w)

    def sample(self):
        return self._samples[self.sample_index()]

    def sample_n(self, n: int):
        return np.stack([self.sample() for _ in range(n)])

    def store(self, path: str):
        info = {"size": self._size, "n_samples": self._n_samples, "frequencies": self._frequencies,
                "dtypes": ["float32"] * self._n_samples}
        with open(os.path.join(path, "info.json"), "w") as f:
            f.write(json.dumps(info))
        sample_path = os.path.join(path, "samples")
        os.makedirs(sample_path, exist_ok=True)
        for i, s in enumerate(self._samples):
            np.save(os.path.join(sample_path, "sample" + str(i) + ".npy"), s)

    @classmethod
    def load(cls, path: str) -> "Distribution":
        with open(os.path.join(path, "info.json"), "r") as f:
            info = json.load(f)
        samples = [np.load(os.path.join(path, "samples", "sample" + str(i) + ".npy")) for i in range(info["n_samples"])]
        return Sampled(samples, info["frequencies"])
