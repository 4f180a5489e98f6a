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
        """info.json + samples/sample<i>.tf, each sample a serialized TensorProto: what tf.io.serialize_tensor /
        tf.io.write_file leave there in the reference (Sampled.py:34-48), so either side loads the other's model."""
        from .tensorproto import serialize_tensor
        info = {"size": self._size, "n_samples": self._n_samples, "frequencies": self._frequencies,
                "dtypes": [s.dtype.name for s in self._samples]}
        with open(os.path.join(path, "info.json"), "w") as f:
            f.write(json.dumps(info))
        sample_path = os.path.join(path, "samples")
        os.makedirs(sample_path, exist_ok=True)
        for i, s in enumerate(self._samples):
            with open(os.path.join(sample_path, "sample" + str(i) + ".tf"), "wb") as f:
                f.write(serialize_tensor(s))

    @classmethod
    def load(cls, path: str) -> "Distribution":
        from .tensorproto import parse_tensor
        with open(os.path.join(path, "info.json"), "r") as f:
            info = json.load(f)
        sample_dir = os.path.join(path, "samples")
        samples = []
        for i in range(info["n_samples"]):
            tf_file = os.path.join(sample_dir, "sample" + str(i) + ".tf")
            if os.path.exists(tf_file):                                    # Sampled.py:50-60
                with open(tf_file, "rb") as f:
                    samples.append(parse_tensor(f.read(), info["dtypes"][i]))
            else:                                                          # models stored by round 1 of this package
                samples.append(np.load(os.path.join(sample_dir, "sample" + str(i) + ".npy")))
        return Sampled(samples, info["frequencies"])
