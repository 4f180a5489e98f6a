"""Base class of the result distributions (mirrors Pyesian/distributions/Distribution.py:6-50)."""

from abc import ABC, abstractmethod


class Distribution(ABC):
    def __init__(self, size: int):
        self._size = size

    @property
    def size(self) -> int:
        return self._size

    @abstractmethod
    def sample(self):
        """one flat weight vector"""

    def sample_n(self, n: int):
        """n draws stacked as (n, size) float32 -- used by the batched predict read-out."""
        import numpy as np
        return np.stack([np.asarray(self.sample(), dtype=np.float32) for _ in range(n)])

    @abstractmethod
    def store(self, path: str):
        pass

    @classmethod
    @abstractmethod
    def load(cls, path: str) -> "Distribution":
        pass
