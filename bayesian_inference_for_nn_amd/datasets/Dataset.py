"""Train / validation / test split of an in-memory data set + the loss factory
(mirrors the surface of Pyesian/datasets/Dataset.py:14-216 the optimizers use:
``training_dataset()``, ``loss()``, ``train_data`` / ``valid_data`` / ``test_data`` and their
sizes, ``likelihood_model``, feature / label normalisation).  Loaders that need the
network or TensorFlow (tfds names, UCI ids) are out of scope; arrays, ``(x, y)`` tuples,
objects exposing ``as_numpy()``, DataFrames and csv paths are accepted."""

from __future__ import annotations

import numpy as np

from ..nn.model import Array


class ArrayDataset:
    """The few ``tf.data.Dataset`` methods the reference's callers use, over two arrays."""

    def __init__(self, x, y):
        self.x, self.y = np.asarray(x), np.asarray(y)
        assert len(self.x) == len(self.y)

    def __len__(self):
        return len(self.x)

    def cardinality(self):
        return Array(np.int64(len(self.x)))

    def as_numpy(self):
        return self.x, self.y

    def batch(self, n):
        n = int(n.numpy() if hasattr(n, "numpy") else n)
        return _Batched(self, n)

    def shuffle(self, buffer_size=None, seed=None):
        perm = np.random.default_rng(seed).permutation(len(self.x))
        return ArrayDataset(self.x[perm], self.y[perm])

    def take(self, n):
        return ArrayDataset(self.x[:n], self.y[:n])

    def skip(self, n):
        return ArrayDataset(self.x[n:], self.y[n:])

    def cache(self):
        return self

    def prefetch(self, *_):
        return self

    def map(self, fn, num_parallel_calls=None):
        x, y = fn(self.x, self.y)
        return ArrayDataset(np.asarray(x), np.asarray(y))

    def __iter__(self):
        for i in range(len(self.x)):
            yield Array(self.x[i]), Array(self.y[i])


class _Batched:
    def __init__(self, ds, n):
        self.ds, self.n = ds, max(1, n)

    def __iter__(self):
        for o in range(0, len(self.ds), self.n):
            yield Array(self.ds.x[o:o + self.n]), Array(self.ds.y[o:o + self.n])


def _to_arrays(dataset, target_dim):
    if isinstance(dataset, ArrayDataset):
        return dataset.x, dataset.y
    if hasattr(dataset, "as_numpy"):
        return dataset.as_numpy()
    if isinstance(dataset, (tuple, list)) and len(dataset) == 2:
        x, y = dataset
        return np.asarray(x.numpy() if hasattr(x, "numpy") else x), np.asarray(y.numpy() if hasattr(y, "numpy") else y)
    try:
        import pandas as pd
        if isinstance(dataset, str):
            dataset = pd.read_csv(dataset)
        if isinstance(dataset, pd.DataFrame):
            return dataset.iloc[:, :-target_dim].values, dataset.iloc[:, -target_dim:].values
    except ImportError:
        pass
    raise ValueError("Unsupported dataset format")


class Dataset:
    def __init__(self, dataset, loss, likelihoodModel="Classification", load_images=False, target_dim=1,
                 feature_normalisation=False, label_normalisation=False, train_proportion=0.8,
                 test_proportion=0.1, valid_proportion=0.1, seed=None):
        if train_proportion + test_proportion + valid_proportion != 1:
            raise ValueError("Dataset split test_proportions must sum up to 1")
        self._train_proportion, self._test_proportion, self._valid_proportion = \
            train_proportion, test_proportion, valid_proportion
        self._loss = loss
        self.likelihood_model = likelihoodModel
        self.target_dim = target_dim
        self._label_mean = None
        self._label_std = None
        if isinstance(dataset, str) and not dataset.endswith(".csv"):
            if dataset == "mnist":
                # tfds.load('mnist') is a network fetch (Dataset.py:65): MNIST-shaped synthetic stand-in
                import warnings
                from .. import synth
                warnings.warn("Dataset('mnist'): tensorflow_datasets needs the network; substituting SYNTHETIC MNIST-shaped "
                              "data (uniform pixels, random labels) -- accuracies on it are meaningless", RuntimeWarning,
                              stacklevel=2)
                x, y = synth.mnist_like(60_000)
                dataset = (x.reshape(-1, 28, 28), y)
            else:
                raise ValueError("Unsupported dataset format (named tfds data sets need the network)")
        x, y = _to_arrays(dataset, target_dim)
        from .. import parallel
        if parallel.world_info()[1] > 1:        # one split for all ranks (sharded SVGD shares batches and validation rows)
            seed = parallel.shared_seed(seed)
        perm = np.random.default_rng(seed).permutation(len(x))       # Dataset.py:114
        x, y = x[perm], y[perm]
        self.size = len(x)
        self.train_size = int(self._train_proportion * self.size)     # Dataset.py:116-122
        self.test_size = int(self._test_proportion * self.size)
        self.valid_size = int(self._valid_proportion * self.size)
        self.train_data = ArrayDataset(x[:self.train_size], y[:self.train_size])
        rest_x, rest_y = x[self.train_size:], y[self.train_size:]
        self.test_data = ArrayDataset(rest_x[:self.test_size], rest_y[:self.test_size])
        self.valid_data = ArrayDataset(rest_x[self.test_size:], rest_y[self.test_size:])
        if feature_normalisation:
            self.feature_normalisation()
        if label_normalisation:
            self.label_normalisation()

    def training_dataset(self) -> ArrayDataset:
        return self.train_data

    def loss(self, reduction='auto'):
        return self._loss(reduction=reduction)

    def input_shape(self):
        return self.train_data.x.shape[1:]

    def _apply(self, fn):
        for name in ("train_data", "valid_data", "test_data"):
            ds = getattr(self, name)
            setattr(self, name, ArrayDataset(*fn(ds.x, ds.y)))

    def label_normalisation(self):
        if self.likelihood_model == "Regression":             # Dataset.py:178-194
            n = max(1, int(len(self.train_data) / 10))
            label = self.train_data.y[:n].astype(np.float64)
            self._label_mean, self._label_std = label.mean(), label.std()
            self._apply(lambda x, y: (x, (y - self._label_mean) / (self._label_std + 1e-8)))

    def feature_normalisation(self):
        if self.likelihood_model == "Regression":             # Dataset.py:196-209
            mean = self.train_data.x.mean(axis=0)
            std = self.train_data.x.astype(np.float64).std(axis=0)
            self._apply(lambda x, y: ((x.astype(np.float64) - mean) / (std + 1e-8), y.astype(np.float64)))
        else:                                                 # Dataset.py:210-216
            self._apply(lambda x, y: (x.astype(np.float32) / 255, y))
