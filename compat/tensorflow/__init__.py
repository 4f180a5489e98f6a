"""Stand-in for the handful of TensorFlow names the reference's driver scripts use.
NOT TensorFlow: it only builds Keras-2.15 model JSON and wraps NumPy arrays."""

import types

import numpy as np

from bayesian_inference_for_nn_amd import losses as _losses
from bayesian_inference_for_nn_amd.datasets import ArrayDataset as _ArrayDataset
from bayesian_inference_for_nn_amd.nn.model import Array as Tensor
from bayesian_inference_for_nn_amd.nn.model import DenseNet as _DenseNet
from bayesian_inference_for_nn_amd.nn.model import model_from_json as _model_from_json
from bayesian_inference_for_nn_amd.nn.model import sequential_json as _sequential_json

__version__ = "2.15.0-pyz-standin"
float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64


def _act_name(a):
    if a is None:
        return "linear"
    return a if isinstance(a, str) else getattr(a, "__name__", "linear")


class _Layer:
    def __init__(self, **kw):
        self.kw = kw


class Dense(_Layer):
    def __init__(self, units, activation=None, input_shape=None, **kw):
        super().__init__(**kw)
        self.units, self.activation, self.input_shape = int(units), _act_name(activation), input_shape


class Flatten(_Layer):
    def __init__(self, input_shape=None, **kw):
        super().__init__(**kw)
        self.input_shape = input_shape


class InputLayer(_Layer):
    def __init__(self, input_shape=None, **kw):
        super().__init__(**kw)
        self.input_shape = input_shape


class Sequential:
    """Collects layers and, once complete, behaves like the DenseNet built from its JSON."""

    def __init__(self, layers=None, name=None):
        self._layers = []
        self._net = None
        for l in layers or []:
            self.add(l)

    def add(self, layer):
        self._layers.append(layer)
        self._net = None

    def _build(self) -> _DenseNet:
        if self._net is None:
            shape, flatten, units, acts = None, False, [], []
            for l in self._layers:
                if getattr(l, "input_shape", None) is not None and shape is None:
                    shape = tuple(l.input_shape)
                if isinstance(l, Flatten):
                    flatten = True
                elif isinstance(l, Dense):
                    units.append(l.units)
                    acts.append(l.activation)
            if shape is None:
                raise ValueError("the first layer needs input_shape")
            self._net = _model_from_json(_sequential_json(shape, units, acts, flatten=flatten))
        return self._net

    def to_json(self):
        return self._build().to_json()

    def __getattr__(self, name):          # layers, get_weights, set_weights, predict, __call__ ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._build(), name)

    def __call__(self, x, training=False):
        return self._build()(x)


def _uniform(shape, minval=0, maxval=1, dtype=np.float32, seed=None):
    return Tensor(np.random.default_rng(seed).uniform(minval, maxval, size=shape).astype(dtype))


def _normal(shape, mean=0.0, stddev=1.0, dtype=np.float32, seed=None):
    return Tensor((mean + stddev * np.random.default_rng(seed).standard_normal(size=shape)).astype(dtype))


def _softmax(x, axis=-1):
    x = np.asarray(x)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return Tensor(e / e.sum(axis=axis, keepdims=True))


_softmax.__name__ = "softmax"


def _named(name):
    f = lambda x: x  # noqa: E731
    f.__name__ = name
    return f


def argmax(x, axis=None):
    return Tensor(np.argmax(np.asarray(x), axis=axis))


def reshape(x, shape):
    return Tensor(np.reshape(np.asarray(x), tuple(shape)))


def cast(x, dtype):
    return Tensor(np.asarray(x).astype(dtype))


def convert_to_tensor(x, dtype=None):
    return Tensor(np.asarray(x, dtype=dtype))


def constant(x, dtype=None):
    return Tensor(np.asarray(x, dtype=dtype))


def reduce_mean(x, axis=None):
    return Tensor(np.mean(np.asarray(x), axis=axis))


def _from_tensor_slices(tensors):
    x, y = tensors
    return _ArrayDataset(np.asarray(x), np.asarray(y))


def _to_categorical(y, num_classes=None):
    y = np.asarray(y).astype(int).reshape(-1)
    n = num_classes or int(y.max()) + 1
    return np.eye(n, dtype=np.float32)[y]


random = types.SimpleNamespace(uniform=_uniform, normal=_normal, set_seed=lambda s: None)
data = types.SimpleNamespace(Dataset=types.SimpleNamespace(from_tensor_slices=_from_tensor_slices), AUTOTUNE=-1)
keras = types.SimpleNamespace(
    Sequential=Sequential,
    Model=_DenseNet,
    models=types.SimpleNamespace(Sequential=Sequential, model_from_json=_model_from_json, Model=_DenseNet),
    layers=types.SimpleNamespace(Dense=Dense, Flatten=Flatten, InputLayer=InputLayer),
    losses=types.SimpleNamespace(SparseCategoricalCrossentropy=_losses.SparseCategoricalCrossentropy,
                                 MeanSquaredError=_losses.MeanSquaredError),
    activations=types.SimpleNamespace(softmax=_softmax, relu=_named("relu"), tanh=_named("tanh"),
                                      sigmoid=_named("sigmoid"), linear=_named("linear")),
    utils=types.SimpleNamespace(to_categorical=_to_categorical),
)
nn = types.SimpleNamespace(softmax=_softmax)
