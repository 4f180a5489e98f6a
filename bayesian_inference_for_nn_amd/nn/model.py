"""Keras-2.15 Sequential JSON -> Dense stack, and a small Keras-like model object.

The reference hands its model to every optimizer as ``model.to_json()``
(``Pyesian/optimizers/Optimizer.py:43`` ``model_config: str``; format fixture
``static/models/sl/dense1.json``) and rebuilds it with
``tf.keras.models.model_from_json`` (``SGLD.py:137``, ``HMC.py:56``, ``BBB.py:256``,
``SVGD.py:222``, ``nn/BayesianModel.py:18``).  TensorFlow is not part of this backend,
so the JSON is parsed here: ``Sequential`` of ``InputLayer`` / ``Flatten`` / ``Dense``.
"""

from __future__ import annotations

import json
from typing import List, Optional, Sequence

import numpy as np

from ..engine import MLPSpec

_ACTS = {"linear", "relu", "tanh", "sigmoid", "softmax"}


class Array(np.ndarray):
    """ndarray with the ``.numpy()`` accessor the reference scripts call on TF tensors."""

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def numpy(self):
        return np.asarray(self)


class Layer:
    def __init__(self, class_name: str, config: dict, in_dim: Optional[int], out_dim: Optional[int]):
        self.class_name, self.config = class_name, config
        self.name = config.get("name", class_name.lower())
        self.in_dim, self.out_dim = in_dim, out_dim
        self._model = None
        self._dense_index = None        # index among the Dense layers

    @property
    def trainable_variables(self) -> List[np.ndarray]:
        """[kernel (in, out), bias (out)] views of the model's flat weight vector."""
        if self._dense_index is None:
            return []
        sl = self._model.spec.layer_slices()[self._dense_index]
        flat = self._model.weights_flat[sl]
        i, o = self.in_dim, self.out_dim
        return [flat[: i * o].reshape(i, o), flat[i * o:]]

    @property
    def trainable_weights(self):
        return self.trainable_variables


def _activation_name(a) -> str:
    if isinstance(a, dict):          # Keras may serialise a function object as {"class_name": ...}
        a = a.get("config", a.get("class_name", "linear"))
    a = str(a)
    if a not in _ACTS:
        raise ValueError(f"unsupported activation '{a}' (supported: {sorted(_ACTS)})")
    return a


class DenseNet:
    """The subset of ``tf.keras.Model`` the hot path and its callers touch: ``layers``,
    ``trainable_variables``, ``get_weights`` / ``set_weights``, ``to_json``, ``__call__`` /
    ``predict`` (evaluated by the HIP kernels)."""

    def __init__(self, model_config: str):
        cfg = json.loads(model_config) if isinstance(model_config, str) else model_config
        if cfg.get("class_name") != "Sequential":
            raise ValueError("only Keras Sequential models of InputLayer / Flatten / Dense are supported")
        self._config = cfg
        self.layers: List[Layer] = []
        dims, acts = [], []
        cur = None
        for lc in cfg["config"]["layers"]:
            cn, c = lc["class_name"], lc.get("config", {})
            shape = c.get("batch_input_shape") or (lc.get("build_config", {}) or {}).get("input_shape")
            if cur is None and shape is not None:
                cur = int(np.prod([d for d in shape[1:]]))
                self.input_shape = tuple(shape[1:])
            if cn == "InputLayer":
                continue                                  # not part of Sequential.layers
            if cn == "Flatten":
                self.layers.append(Layer(cn, c, cur, cur))
            elif cn == "Dense":
                if cur is None:
                    raise ValueError("the first layer needs an input shape")
                if not c.get("use_bias", True):
                    raise ValueError("Dense layers without bias are not supported")
                units = int(c["units"])
                layer = Layer(cn, c, cur, units)
                layer._dense_index = len(acts)
                if not dims:
                    dims.append(cur)
                dims.append(units)
                acts.append(_activation_name(c.get("activation", "linear")))
                self.layers.append(layer)
                cur = units
            else:
                raise ValueError(f"unsupported layer class '{cn}'")
        if not acts:
            raise ValueError("the model has no Dense layer")
        self.dims, self.acts = tuple(dims), tuple(acts)
        self.spec = MLPSpec(self.dims, self.acts, "scce" if acts[-1] == "softmax" else "mse")
        for l in self.layers:
            l._model = self
        self.weights_flat = np.zeros(self.spec.n_params, dtype=np.float32)
        self._plan = None
        self.reset_glorot(np.random.default_rng())

    # ------------------------------------------------------------------ weights
    def reset_glorot(self, rng: np.random.Generator):
        """Keras default of model_from_json: GlorotUniform kernels, zero biases (Appendix A6)."""
        off = 0
        for i, o in zip(self.dims[:-1], self.dims[1:]):
            lim = np.sqrt(6.0 / (i + o))
            self.weights_flat[off:off + i * o] = rng.uniform(-lim, lim, size=i * o).astype(np.float32)
            self.weights_flat[off + i * o:off + (i + 1) * o] = 0.0
            off += (i + 1) * o

    @property
    def trainable_variables(self) -> List[np.ndarray]:
        return [v for l in self.layers for v in l.trainable_variables]

    def get_weights(self) -> List[np.ndarray]:
        return [v.copy() for v in self.trainable_variables]

    def set_weights(self, weights: Sequence[np.ndarray]):
        tv = self.trainable_variables
        if len(weights) != len(tv):
            raise ValueError(f"expected {len(tv)} weight arrays, got {len(weights)}")
        for dst, src in zip(tv, weights):
            dst[...] = np.asarray(src, dtype=np.float32).reshape(dst.shape)

    def set_flat(self, flat):
        self.weights_flat[...] = np.asarray(flat, dtype=np.float32).reshape(-1)

    def to_json(self) -> str:
        return json.dumps(self._config)

    def count_params(self) -> int:
        return self.spec.n_params

    # ------------------------------------------------------------------ inference
    def _forward(self, x) -> np.ndarray:
        import torch
        from ..engine import MLPPlan
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(len(x), -1))
        if self._plan is None or self._plan.max_batch < len(x):
            self._plan = MLPPlan(self.spec, max_batch=max(len(x), 1))
        th = torch.as_tensor(self.weights_flat).cuda()
        out = self._plan.forward(th, torch.as_tensor(x).cuda())
        return out[0].cpu().numpy()

    def __call__(self, x, training=False):
        return Array(self._forward(x))

    def predict(self, x, verbose=0, batch_size=None):
        return np.asarray(self._forward(x))


def model_from_json(model_config: str) -> DenseNet:
    return DenseNet(model_config)


def sequential_json(input_shape, units: Sequence[int], activations: Sequence[str], flatten: bool = False) -> str:
    """Builds the Keras-2.15 JSON a ``tf.keras.Sequential`` of Dense layers serialises to
    (for callers that have no TensorFlow to produce it)."""
    input_shape = tuple(int(d) for d in (input_shape if isinstance(input_shape, (tuple, list)) else (input_shape,)))
    layers = [{"module": "keras.layers", "class_name": "InputLayer",
               "config": {"batch_input_shape": [None, *input_shape], "dtype": "float32", "sparse": False,
                          "ragged": False, "name": "input_1"}, "registered_name": None}]
    if flatten or len(input_shape) > 1:
        layers.append({"module": "keras.layers", "class_name": "Flatten",
                       "config": {"name": "flatten", "trainable": True, "dtype": "float32",
                                  "batch_input_shape": [None, *input_shape], "data_format": "channels_last"},
                       "registered_name": None, "build_config": {"input_shape": [None, *input_shape]}})
    prev = int(np.prod(input_shape))
    for k, (u, a) in enumerate(zip(units, activations)):
        layers.append({"module": "keras.layers", "class_name": "Dense",
                       "config": {"name": "dense" if k == 0 else f"dense_{k}", "trainable": True, "dtype": "float32",
                                  "units": int(u), "activation": a, "use_bias": True,
                                  "kernel_initializer": {"module": "keras.initializers", "class_name": "GlorotUniform",
                                                         "config": {"seed": None}, "registered_name": None},
                                  "bias_initializer": {"module": "keras.initializers", "class_name": "Zeros",
                                                       "config": {}, "registered_name": None},
                                  "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
                                  "kernel_constraint": None, "bias_constraint": None},
                       "registered_name": None, "build_config": {"input_shape": [None, prev]}})
        prev = int(u)
    return json.dumps({"class_name": "Sequential", "config": {"name": "sequential", "layers": layers},
                       "keras_version": "2.15.0", "backend": "tensorflow"})
