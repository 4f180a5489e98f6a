"""Oracle (test infrastructure): Dense-MLP forward, loss and reverse-mode gradient.

Restates the third-party arithmetic the reference calls on every step
(SURVEY.md section 8a rows G1-G3):

* Keras ``Dense`` forward ``act(x @ W + b)`` with ``W`` of shape (in, out) --
  called at ``Pyesian/optimizers/SGLD.py:55``, ``SGD.py:57``, ``HMC.py:155``,
  ``BBB.py:144``, ``SVGD.py:106``.
* The loss produced by ``Pyesian/datasets/Dataset.py:152-159`` (the loss *class*
  is instantiated with ``reduction='auto'``): ``SparseCategoricalCrossentropy``
  on a softmax-activated last layer = mean over the batch of
  ``logsumexp(z) - z[y]`` (Keras 2.15 re-uses the cached logits, Appendix A1);
  ``MeanSquaredError`` = mean over the last axis, then over the batch.
* ``tf.GradientTape.gradient`` (``SGLD.py:64``, ``HMC.py:134``,
  ``BBB.py:152-153,173``, ``SVGD.py:110``), written out by hand here and
  cross-checked against torch autograd in tests/test_oracle_kat.py.

Flat parameter order (``HMC.py:177-183``, ``SVGD.py:159-160,230-239``,
``nn/BayesianModel.py:73-77``): for each layer in model order, ``kernel``
(in x out, row-major) then ``bias`` (out).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

ACTIVATIONS = ("linear", "relu", "tanh", "sigmoid", "softmax")
LOSSES = ("scce", "mse")


@dataclass(frozen=True)
class MLPSpec:
    """dims = [in, h1, ..., out]; acts[l] is the activation of Dense layer l."""

    dims: Tuple[int, ...]
    acts: Tuple[str, ...]
    loss: str = "scce"

    def __post_init__(self):
        assert len(self.dims) == len(self.acts) + 1
        for a in self.acts:
            assert a in ACTIVATIONS, a
        assert "softmax" not in self.acts[:-1], "softmax only as last activation"
        assert self.loss in LOSSES

    @property
    def n_layers(self) -> int:
        return len(self.acts)

    @property
    def n_params(self) -> int:
        return sum((i + 1) * o for i, o in zip(self.dims[:-1], self.dims[1:]))

    def offsets(self) -> List[Tuple[int, int]]:
        """[(kernel_offset, bias_offset)] per layer in the flat vector."""
        out, off = [], 0
        for i, o in zip(self.dims[:-1], self.dims[1:]):
            out.append((off, off + i * o))
            off += (i + 1) * o
        return out

    def layer_slices(self) -> List[slice]:
        """Flat slice holding kernel+bias of each layer."""
        out, off = [], 0
        for i, o in zip(self.dims[:-1], self.dims[1:]):
            out.append(slice(off, off + (i + 1) * o))
            off += (i + 1) * o
        return out


def unpack(theta: np.ndarray, spec: MLPSpec):
    ws = []
    for (ko, bo), i, o in zip(spec.offsets(), spec.dims[:-1], spec.dims[1:]):
        ws.append((theta[ko:ko + i * o].reshape(i, o), theta[bo:bo + o]))
    return ws


def pack(weights: Sequence[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    return np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)]) for w, b in weights])


def glorot_uniform(spec: MLPSpec, rng: np.random.Generator, dtype=np.float32) -> np.ndarray:
    """Keras default initialisation of ``model_from_json`` (Appendix A6):
    GlorotUniform kernels, zero biases."""
    parts = []
    for i, o in zip(spec.dims[:-1], spec.dims[1:]):
        lim = np.sqrt(6.0 / (i + o))
        parts.append(rng.uniform(-lim, lim, size=(i, o)).astype(dtype).reshape(-1))
        parts.append(np.zeros(o, dtype=dtype))
    return np.concatenate(parts)


def _act(z: np.ndarray, name: str) -> np.ndarray:
    if name == "linear":
        return z
    if name == "relu":
        return np.maximum(z, 0)
    if name == "tanh":
        return np.tanh(z)
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-z))
    if name == "softmax":
        zs = z - z.max(axis=-1, keepdims=True)
        e = np.exp(zs)
        return e / e.sum(axis=-1, keepdims=True)
    raise ValueError(name)


def _act_grad_from_output(h: np.ndarray, name: str) -> np.ndarray:
    """d act / d z expressed through the activation output h."""
    if name == "linear":
        return np.ones_like(h)
    if name == "relu":
        return (h > 0).astype(h.dtype)
    if name == "tanh":
        return 1.0 - h * h
    if name == "sigmoid":
        return h * (1.0 - h)
    raise ValueError(name)


def forward(theta: np.ndarray, x: np.ndarray, spec: MLPSpec, dtype=np.float64):
    """Returns (activations, logits): activations[0] = x, activations[l+1] =
    output of Dense layer l (post-activation); logits = pre-activation of the
    last layer."""
    theta = np.asarray(theta, dtype=dtype)
    h = np.asarray(x, dtype=dtype).reshape(len(x), -1)  # Flatten layer
    acts = [h]
    z = None
    for (w, b), a in zip(unpack(theta, spec), spec.acts):
        z = h @ w + b
        h = _act(z, a)
        acts.append(h)
    return acts, z


def predict(theta, x, spec: MLPSpec, dtype=np.float64) -> np.ndarray:
    return forward(theta, x, spec, dtype)[0][-1]


def loss_value(out: np.ndarray, logits: np.ndarray, y: np.ndarray, spec: MLPSpec):
    if spec.loss == "scce":
        assert spec.acts[-1] == "softmax", "SCCE is restated for a softmax last layer only"
        y = np.asarray(y).reshape(-1).astype(np.int64)
        zs = logits - logits.max(axis=-1, keepdims=True)
        lse = np.log(np.exp(zs).sum(axis=-1))
        return (lse - zs[np.arange(len(y)), y]).mean()
    y = np.asarray(y, dtype=out.dtype).reshape(out.shape)
    return ((out - y) ** 2).mean(axis=-1).mean()


def loss_and_grad(theta, x, y, spec: MLPSpec, dtype=np.float64):
    """(mean loss, d loss / d theta as a flat vector, model output)."""
    theta = np.asarray(theta, dtype=dtype)
    acts, logits = forward(theta, x, spec, dtype)
    out = acts[-1]
    n = len(out)
    loss = loss_value(out, logits, y, spec)
    if spec.loss == "scce":
        yi = np.asarray(y).reshape(-1).astype(np.int64)
        delta = out.copy()
        delta[np.arange(n), yi] -= 1.0
        delta /= n
    else:
        yt = np.asarray(y, dtype=dtype).reshape(out.shape)
        delta = 2.0 * (out - yt) / (n * out.shape[1])
        if spec.acts[-1] == "softmax":
            s = (delta * out).sum(axis=-1, keepdims=True)
            delta = out * (delta - s)
        else:
            delta = delta * _act_grad_from_output(out, spec.acts[-1])
    grads = [None] * spec.n_layers
    ws = unpack(theta, spec)
    for l in range(spec.n_layers - 1, -1, -1):
        h_in = acts[l]
        gw = h_in.T @ delta
        gb = delta.sum(axis=0)
        grads[l] = (gw, gb)
        if l > 0:
            delta = (delta @ ws[l][0].T) * _act_grad_from_output(h_in, spec.acts[l - 1])
    return dtype(loss), pack(grads).astype(dtype), out
