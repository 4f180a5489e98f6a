"""Oracle (test infrastructure): ``BayesianModel.predict`` Monte-Carlo read-out.

Follows ``Pyesian/nn/BayesianModel.py:106-129``: for each of ``nb_samples``
weight draws -- ``_sample_weights`` (``:63-77``) assigns one
``Distribution.sample()`` per layer interval in flat order -- run the model,
replace NaN by 0 (``:125``), accumulate; return (list of per-sample outputs,
their mean).  The weight draws are an argument here (one flat vector per
sample) so that CPU and GPU evaluate identical weights.
"""

from __future__ import annotations

import numpy as np

from .mlp import MLPSpec, predict as _forward_out


def predict(weight_samples: np.ndarray, x: np.ndarray, spec: MLPSpec, dtype=np.float64):
    """weight_samples: (S, D).  Returns (samples_results (S, N, out), mean (N, out))."""
    outs = []
    for w in weight_samples:
        p = _forward_out(w, x, spec, dtype)
        p = np.where(np.isnan(p), 0.0, p)                 # BayesianModel.py:125
        outs.append(p)
    outs = np.stack(outs)
    return outs, outs.sum(axis=0) / len(weight_samples)   # BayesianModel.py:126-128


def sampled_index(acc_frequencies, w: int) -> int:
    """``Sampled.sample`` (distributions/Sampled.py:29-32): ``w`` is the host
    ``random.randint(1, total)``; returns the index picked by ``bisect_left``."""
    import bisect
    return bisect.bisect_left(list(acc_frequencies), w)
