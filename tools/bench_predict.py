#!/usr/bin/env python3
"""Secondary measurement (GPU box): BayesianModel.predict -- nb_samples weight draws of a
Normal posterior over 784->200->10 and the forwards of a 10 000-row test split as one particle-batched launch
sequence.  Prints one JSON line per setting of PYZ_FWD_LDS, alternating (PYZ_FWD_LDS_MAXROWS is lifted so that the
switch decides; by default launches of more than 2 048 rows keep the one-wave-per-tile forward)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayesian_inference_for_nn_amd import synth  # noqa: E402
from bayesian_inference_for_nn_amd.distributions import tfd  # noqa: E402
from bayesian_inference_for_nn_amd.distributions.tf import TensorflowProbabilityDistribution  # noqa: E402
from bayesian_inference_for_nn_amd.nn import BayesianModel, sequential_json  # noqa: E402


def main():
    cfg = sequential_json(784, [200, 10], ["relu", "softmax"])
    x, _ = synth.mnist_like(10000)
    model = BayesianModel(cfg)
    rng = np.random.default_rng(0)
    layers = [i for i, l in enumerate(model._model.layers) if len(l.trainable_variables) != 0]
    for li in layers:
        n = sum(int(np.prod(v.shape)) for v in model._model.layers[li].trainable_variables)
        dist = TensorflowProbabilityDistribution(tfd.Normal((rng.normal(size=n) * 0.05).astype(np.float32),
                                                            np.full(n, 0.01, np.float32)))
        model.apply_distribution(dist, li, li)
    os.environ["PYZ_FWD_LDS_MAXROWS"] = "1000000"     # let the switch decide at any row count
    for lds in ("0", "1", "0", "1"):                  # alternating: the first timing of a process runs slower
        os.environ["PYZ_FWD_LDS"] = lds
        model.predict(x, nb_samples=100)            # warm-up (plan, buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            model.predict(x, nb_samples=100)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"config": "predict 784->200->10, 100 draws x 10000 rows", "PYZ_FWD_LDS": int(lds),
                          "ms_per_predict": round(ms, 2)}), flush=True)


if __name__ == "__main__":
    main()
