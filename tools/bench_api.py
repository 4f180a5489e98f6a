#!/usr/bin/env python3
"""Secondary measurement (GPU box): the BASELINE.json configurations through the drop-in Python surface
(Optimizer.compile / train), i.e. including the host work of each step() where the method has no
device-resident run.  Prints one JSON line per configuration."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayesian_inference_for_nn_amd import synth  # noqa: E402
from bayesian_inference_for_nn_amd.datasets import Dataset  # noqa: E402
from bayesian_inference_for_nn_amd.distributions import GaussianPrior  # noqa: E402
from bayesian_inference_for_nn_amd.losses import MeanSquaredError, SparseCategoricalCrossentropy  # noqa: E402
from bayesian_inference_for_nn_amd.nn import model_from_json, sequential_json  # noqa: E402
from bayesian_inference_for_nn_amd.optimizers import BBB, HMC, SGD, SGLD, SVGD, SWAG  # noqa: E402
from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters  # noqa: E402


def timed_train(opt, n, warm, after_warm=None):
    """(first train(n) call, a repeated train(n) call) in us per step; the first one of the device-resident
    methods includes capturing and instantiating the hipGraph."""
    opt.train(warm)
    if after_warm:
        after_warm()
    res = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.train(n)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e6)
    return res


def main():
    out = []
    x, y = synth.linreg(600)
    ds = Dataset((x, y), MeanSquaredError, "Regression", seed=0)
    cfg = sequential_json(1, [1], ["linear"])
    start = model_from_json(cfg)
    opt = SGD()
    opt.compile(HyperParameters(lr=1e-3, frequency=1, batch_size=64), cfg, ds, verbose=False, starting_model=start, seed=1)
    out.append(("C1 SGD 1->1, batch 64, train(verbose=False)", timed_train(opt, 4000, 64)))

    xm, ym = synth.mnist_like(12000)
    dsm = Dataset((xm, ym), SparseCategoricalCrossentropy, "Classification", seed=0)
    cfg2 = sequential_json(784, [200, 10], ["relu", "softmax"])
    opt = SGLD()
    opt.compile(HyperParameters(lr_upper=0.01, lr_lower=0.003, lr_gamma=0.99, batch_size=1024), cfg2, dsm, verbose=False, seed=2)
    out.append(("C2 SGLD 784->200->10, batch 1024, train(verbose=False)", timed_train(opt, 2000, 64)))
    opt = SWAG()
    opt.compile(HyperParameters(lr=0.01, k=10, frequency=10, scale=1.0, batch_size=1024), cfg2, dsm, verbose=False,
                starting_model=model_from_json(cfg2), seed=3)
    out.append(("SWAG 784->200->10, batch 1024, train(verbose=False)", timed_train(opt, 2000, 64)))

    xs, ys = synth.moons(2000, seed=42)
    dmo = Dataset((xs, ys), SparseCategoricalCrossentropy, "Classification", seed=3)
    cfg3 = sequential_json(2, [50, 2], ["relu", "softmax"])
    opt = HMC()
    opt.compile(HyperParameters(epsilon=0.005, m=0.5, L=20), cfg3, dmo, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=4)
    # the warm-up call runs the 10 burn-in proposals of HMC.train (the chain leaves q = 0); the timed calls are sampling only
    res = timed_train(opt, 300, 20, after_warm=lambda: setattr(opt, "_nb_burn_epoch", 0))
    out.append((f"C3 HMC moons 2->50->2, L=20, eps=0.005, N={dmo.train_size}, train(), per sample "
                f"(accept rate {opt._accepted_runs / max(opt._total_runs, 1):.2f})", res))

    # C4 at BASELINE.md's settings: 60 000 MNIST-shaped rows -> 48 000 training / 6 000 validation / 6 000 test rows (Dataset's
    # 0.8 / 0.1 / 0.1 split, Dataset.py:113-122), lr 5e-4, alpha 0.3 (best row of logs/BBB_mnist.txt:15), prior (0, 1) (BBB_mnist.py:42)
    x60, y60 = synth.mnist_like(60000)
    ds60 = Dataset((x60, y60), SparseCategoricalCrossentropy, "Classification", seed=0)
    assert (ds60.train_size, ds60.valid_size) == (48000, 6000)
    cfg4 = sequential_json(784, [400, 400, 10], ["relu", "relu", "softmax"])
    opt = BBB()
    opt.compile(HyperParameters(lr=5e-4, alpha=0.3, batch_size=1024), cfg4, ds60, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=5)
    out.append(("C4 BBB 784->400->400->10, batch 1024, 48 000 rows, lr 5e-4, alpha 0.3, prior (0, 1): train(verbose=False) incl. the "
                "9-in-10 validation forward over 6 000 rows (device-resident run)", timed_train(opt, 300, 20)))
    opt = BBB()
    opt.compile(HyperParameters(lr=5e-4, alpha=0.3, batch_size=1024), cfg4, ds60, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=5)
    opt._val_n = 0                                              # the same run without the validation forward
    out.append(("C4 BBB (same settings) train(verbose=False) WITHOUT the validation forward", timed_train(opt, 300, 20)))
    opt = BBB()
    opt.compile(HyperParameters(lr=5e-4, alpha=0.3, batch_size=1024), cfg4, ds60, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=5)
    opt._train_resident = lambda n: False                       # the per-step Python loop of round 2, for comparison
    out.append(("C4 BBB (same settings) per-step Python loop incl. validation (round 2's path)", timed_train(opt, 300, 20)))

    opt = SVGD()
    opt.compile(HyperParameters(lr=1e-3, M=64, batch_size=1024), cfg2, dsm, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=6)
    out.append(("C5 SVGD M=64 784->200->10, batch 1024, train(), Gauss-Seidel (reference order)", timed_train(opt, 20, 3)))
    for name, (first, us) in out:
        print(json.dumps({"config": name, "us_per_step": round(us, 1), "steps_per_s": round(1e6 / us, 1),
                          "us_per_step_first_call": round(first, 1)}), flush=True)


if __name__ == "__main__":
    main()
