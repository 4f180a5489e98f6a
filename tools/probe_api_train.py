#!/usr/bin/env python3
"""Diagnostic (GPU box): where the host spends SGLD.train(2000) at C2 (cProfile), and the device-side timeline of the run
(kernel probe is not usable here: it would force eager launches)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesian_inference_for_nn_amd import synth
from bayesian_inference_for_nn_amd.datasets import Dataset
from bayesian_inference_for_nn_amd.losses import SparseCategoricalCrossentropy
from bayesian_inference_for_nn_amd.nn import sequential_json
from bayesian_inference_for_nn_amd.optimizers import SGLD
from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters

xm, ym = synth.mnist_like(12000)
dsm = Dataset((xm, ym), SparseCategoricalCrossentropy, "Classification", seed=0)
cfg2 = sequential_json(784, [200, 10], ["relu", "softmax"])
opt = SGLD()
opt.compile(HyperParameters(lr_upper=0.01, lr_lower=0.003, lr_gamma=0.99, batch_size=1024), cfg2, dsm, verbose=False, seed=2)
opt.train(64); opt.train(2000); torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); opt.train(2000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"train(2000): returned after {1e3 * (t1 - t0):.2f} ms, device done after {1e3 * (t2 - t0):.2f} ms ({1e6 * (t2 - t0) / 2000:.2f} us per step)")
pr = cProfile.Profile(); pr.enable(); opt.train(2000); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4200])

# the same 2000 steps through the plan directly, on the optimizer's own buffers: one call, then the optimizer's chunks
import numpy as np
table, sizes = opt._batch_plan(2000)
idx, losses = opt._resident_buffers(table, 2000)
lrs = np.asarray(opt._lr(opt._n + np.arange(2000, dtype=np.float64))).astype(np.float32).tolist()
st = torch.cuda.Stream()
def direct(chunks):
    s0 = 0
    with torch.cuda.stream(st):
        for n in chunks:
            opt._plan.sgld_run(opt._theta, opt._mean_dev, opt._sq_mean_dev, opt._x_dev, opt._y_dev, idx, sizes[s0:s0 + n],
                               lrs[s0:s0 + n], opt._n + s0, opt._seed, losses, use_graph=True, slot0=s0)
            s0 += n
for name, chunks in (("one call", [2000]), ("optimizer's chunks", [32, 48, 72, 108, 162, 243, 364, 512, 459]), ("one call", [2000])):
    direct(chunks); torch.cuda.synchronize()
    t0 = time.perf_counter(); direct(chunks); torch.cuda.synchronize()
    print(f"plan.sgld_run on the optimizer's buffers, {name}: {1e6 * (time.perf_counter() - t0) / 2000:.2f} us per step; sizes {sorted(set(sizes))}")
