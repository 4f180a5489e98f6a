#!/usr/bin/env python3
"""Diagnostic (GPU box): per-kernel durations of pyz_predict at the size of tools/bench_predict.py
(100 weight draws x 10 000 rows of 784->200->10), through engine.KernelProbe, plus the wall time of the call
with everything resident on the device."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bayesian_inference_for_nn_amd import engine, synth

spec = engine.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
n, S = 10000, 100
plan = engine.MLPPlan(spec, max_batch=n, max_particles=S)
x = torch.as_tensor(synth.mnist_like(n)[0]).cuda()
w = (torch.randn((S, spec.n_params), device="cuda") * 0.05)
plan.predict(w, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    plan.predict(w, x)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
with engine.KernelProbe(4096) as kp:
    plan.predict(w, x)
per = kp.by_kernel()
print(json.dumps({"config": "pyz_predict 100 draws x 10000 rows, 784->200->10, operands resident", "ms_per_call": round(wall, 3),
                  "kernels": {k: [c, round(us, 1), round(c * us / 1e3, 3)] for k, (c, us) in per.items()},
                  "env": {k: v for k, v in os.environ.items() if k.startswith("PYZ_")}}))
