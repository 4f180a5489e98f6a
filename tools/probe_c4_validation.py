#!/usr/bin/env python3
"""Diagnostic (GPU box): the kernels of the 6 000-row validation forward of C4 (784 -> 400 -> 400 -> 10), ring on / off."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesian_inference_for_nn_amd import engine, synth
spec = engine.MLPSpec((784, 400, 400, 10), ("relu", "relu", "softmax"), "scce")
x_h, y_h = synth.mnist_like(6000)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
w = torch.as_tensor(synth.glorot_uniform(spec.dims)).cuda()
plan = engine.MLPPlan(spec, max_batch=6000)
for _ in range(3):
    plan.loss_grad(w, x, y, want_grad=False)
with engine.KernelProbe(16) as kp:
    plan.loss_grad(w, x, y, want_grad=False)
print(json.dumps({"ring": os.environ.get("PYZ_FWD_RING", "1"), "sub": os.environ.get("PYZ_FWD_RING_SUB", "auto"),
                  "kernels_us": [(n, round(v, 1)) for n, v in kp.launches]}))
