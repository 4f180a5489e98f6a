#!/usr/bin/env python3
"""Diagnostic (GPU box): time the headline SGLD step with an alternative build of the library.
usage: python tools/variants.py <path-to-lib.so> [label]    (env switches of the library apply)
Prints one JSON line: graph-replay us/step and the per-kernel durations (engine.KernelProbe)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(lib)
from bayesian_inference_for_nn_amd import _build
_build.LIB = lib
_build.build = lambda *a, **k: lib
import torch
from bayesian_inference_for_nn_amd import engine, synth
DIMS = (784, 200, 10)
spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=1024)
x_h, y_h = synth.mnist_like(48000)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
D = spec.n_params
theta = torch.as_tensor(synth.glorot_uniform(DIMS)).cuda()
mean, sq = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
n = 2048 + 256
idx_h, sizes = synth.batch_plan(48000, 1024, n)
idx = torch.as_tensor(idx_h).cuda()
lrs = synth.sgld_lr_table(n, 0.01, 0.003, 0.99, 0, n)
losses = torch.zeros(n, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    plan.sgld_run(theta, mean, sq, x, y, idx, sizes[:256], lrs[:256], 0, 1, losses, use_graph=True)
    st.synchronize()
    t0 = time.perf_counter()
    plan.sgld_run(theta, mean, sq, x, y, idx, sizes[256:], lrs[256:], 256, 1, losses, use_graph=True, slot0=256)
    st.synchronize()
    dt = time.perf_counter() - t0
    plan.sgld_run(theta, mean, sq, x, y, idx, sizes[:16], lrs[:16], n, 1, losses, use_graph=False)
    with engine.KernelProbe(1024) as kp:
        plan.sgld_run(theta, mean, sq, x, y, idx, sizes[:256], lrs[:256], n + 16, 1, losses, use_graph=False)
step_us = dt / 2048 * 1e6
us = [round(v[1], 2) for v in kp.by_kernel().values()]
print(json.dumps({"label": label, "us_per_step": round(step_us, 2), "steps_per_s": round(1e6 / step_us),
                  "kernels_us": us, "loss": round(float(losses[n - 1].item()), 5), "theta_bits_sum": int(theta.view(torch.int32).to(torch.int64).sum().item()),
                  "env": {k_: v for k_, v in os.environ.items() if k_.startswith("PYZ_")}}), flush=True)
