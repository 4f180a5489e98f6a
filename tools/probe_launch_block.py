#!/usr/bin/env python3
"""Diagnostic (GPU box): does a chained run return to the host before the device has worked through it?
Times, for runs of several lengths, the host-side duration of `sgld_run` (enqueue only) against the time until the
stream has drained.  A run is replayed as 32-step hipGraphs; if the call returns only when most of the run is done,
the host cannot lay out the next chunk's batches while the device works (Optimizer._run_resident_chunks)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
n = 4096
idx_h, sizes = synth.batch_plan(48000, 1024, n)
idx = torch.as_tensor(idx_h).cuda()
lrs = synth.sgld_lr_table(n, 0.01, 0.003, 0.99, 0, n)
losses = torch.zeros(n, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    plan.sgld_run(theta, mean, sq, x, y, idx, sizes[:100], lrs[:100], 0, 1, losses, use_graph=True)
    st.synchronize()
    for m in (32, 64, 128, 256, 512, 1024, 2048):
        for rep in range(2):
            t0 = time.perf_counter()
            plan.sgld_run(theta, mean, sq, x, y, idx, sizes[:m], lrs[:m], 0, 1, losses, use_graph=True)
            t1 = time.perf_counter()
            st.synchronize()
            t2 = time.perf_counter()
        print(json.dumps({"steps": m, "host_return_us": round((t1 - t0) * 1e6, 1), "drained_us": round((t2 - t0) * 1e6, 1),
                          "us_per_step": round((t2 - t0) * 1e6 / m, 2)}), flush=True)
