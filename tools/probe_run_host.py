#!/usr/bin/env python3
"""Diagnostic (GPU box): host time of one 20-step `sgld_run` call split into the Python checks / table conversion in front
of the C call, the C call itself (run start + graph launch), and the wait for the device (bench.py's driver invocation
times exactly this sequence: ~547 us for 20 steps of 24 us)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesian_inference_for_nn_amd import engine, synth
from bayesian_inference_for_nn_amd.engine import ptr, _stream, check
DIMS = (784, 200, 10)
spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=1024)
x_h, y_h = synth.mnist_like(48000)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
D = spec.n_params
theta = torch.as_tensor(synth.glorot_uniform(DIMS)).cuda()
mean, sq = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
n = 20
idx_h, sizes = synth.batch_plan(48000, 1024, n)
idx = torch.as_tensor(idx_h).cuda()
lrs = synth.sgld_lr_table(n, 0.01, 0.003, 0.99, 0, n)
losses = torch.zeros(n, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for rep in range(3):
        plan.sgld_run(theta, mean, sq, x, y, idx, sizes, lrs, 0, 1, losses, use_graph=True)
    torch.cuda.synchronize()
    for rep in range(5):
        t0 = time.perf_counter()
        plan.sgld_run(theta, mean, sq, x, y, idx, sizes, lrs, 0, 1, losses, use_graph=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # the bare C call with everything converted beforehand
        bs = (C.c_int32 * n)(*[int(b) for b in sizes]); lr = (C.c_float * n)(*[float(v) for v in lrs])
        args = (plan.h, ptr(theta), ptr(mean), ptr(sq), ptr(x), ptr(y), ptr(idx), bs, lr, n, 0, 0, 1, ptr(losses), 1, _stream())
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        check(plan.lib.pyz_sgld_run(*args))
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        print(f"engine.sgld_run: call {1e6 * (t1 - t0):6.1f} us, drained after {1e6 * (t2 - t0):6.1f} us | bare pyz_sgld_run: call "
              f"{1e6 * (t4 - t3):6.1f} us, drained after {1e6 * (t5 - t3):6.1f} us", flush=True)
