#!/usr/bin/env python3
"""Diagnostic (GPU box): which part of a chunked device-resident run costs the device 2.5 us per step at C2
(train(2000) took 26.1 us per step against 23.6 for the same sgld_run calls on the same buffers).  Adds the pieces one at
a time to the bare chunk calls: the caller's stream waiting for the run stream at the end (THAT is it: 25.8 against
23.5 us), a wait on the main stream before each chunk, the chunk's table copy on the main stream (pageable / pinned
source), a fresh stream per call against one stream for all calls (none of these matters)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
N = 2000
table, sizes = opt._batch_plan(N)
idx, losses = opt._resident_buffers(table, N)
chunks = [32, 48, 72, 108, 162, 243, 364, 512, 459]
lrs = np.asarray(opt._lr(opt._n + np.arange(N, dtype=np.float64))).astype(np.float32).tolist()

def direct(st, final_wait):
    s0 = 0
    with torch.cuda.stream(st):
        for n in chunks:
            opt._plan.sgld_run(opt._theta, opt._mean_dev, opt._sq_mean_dev, opt._x_dev, opt._y_dev, idx, sizes[s0:s0 + n],
                               lrs[s0:s0 + n], opt._n + s0, opt._seed, losses, use_graph=True, slot0=s0)
            s0 += n
    if final_wait:
        torch.cuda.current_stream().wait_stream(st)


def timed(label, fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:34s} {1e6 * best / N:.2f} us per step", flush=True)



one_stream = torch.cuda.Stream()
timed("direct, one with, no final wait", lambda: direct(one_stream, False))
timed("direct, one with, final wait", lambda: direct(one_stream, True))
timed("direct, fresh stream, no final wait", lambda: direct(torch.cuda.Stream(), False))
pinned = table.pin_memory()
timed("direct again after pin_memory", lambda: direct(one_stream, False))


def run(mode):
    main, st = torch.cuda.current_stream(), (one_stream if mode.endswith("_1stream") else torch.cuda.Stream())
    s0 = 0
    for n in chunks:
        if mode.startswith("copy_main_pageable"):
            idx[s0:s0 + n].copy_(table[s0:s0 + n])
        elif mode == "copy_main_pinned":
            idx[s0:s0 + n].copy_(pinned[s0:s0 + n], non_blocking=True)
        if not mode.startswith("bare"):
            st.wait_stream(main)
        with torch.cuda.stream(st):
            if mode == "copy_run_pinned":
                idx[s0:s0 + n].copy_(pinned[s0:s0 + n], non_blocking=True)
            opt._plan.sgld_run(opt._theta, opt._mean_dev, opt._sq_mean_dev, opt._x_dev, opt._y_dev, idx, sizes[s0:s0 + n],
                               lrs[s0:s0 + n], opt._n + s0, opt._seed, losses, use_graph=True, slot0=s0)
        s0 += n
    main.wait_stream(st)      # (every mode of run() ends with the device-side wait)

for mode in ("bare", "bare_1stream", "wait_only", "copy_main_pageable", "copy_main_pageable_1stream", "bare"):
    run(mode); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(mode); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{mode:20s} {1e6 * best / N:.2f} us per step", flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); opt.train(N); torch.cuda.synchronize()
    print(f"train({N})          {1e6 * (time.perf_counter() - t0) / N:.2f} us per step", flush=True)
