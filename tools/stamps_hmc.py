#!/usr/bin/env python3
"""Diagnostic (GPU box): -DPYZ_STAMPS build; where wave 0 of workgroup 0 of k_hmc_resident spends its cycles per
gradient evaluation at C3 (2->50->2, 1600 rows, L = 20, one chain): the sections of pyz_hf_loss_grad and the exchange."""
import ctypes as C
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "bayesian_inference_for_nn_amd", "csrc")
lib = os.path.join(csrc, "libpyz_stamps.so")
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DPYZ_STAMPS",
                       "-Wno-unused-function", "-Wno-pass-failed", "pyz_api.hip", "-o", lib], cwd=csrc)
from bayesian_inference_for_nn_amd import _build
_build.LIB = lib
_build.build = lambda *a, **k: lib
import torch
from bayesian_inference_for_nn_amd import engine, synth, _lib
from bayesian_inference_for_nn_amd._lib import check
xm, ym = synth.moons(2000)
spec = engine.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=1600, max_particles=1)
q = torch.zeros((1, spec.n_params), device="cuda")
stats = torch.zeros((1, 8), device="cuda")
xd, yd = torch.as_tensor(xm[:1600]).cuda(), torch.as_tensor(ym[:1600]).cuda()
st = torch.cuda.Stream()
L = 20
with torch.cuda.stream(st):
    for k in range(5):
        plan.hmc_step(q, xd, yd, L, 0.005, 0.5, 0.0, 1.0, [0.5], k, 7, stats, burning=True)
st.synchronize()
lib_ = _lib.load()
K, B, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * B * W * S * 2))()
check(lib_.pyz_debug_stamps(buf, K * B * W * S * 2))
raw = np.frombuffer(buf, dtype=np.uint64).reshape(K, B, W, S, 2).astype(np.int64)
lap = raw[3, 0, 0, :, 0]
names = ["weight records + barrier", "phase A (rows)", "loss block sum", "phase B (hidden units)", "combine partials",
         "publish granules", "sweep + kick / drift", "loss granules + barriers"]
print(f"k_hmc_resident, C3, L = {L}: cycles of wave 0 / workgroup 0 per gradient evaluation (s_memtime, 100 MHz ticks x core ratio)")
for n, v in zip(names, lap):
    print(f"  {n:28s} {v / (L + 1):8.0f}")
print(f"  {'sum':28s} {lap.sum() / (L + 1):8.0f}")
