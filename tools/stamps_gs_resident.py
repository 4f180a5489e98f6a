#!/usr/bin/env python3
"""Diagnostic (GPU box): -DPYZ_STAMPS build; where the waves of k_svgd_gs_resident spend their cycles per particle at C5
(64 particles of 784->200->10): wave 0 (polls the kernel row, publishes the partials), wave 1 of a reducer workgroup
(sums a column's partials) and of a plain one."""
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
spec = engine.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
M, D = 64, spec.n_params
x_h, y_h = synth.mnist_like(2048)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
plan = engine.MLPPlan(spec, max_batch=1024, max_particles=M)
p = torch.empty((M, D), device="cuda")
engine.fill_normal(p, 1, _lib.STREAM_INIT, 0, 0.0, 1e-3)
am, av, loss = torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda"), torch.zeros(1, device="cuda")
for t in range(1, 4):
    plan.svgd_step(p, p, 0, am, av, x, y, 0.01, 1.0, t, loss, sweep="gauss_seidel", batch=1024)
torch.cuda.synchronize()
lib_ = _lib.load()
K, B, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * B * W * S * 2))()
check(lib_.pyz_debug_stamps(buf, K * B * W * S * 2))
raw = np.frombuffer(buf, dtype=np.uint64).reshape(K, B, W, S, 2).astype(np.int64)
names = ["partials of row i+1 (64 rows)", "wait for K_i + barrier", "repulsion, Adam, patch", "barrier, column sums, barrier",
         "publish / reduce a column", "next operands (loop top)"]
NL = 6
if os.environ.get("PYZ_SVGD_GS_RESIDENT", "1") == "2":   # k_svgd_gs_resident2 (distances one step early): its own laps
    names = ["wait for K_i (reducers' 63 + the critical distance) + barrier", "repulsion, Adam", "critical partial: column sum, publish",
             "next operands, patch", "partials of row i+2 (64 rows)", "barrier, column sums, publish, barriers", "rotate (loop end)"]
    NL = 7
print("resident Gauss-Seidel sweep at C5: cycles per particle (s_memtime ticks), wave w of workgroup b")
for b, w in ((0, 0), (0, 1), (0, 2), (100, 0), (100, 1), (207, 0)):
    lap = raw[3, b, w, :NL, 0] / 64.0
    print(f"  b={b:3d} w={w}: " + "  ".join(f"{n}: {v:7.0f}" for n, v in zip(names, lap)) + f"   sum {lap.sum():7.0f}")
