#!/usr/bin/env python3
"""Diagnostic (GPU box): in-kernel stamps of k_svgd_gram_tile at C5 (M = 64, 784->200->10), Jacobi sweep.
Compiles a -DPYZ_STAMPS library (never shipped) unless PYZ_STAMPS_LIB names one."""
import ctypes as C
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "bayesian_inference_for_nn_amd", "csrc")
lib = os.environ.get("PYZ_STAMPS_LIB") or os.path.join(csrc, "libpyz_stamps.so")
if not os.environ.get("PYZ_STAMPS_LIB"):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DPYZ_STAMPS",
                           "-Wno-unused-function", "-Wno-pass-failed", "pyz_api.hip", "-o", lib], cwd=csrc)
from bayesian_inference_for_nn_amd import _build
_build.LIB = lib
_build.build = lambda *a, **k: lib
import torch
from bayesian_inference_for_nn_amd import engine, synth, _lib
DIMS, M, B = (784, 200, 10), 64, 1024
spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
n_local = int(sys.argv[1]) if len(sys.argv) > 1 else 64
plan = engine.MLPPlan(spec, max_batch=B, max_particles=n_local)
x_h, y_h = synth.mnist_like(4096)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
D = spec.n_params
allp = torch.as_tensor(np.stack([synth.glorot_uniform(DIMS, seed=s) for s in range(M)])).cuda()
local = allp[:n_local].clone()
am, av = torch.zeros((n_local, D), device="cuda"), torch.zeros((n_local, D), device="cuda")
loss = torch.zeros(1, device="cuda")
idx = torch.arange(B, dtype=torch.int32, device="cuda")
for t in range(1, 4):
    plan.svgd_step(local, allp, 0, am, av, x, y, 1e-3, 1.0, t, loss, sweep="jacobi", batch=B, row_idx=idx)
torch.cuda.synchronize()
K, Bk, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * Bk * W * S * 2))()
_lib.check(_lib.load().pyz_debug_stamps(buf, K * Bk * W * S * 2))
a = np.frombuffer(buf, dtype=np.uint64).reshape(K, Bk, W, S, 2).astype(np.int64)
t = a[4, :, :4, :5, 1] * 10.0
live = t[:, 0, 0] > 0
t = t[live]
labels = ["start", "first slab: matrix instructions issued", "all slabs done", "four partial Grams combined", "partials stored"]
g0 = t[:, :, 0].min()
print(f"k_svgd_gram_tile ({n_local} local rows): {t.shape[0]} stamped workgroups; last start {t[:, :, 0].max() - g0:.0f}, last end {t[:, :, 4].max() - g0:.0f} ns")
t0 = t[:, :, 0].min(axis=1, keepdims=True)
for s_, lab in enumerate(labels):
    rel = t[:, :, s_] - t0
    print(f"   {lab:40s} median {np.median(rel):7.0f}   p90 {np.percentile(rel, 90):7.0f} ns")
