#!/usr/bin/env python3
"""Diagnostic (GPU box): in-kernel stamps of k_dense_fwd_lds<7> on the particle-batched C5 forward (64 x 784->200, B = 1024).
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
plan = engine.MLPPlan(spec, max_batch=B, max_particles=M)
x_h, y_h = synth.mnist_like(4096)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
D = spec.n_params
parts = torch.as_tensor(np.stack([synth.glorot_uniform(DIMS, seed=s) for s in range(M)])).cuda()
idx = torch.arange(B, dtype=torch.int32, device="cuda")
for _ in range(3):
    plan.svgd_gradients(parts, x, y, batch=B, row_idx=idx)
torch.cuda.synchronize()
K, Bk, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * Bk * W * S * 2))()
_lib.check(_lib.load().pyz_debug_stamps(buf, K * Bk * W * S * 2))
a = np.frombuffer(buf, dtype=np.uint64).reshape(K, Bk, W, S, 2).astype(np.int64)
t = a[0, :, :4, :, 1] * 10.0
t = t[t[:, 0, 0] > 0]
labels = ["start", "first slab staged", "slab 24: begin", "slab 24: matrix instructions issued", "slab 24: next slab staged",
          "slab 24: barrier passed", "all slabs done", "end"]
g0 = t[:, :, 0].min()
print(f"k_dense_fwd_lds<7>: {t.shape[0]} stamped workgroups (blockIdx.y == 0); last start {t[:, :, 0].max() - g0:.0f}, last end {t[:, :, 7].max() - g0:.0f} ns")
t0 = t[:, :, 0].min(axis=1, keepdims=True)
for s_, lab in enumerate(labels):
    rel = t[:, :, s_] - t0
    print(f"   {lab:40s} median {np.median(rel):8.0f}   p10 {np.percentile(rel, 10):8.0f}   p90 {np.percentile(rel, 90):8.0f} ns")
sh = a[0, :, :4, :, 0][a[0, :, 0, 0, 1] > 0]          # shader-clock stamps (s_memtime) of the same workgroups
clk = (sh[:, :, 6] - sh[:, :, 1]) / ((t[:, :, 6] - t[:, :, 1]) * 1e-9) / 1e9
print(f"   shader clock over the slab loop: median {np.median(clk):.2f} GHz (min {clk.min():.2f}, max {clk.max():.2f})")
d = t[:, :, 3] - t[:, :, 2]
print(f"   slab 24, matrix instructions: median {np.median(d):.0f} ns; stage {np.median(t[:, :, 4] - t[:, :, 3]):.0f}; barrier wait {np.median(t[:, :, 5] - t[:, :, 4]):.0f}")
