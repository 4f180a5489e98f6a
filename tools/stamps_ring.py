#!/usr/bin/env python3
"""Diagnostic (GPU box): -DPYZ_STAMPS build; where a wave of k_dense_fwd_ring spends its cycles per slab
(own DMA wait / barrier / DMA issue / fragment reads + matrix instructions), 8 particles of 784 -> 200, batch 1024."""
import ctypes as C
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "bayesian_inference_for_nn_amd", "csrc")
lib = os.path.join(csrc, "libpyz_stamps.so")
if os.environ.get("PYZ_STAMPS_LIB"):
    lib = os.environ["PYZ_STAMPS_LIB"]
else:
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DPYZ_STAMPS",
                           "-Wno-unused-function", "-Wno-pass-failed", "pyz_api.hip", "-o", lib], cwd=csrc)
from bayesian_inference_for_nn_amd import _build
_build.LIB = lib
_build.build = lambda *a, **k: lib
import torch
from bayesian_inference_for_nn_amd import engine, synth, _lib
from bayesian_inference_for_nn_amd._lib import check, ptr
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spec = engine.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=1024, max_particles=P)
x = torch.as_tensor(synth.mnist_like(4096)[0]).cuda()
th = torch.empty((P, spec.n_params), device="cuda")
engine.fill_normal(th, 3, _lib.STREAM_INIT, 0, 0.0, 0.05)
lib_ = _lib.load()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    check(lib_.pyz_bench_dense_kernel(plan.h, 0, 0, ptr(th), P, ptr(x), None, 1024, None, 10, C.c_void_p(st.cuda_stream)))
st.synchronize()
K, B, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * B * W * S * 2))()
check(lib_.pyz_debug_stamps(buf, K * B * W * S * 2))
raw = np.frombuffer(buf, dtype=np.uint64).reshape(K, B, W, S, 2).astype(np.int64)
a = raw[0, :, :, :, 0]   # [block, wave, slot]
steps = raw[0, :, :, :4, 1]
nw = int((a[0, :, 4] > 0).sum())
ns = a[0, 0, 5]
print(f"P {P}: {nw} waves per workgroup, {ns} slabs; cycles per slab and wave (median over {B} workgroups):")
names = ["own DMA wait", "barrier", "DMA issue", "reads + MFMA issue"]
for i, nm in enumerate(names):
    per = np.median(a[:, :nw, i], axis=0) / ns
    print(f"  {nm:20s} " + " ".join(f"{v:6.0f}" for v in per))
tot = np.median(a[:, :nw, 4], axis=0)
print("  whole kernel (cycles)  " + " ".join(f"{v:6.0f}" for v in tot), " per slab", round(float(np.median(tot)) / ns))
