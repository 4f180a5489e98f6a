#!/usr/bin/env python3
"""Diagnostic (GPU box): phase timing of the fused HMC kernel (-DPYZ_STAMPS build)."""
import ctypes as C, os, subprocess, sys
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
xm, ym = synth.moons(2000)
xm, ym = xm[:1600], ym[:1600]
spec = engine.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=1600, max_particles=4)
q = torch.zeros((4, spec.n_params), device="cuda")
stats = torch.zeros((4, 8), device="cuda")
xd, yd = torch.as_tensor(xm).cuda(), torch.as_tensor(ym).cuda()
for k in range(5):
    plan.hmc_step(q, xd, yd, 20, 0.002, 0.5, 0.0, 1.0, [0.5] * 4, k, 7, stats)
torch.cuda.synchronize()
K, B, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * B * W * S * 2))()
_lib.check(_lib.load().pyz_debug_stamps(buf, K * B * W * S * 2))
a = np.frombuffer(buf, dtype=np.uint64).reshape(K, B, W, S, 2).astype(np.int64)
t = a[3, 0, :, :, 1] * 10.0   # block 0, [wave, slot] ns
t0 = t[:, 0].min()
for sl, nm in ((0, "start"), (1, "staged + momentum"), (2, "first loss_grad done"), (5, "last phase A done"), (6, "last phase B done"), (3, "leapfrog done"), (4, "end")):
    print(f"{nm:24s}", " ".join(f"{v - t0:8.0f}" for v in t[:, sl]))
print("one gradient evaluation (wave 0): %.0f ns; phase A %.0f ns (of the last call: A done - previous)" % (t[0, 2] - t[0, 1], 0))
print("last call: phase B = %.0f ns (wave 0)" % (t[0, 6] - t[0, 5]))
