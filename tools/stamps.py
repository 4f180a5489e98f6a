#!/usr/bin/env python3
"""Diagnostic (GPU box): build csrc/libpyz_stamps.so with -DPYZ_STAMPS, run a few SGLD steps of
the headline configuration and print where wave 0 of each workgroup spends its time.
Never used for reported numbers (the stamps forbid overlaps the real kernels have)."""
import ctypes as C
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "bayesian_inference_for_nn_amd", "csrc")
lib = os.path.join(csrc, "libpyz_stamps.so")
if os.environ.get("PYZ_STAMPS_LIB"):          # a prebuilt -DPYZ_STAMPS library (A/B of two source states on one box)
    lib = os.environ["PYZ_STAMPS_LIB"]
else:
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DPYZ_STAMPS",
                           "-Wno-unused-function", "-Wno-pass-failed"] + os.environ.get("PYZ_STAMP_FLAGS", "").split() +
                          ["pyz_api.hip", "-o", lib], cwd=csrc)
from bayesian_inference_for_nn_amd import _build
_build.LIB = lib
_build.build = lambda *a, **k: lib
import torch
from bayesian_inference_for_nn_amd import engine, synth, _lib
DIMS = (784, 200, 10)
spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=1024)
x_h, y_h = synth.mnist_like(48000)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
D = spec.n_params
theta = torch.as_tensor(synth.glorot_uniform(DIMS)).cuda()
mean, sq = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
n = 64
idx_h, sizes = synth.batch_plan(48000, 1024, n)
idx = torch.as_tensor(idx_h).cuda()
lrs = synth.sgld_lr_table(n, 0.01, 0.003, 0.99, 0, n)
losses = torch.zeros(n, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for rep in range(3):
        plan.sgld_run(theta, mean, sq, x, y, idx, sizes, lrs, 0, 1, losses, use_graph=True)
st.synchronize()
K, B, W, S = 6, 256, 16, 8
buf = (C.c_uint64 * (K * B * W * S * 2))()
_lib.check(_lib.load().pyz_debug_stamps(buf, K * B * W * S * 2))
a = np.frombuffer(buf, dtype=np.uint64).reshape(K, B, W, S, 2).astype(np.int64)
names = {0: ("k_dense_fwd", [(0, "start"), (1, "addr ready"), (2, "accumulate done"), (3, "end")], 224, 16),
         1: ("k_head_rows", [(0, "start"), (1, "loads issued"), (2, "z reduced"), (3, "loss row done"), (4, "row stored"), (5, "end")], 256, 4),
         2: ("k_wgrad_all", [(0, "start"), (1, "addr+prefetch issued"), (2, "accumulate done"), (4, "tile reduced"), (3, "end")], 182, 16)}
# global timeline of the last recorded step (s_memrealtime is one 100 MHz counter for the chip)
g0 = None
for k, (nm, slots, nb, nw) in names.items():
    tt = a[k, :nb, :nw, :, 1] * 10.0
    used = tt[:, :, 0] > 0
    st = np.where(used, tt[:, :, 0], np.inf).min(axis=1)
    en = np.where(used, tt[:, :, slots[-1][0]], 0).max(axis=1)
    if g0 is None:
        g0 = st.min()
    order = np.argsort(-en)[:6]
    print(f"## {nm}: slowest workgroups (id: start..end): " + "  ".join(f"{int(b)}: {st[b] - g0:.0f}..{en[b] - g0:.0f}" for b in order))
    print(f"## {nm}: first wg start {st.min() - g0:7.0f}  last wg start {st.max() - g0:7.0f}  first wg end {en.min() - g0:7.0f}  "
          f"last wg end {en.max() - g0:7.0f}  (ns since the first forward workgroup started; waves/wg {used.sum(axis=1).max()})")
for k, (nm, slots, nb, nw) in names.items():
    t = a[k, :nb, :nw, :, 1] * 10.0          # ns, [block, wave, slot]
    cyc = a[k, :nb, :nw, :, 0]
    t0 = t[:, :, 0].min(axis=1, keepdims=True)  # block start = earliest wave start
    print(f"== {nm}: {nb} workgroups x {nw} waves; times in ns since the workgroup's first wave started")
    clk = np.median((cyc[:, 0, slots[-1][0]] - cyc[:, 0, 0]) / np.maximum(t[:, 0, slots[-1][0]] - t[:, 0, 0], 1))
    print(f"   clock {clk:.2f} GHz")
    for sl, label in slots:
        rel = t[:, :, sl] - t0
        per_wave = np.median(rel, axis=0)
        print(f"   {label:22s} median over waves {np.median(rel):7.0f}  min-wave {per_wave.min():7.0f}  max-wave {per_wave.max():7.0f}   per wave: "
              + " ".join(f"{v:5.0f}" for v in per_wave))

# batch workers of k_wgrad_all (the workgroups past the duties workgroup): start .. end relative to the kernel's first workgroup
tw = a[2, :, :, :, 1] * 10.0
k0 = tw[:182, :, 0][tw[:182, :, 0] > 0].min()
pw = tw[183:256]
ok = pw[:, 0, 3] > 0
if ok.any():
    st_, en_ = pw[ok][:, 0, 0] - k0, pw[ok][:, :, 3].max(axis=1) - k0
    print(f"== k_wgrad_all batch workers: {ok.sum()} workgroups; start median {np.median(st_):.0f} (max {st_.max():.0f}); end median {np.median(en_):.0f}  max {en_.max():.0f} ns after the kernel's first workgroup started")
