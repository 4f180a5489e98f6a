#!/usr/bin/env python3
"""Diagnostic (GPU box): back-to-back launches of ONE kernel of the headline step
(pyz_bench_dense_kernel) with random-gathered, identity-gathered and ungathered batch rows."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from bayesian_inference_for_nn_amd import _build
    lib = os.path.abspath(sys.argv[1])
    _build.LIB = lib
    _build.build = lambda *a, **k: lib
import numpy as np, torch
from bayesian_inference_for_nn_amd import engine, synth, _lib
from bayesian_inference_for_nn_amd._lib import check
from bayesian_inference_for_nn_amd.engine import ptr
DIMS = (784, 200, 10)
B = int(os.environ.get("PROBE_BATCH", 1024))
spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
plan = engine.MLPPlan(spec, max_batch=B)
x_h, y_h = synth.mnist_like(48000)
x, y = torch.as_tensor(x_h).cuda(), torch.as_tensor(y_h).cuda()
theta = torch.as_tensor(synth.glorot_uniform(DIMS)).cuda()
grad = torch.zeros(spec.n_params, device="cuda")
lib = _lib.load()
rng = np.random.default_rng(0)
modes = {"random": torch.as_tensor(rng.permutation(48000)[:B].astype(np.int32)).cuda(),
         "identity": torch.arange(B, dtype=torch.int32, device="cuda"), "none": None}
stream = torch.cuda.current_stream().cuda_stream
out = {}
for kind, nm in ((0, "k_dense_fwd"), (2, "k_wgrad_all")):
    for mode, idx in modes.items():
        xx = x if idx is not None else x[:B].contiguous()
        yy = y if idx is not None else y[:B].contiguous()
        plan.loss_grad(theta, xx, yy, row_idx=idx, batch=B)
        def go(it):
            check(lib.pyz_bench_dense_kernel(plan.h, kind, 0, ptr(theta), 1, ptr(xx), ptr(idx), B, ptr(grad), it, C.c_void_p(stream)))
        go(20)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(200); e1.record(); torch.cuda.synchronize()
        out[f"{nm}/{mode}"] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
print(json.dumps(out), flush=True)
