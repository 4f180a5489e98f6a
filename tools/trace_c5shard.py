#!/usr/bin/env python3
"""Diagnostic (GPU box, under rocprofv3 --kernel-trace): a few sharded SVGD steps of one rank of eight (C5), the kernel
matrix on a second stream beside the gradient pass -- the trace shows where the kernels of the two streams really run.
    rocprofv3 --kernel-trace -d gpurun_out/trace -o c5shard --output-format csv -- python3 tools/trace_c5shard.py [groups|empty]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayesian_inference_for_nn_amd import _lib, engine, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "local"
dev = "cuda"
spec = engine.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
M, D, n_local = 64, spec.n_params, 8
x_h, y_h = synth.mnist_like(8192)
x, y = torch.as_tensor(x_h).to(dev), torch.as_tensor(y_h).to(dev)
idx_h, sizes = synth.batch_plan(8192, 1024, 8)
idx = torch.as_tensor(idx_h).to(dev)
loss = torch.zeros(1, device=dev)
plan = engine.MLPPlan(spec, max_batch=1024, max_particles=n_local)
allp = torch.empty((M, D), device=dev)
engine.fill_normal(allp, 1, _lib.STREAM_INIT, 0, 0.0, 1.0)
local = allp[:n_local].clone()
am, av = torch.zeros((n_local, D), device=dev), torch.zeros((n_local, D), device=dev)
groups = torch.zeros((8, 64 * 64), dtype=torch.float64, device=dev)
plan.svgd_gram_groups(allp, 0, 8, groups)
aux = torch.cuda.Stream()
prio = torch.cuda.Stream(priority=-1)
if mode.endswith("_prio"):
    aux = prio
for it in range(8):
    main = torch.cuda.current_stream()
    aux.wait_stream(main)
    if mode.startswith("groups"):
        plan.svgd_gram_groups(allp, 0, 1, groups, stream=aux)
        plan.svgd_kernel_matrix_groups(groups, allp, 0, n_local, 1.0, stream=aux)
    else:
        plan.svgd_kernel_matrix(allp, 0, n_local, 1.0, stream=aux)
    done = aux.record_event()
    plan.svgd_gradients(local, x, y, batch=sizes[it], row_idx=idx[it])
    main.wait_event(done)
    plan.svgd_combine(local, allp, 0, am, av, 0.01, 1.0, it + 1, loss)
torch.cuda.synchronize()
print("done", float(loss))
