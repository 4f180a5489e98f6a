#!/usr/bin/env python3
"""Diagnostic (GPU box): one rank's share of the sharded SVGD step (8 local particles of 64, C5) with the kernel matrix beside
the gradient pass -- eager on two streams (events) against the same fork / join captured once as a hipGraph and replayed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesian_inference_for_nn_amd import _lib, engine, synth

dev = "cuda"
spec = engine.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
M, D, n_local = 64, spec.n_params, int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
aux = torch.cuda.Stream()


def step(main):
    aux.wait_stream(main)
    plan.svgd_kernel_matrix(allp, 0, n_local, 1.0, stream=aux)
    done = aux.record_event()
    plan.svgd_gradients(local, x, y, batch=1024, row_idx=idx[0])
    main.wait_event(done)
    plan.svgd_combine(local, allp, 0, am, av, 0.01, 1.0, 3, loss)


def timed(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    us_eager = timed(lambda: step(torch.cuda.current_stream()))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(torch.cuda.current_stream())
    us_graph = timed(g.replay)

    def seq():
        plan.svgd_gradients(local, x, y, batch=1024, row_idx=idx[0])
        plan.svgd_sweep(local, allp, 0, am, av, 0.01, 1.0, 3, loss, sweep="jacobi")
    us_seq = timed(seq)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        seq()
    us_seq_graph = timed(g2.replay)
hi = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hi):
    us_prio = timed(lambda: step(torch.cuda.current_stream()))
print(json.dumps({"n_local": n_local, "two_streams_eager_main_high_priority_us": round(us_prio, 1), "two_streams_eager_us": round(us_eager, 1), "two_streams_graph_us": round(us_graph, 1),
                  "one_stream_eager_us": round(us_seq, 1), "one_stream_graph_us": round(us_seq_graph, 1)}))
