#!/usr/bin/env python3
"""GPU box: pyz_bbb_run at C4's shape (784->400->400->10, batch 1024, 48 000 rows), no validation forward --
per-kernel times of four eager steps of the run (KernelProbe) and us/step of the graph-replayed run.
PYZ_BBB_FUSE_SAMPLE=0 gives the run with k_bbb_sample in every step (compare in a second process)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayesian_inference_for_nn_amd import engine, synth  # noqa: E402


def main():
    dev = "cuda"
    spec = engine.MLPSpec((784, 400, 400, 10), ("relu", "relu", "softmax"), "scce")
    plan = engine.MLPPlan(spec, max_batch=1024)
    D = spec.n_params
    x_h, y_h = synth.mnist_like(48000)
    x, y = torch.as_tensor(x_h).to(dev), torch.as_tensor(y_h).to(dev)
    n = 320
    idx_h, sizes = synth.batch_plan(48000, 1024, n)
    idx = torch.as_tensor(idx_h).to(dev)
    mu, rho, w = torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.zeros(D, device=dev)
    costs = torch.zeros((n + 8, 4), device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def run(k):
            plan.bbb_run(mu, rho, w, x, y, idx[:k], sizes[:k], [5e-4] * k, 0.3, 0.0, 1.0, 1, 2024, costs)
        run(n)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            run(n)
        e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * n)
        with engine.KernelProbe(64) as kp:
            run(4)
        st.synchronize()
    print(json.dumps({"config": "C4 pyz_bbb_run, no validation", "fuse_sample": os.environ.get("PYZ_BBB_FUSE_SAMPLE", "1"),
                      "us_per_step_graph": round(us, 2), "kernels_us_4_eager_steps": [(k, round(v, 2)) for k, v in kp.launches],
                      "cost_last": float(costs[n - 1, 0])}))


if __name__ == "__main__":
    main()
