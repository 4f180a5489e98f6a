#!/usr/bin/env python3
"""Secondary measurements (GPU box): the other BASELINE.json configurations through the C-ABI step
entry points, eager launches, HIP-event timed.  Prints one JSON line per configuration.
(bench.py is the headline; these numbers go to DESIGN.md / profiles/.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayesian_inference_for_nn_amd import _lib, engine, synth  # noqa: E402


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def main():
    only = sys.argv[1:] or ["c1", "c3", "c4", "c5"]
    dev = "cuda"
    if "c1" in only:   # SGD linreg 1->1, batch 64 and whole set
        x, y = synth.linreg(600)
        spec = engine.MLPSpec((1, 1), ("linear",), "mse")
        plan = engine.MLPPlan(spec, max_batch=600)
        th = torch.tensor([0.3, -0.1], device=dev)
        loss = torch.zeros(1, device=dev)
        xd, yd = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
        for b in (64, 480):
            us = timed(lambda: plan.sgd_step(th, xd, yd, 1e-3, loss, batch=b), 500)
            print(json.dumps({"config": "C1 SGD linreg 1->1", "batch": b, "us_per_step": round(us, 2), "steps_per_s": round(1e6 / us, 1)}))
    if "c4" in only:   # BBB 784->400->400->10
        dims = (784, 400, 400, 10)
        spec = engine.MLPSpec(dims, ("relu", "relu", "softmax"), "scce")
        plan = engine.MLPPlan(spec, max_batch=1024)
        D = spec.n_params
        x_h, y_h = synth.mnist_like(48000)
        x, y = torch.as_tensor(x_h).to(dev), torch.as_tensor(y_h).to(dev)
        mu, rho, w = torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.zeros(D, device=dev)
        cost = torch.zeros(4, device=dev)
        idx_h, sizes = synth.batch_plan(48000, 1024, 64)
        idx = torch.as_tensor(idx_h).to(dev)
        k = [0]

        def step():
            s = k[0] % 64
            k[0] += 1
            plan.bbb_step(mu, rho, w, x, y, 5e-4, 0.3, 0.0, 1.0, k[0], 2024, cost, batch=sizes[s], row_idx=idx[s])
        us = timed(step, 300)
        vplan = engine.MLPPlan(spec, max_batch=6000)
        xv = x[:6000].contiguous()
        yv = y[:6000].contiguous()
        usv = timed(lambda: vplan.loss_grad(w, xv, yv, want_grad=False), 100)
        with engine.KernelProbe(64) as kp:        # one step, every kernel with its own begin / end timestamps
            step()
        print(json.dumps({"config": "C4 BBB 784->400->400->10 B=1024", "kernels_us": [(n, round(v, 2)) for n, v in kp.launches]}))
        print(json.dumps({"config": "C4 BBB 784->400->400->10 B=1024", "us_per_step": round(us, 2), "steps_per_s": round(1e6 / us, 1),
                          "tflops": round(2292e6 / us / 1e6, 2), "validation_forward_us": round(usv, 2),
                          "steps_per_s_with_reference_validation": round(1e6 / (us + 0.9 * usv), 1), "cost": float(cost[0])}))
    if "c3" in only:   # HMC moons 2->50->2, L=20, epsilon 0.005, m 0.5 (HMC_classification.py:135-136)
        xm, ym = synth.moons(2000)
        xm, ym = xm[:1600], ym[:1600]
        spec = engine.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce")
        for chains in (1, 8, 64):
            plan = engine.MLPPlan(spec, max_batch=1600, max_particles=chains)
            q = torch.zeros((chains, spec.n_params), device=dev)          # HMC.py:69-72: q <- prior mean
            stats = torch.zeros((chains, 8), device=dev)
            xd, yd = torch.as_tensor(xm).to(dev), torch.as_tensor(ym).to(dev)
            k, acc = [0], []

            def step(burning=False):
                k[0] += 1
                plan.hmc_step(q, xd, yd, 20, 0.005, 0.5, 0.0, 1.0, np.random.default_rng(k[0]).random(chains), k[0], 7, stats,
                              burning=burning)
                if not burning:
                    acc.append(stats[:, 0].clone())
            side = torch.cuda.Stream()       # a stream that can be captured (sliced proposals replay a hipGraph)
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(10):          # HMC.train's 10 forced burn-in proposals (HMC.py:111-116): the chain leaves q = 0
                    step(burning=True)
                us = timed(step, 50)
            rate = float(torch.stack(acc[3:]).mean())     # (timed() warms up with 3 proposals)
            print(json.dumps({"config": "C3 HMC moons 2->50->2 L=20 N=1600 eps=0.005 m=0.5 prior (0, 1)", "chains_on_gpu": chains,
                              "us_per_sample": round(us, 1), "samples_per_s_per_chain": round(1e6 / us, 1),
                              "samples_per_s_aggregate": round(chains * 1e6 / us, 1),
                              "grad_evals_per_s": round(chains * 21 * 1e6 / us, 1), "accept_rate_timed": round(rate, 3),
                              "kernel_path": os.environ.get("PYZ_HMC_MULTI", "1") == "1" and chains <= 16 and
                              (os.environ.get("PYZ_HMC_RESIDENT", "1") != "0" and "k_hmc_resident" or "k_hmc_multi") or "k_hmc_fused"}))
    if "c5" in only:   # SVGD 64 particles 784->200->10
        dims = (784, 200, 10)
        spec = engine.MLPSpec(dims, ("relu", "softmax"), "scce")
        M, D = 64, spec.n_params
        plan = engine.MLPPlan(spec, max_batch=1024, max_particles=M)
        x_h, y_h = synth.mnist_like(48000)
        x, y = torch.as_tensor(x_h).to(dev), torch.as_tensor(y_h).to(dev)
        idx_h, sizes = synth.batch_plan(48000, 1024, 16)
        idx = torch.as_tensor(idx_h).to(dev)
        loss = torch.zeros(1, device=dev)
        for sweep, scale in (("gauss_seidel", 1.0), ("jacobi", 1.0), ("gauss_seidel", 0.001)):
            p = torch.empty((M, D), device=dev)
            engine.fill_normal(p, 1, _lib.STREAM_INIT, 0, 0.0, scale)
            am, av = torch.zeros((M, D), device=dev), torch.zeros((M, D), device=dev)
            k = [0]

            bufs = [p, torch.empty_like(p)]

            def step():
                s = k[0] % 16
                k[0] += 1
                if sweep == "jacobi":     # snapshot and updated matrix alternate (the sweep only writes its output rows)
                    plan.svgd_step(bufs[1], bufs[0], 0, am, av, x, y, 0.01, 1.0, k[0], loss, sweep=sweep, batch=sizes[s], row_idx=idx[s])
                    bufs.reverse()
                else:
                    plan.svgd_step(p, p, 0, am, av, x, y, 0.01, 1.0, k[0], loss, sweep=sweep, batch=sizes[s], row_idx=idx[s])
            us = timed(step, 20)
            with engine.KernelProbe(256) as kp:
                step()
            agg = {}
            for n, v in kp.launches:
                c, t = agg.get(n, (0, 0.0))
                agg[n] = (c + 1, t + v)
            print(json.dumps({"config": "C5 SVGD M=64 784->200->10 B=1024 1 GPU", "sweep": sweep, "init_scale": scale,
                              "kernels_us_total": {n: [c, round(t, 1)] for n, (c, t) in agg.items()}}))
            print(json.dumps({"config": "C5 SVGD M=64 784->200->10 B=1024 1 GPU", "sweep": sweep, "init_scale": scale,
                              "us_per_step": round(us, 1), "svgd_steps_per_s": round(1e6 / us, 2),
                              "particle_grad_steps_per_s": round(M * 1e6 / us, 1), "tflops": round(45.2e9 / us / 1e6, 2),
                              "loss": float(loss)}))
    if "c5shard" in only:   # ONE rank's share of the 8-way sharded SVGD step (no gather: compute only), on this one GPU
        dims = (784, 200, 10)
        spec = engine.MLPSpec(dims, ("relu", "softmax"), "scce")
        M, D = 64, spec.n_params
        x_h, y_h = synth.mnist_like(48000)
        x, y = torch.as_tensor(x_h).to(dev), torch.as_tensor(y_h).to(dev)
        idx_h, sizes = synth.batch_plan(48000, 1024, 16)
        idx = torch.as_tensor(idx_h).to(dev)
        loss = torch.zeros(1, device=dev)
        for n_local in (8, 16, 32, 64):
            plan = engine.MLPPlan(spec, max_batch=1024, max_particles=n_local)
            allp = torch.empty((M, D), device=dev)
            engine.fill_normal(allp, 1, _lib.STREAM_INIT, 0, 0.0, 1.0)
            local = allp[:n_local].clone()
            am, av = torch.zeros((n_local, D), device=dev), torch.zeros((n_local, D), device=dev)
            k = [0]

            def grad():
                s = k[0] % 16
                k[0] += 1
                plan.svgd_gradients(local, x, y, batch=sizes[s], row_idx=idx[s])

            def sweep():
                plan.svgd_sweep(local, allp, 0, am, av, 0.01, 1.0, k[0] + 1, loss, sweep="jacobi")

            aux = torch.cuda.Stream()

            def step_overlapped():     # what SVGD.step does: kernel matrix of the snapshot on a second stream beside the gradients
                main = torch.cuda.current_stream()
                aux.wait_stream(main)
                plan.svgd_kernel_matrix(allp, 0, n_local, 1.0, stream=aux)
                done = aux.record_event()
                grad()
                main.wait_event(done)
                plan.svgd_combine(local, allp, 0, am, av, 0.01, 1.0, k[0] + 1, loss)

            # the distance pass sharded over the elements too (SVGD.step with shard_gram): this rank's share of the groups,
            # the other ranks' group sums standing in the buffer already (no exchange: compute only)
            per = max(1, 8 * n_local // M)
            groups = torch.zeros((8, 64 * 64), dtype=torch.float64, device=dev)
            plan.svgd_gram_groups(allp, 0, 8, groups)

            def step_overlapped_groups():
                main = torch.cuda.current_stream()
                aux.wait_stream(main)
                plan.svgd_gram_groups(allp, 0, per, groups, stream=aux)
                plan.svgd_kernel_matrix_groups(groups, allp, 0, n_local, 1.0, stream=aux)
                done = aux.record_event()
                grad()
                main.wait_event(done)
                plan.svgd_combine(local, allp, 0, am, av, 0.01, 1.0, k[0] + 1, loss)

            def step_sequential():
                grad()
                sweep()
            grad()
            us_g = timed(grad, 30)
            us_s = timed(sweep, 30)
            us_seq = timed(step_sequential, 30)
            us_ovl = timed(step_overlapped, 30)
            us_grp = timed(step_overlapped_groups, 30)
            with engine.KernelProbe(64) as kp:
                grad()
                sweep()
            with engine.KernelProbe(64) as kp2:
                plan.svgd_gram_groups(allp, 0, per, groups)
                plan.svgd_kernel_matrix_groups(groups, allp, 0, n_local, 1.0)
                plan.svgd_combine(local, allp, 0, am, av, 0.01, 1.0, k[0] + 1, loss)
            print(json.dumps({"config": f"C5 SVGD one rank of {M // n_local}: {n_local} local particles of 64, B=1024 (compute only, no gather)",
                              "gradients_us": round(us_g, 1), "sweep_us": round(us_s, 1), "step_sequential_us": round(us_seq, 1),
                              "step_kernel_matrix_on_second_stream_us": round(us_ovl, 1),
                              "step_distance_pass_sharded_over_elements_us": round(us_grp, 1),
                              "kernels_us": [(n, round(v, 1)) for n, v in kp.launches],
                              "kernels_us_sharded_distance_pass": [(n, round(v, 1)) for n, v in kp2.launches]}))

if __name__ == "__main__":
    main()
