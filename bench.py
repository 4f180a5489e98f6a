#!/usr/bin/env python3
"""Headline benchmark: SGLD grad-steps/s on the MLP 784->200->10, batch 1024
(BASELINE.json configs[1]), one chain per GPU.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: forward, loss, backward, fused
noise + parameter + moment update (Pyesian/optimizers/SGLD.py:46-95).  Inputs
(data set, row-index plan, weights, moments) are resident in HBM before the timed
region; the timed region is K steps bracketed by barrier + synchronize; the time
is the max over ranks and `value` the whole-job steps/s.

Extra objects on the JSON line:
  roofline      the dominant kernel of the step (the slowest of k_dense_fwd, k_head,
                k_wgrad_all): algorithmic FLOP per launch / its average in-pipeline
                duration, measured live with HIP events recorded on the bench stream
                around every kernel of 256 further steps, against the dense fp32 MFMA peak.
  cpu_baseline  the oracle's eager torch-CPU restatement of the same step
                (oracle/torch_eager.py, kind "port") on the host cores.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
DIMS = (784, 200, 10)
BATCH = 1024
GRAPH_STEPS = 32          # steps per captured graph in csrc/pyz_api.hip (PYZ_GRAPH_STEPS)
N_ROWS = 48_000
LR_UPPER, LR_LOWER, LR_GAMMA = 0.01, 0.003, 0.99      # reference tests/unittest2.py:73
SEED = 2024
FLOP_PER_STEP = 654.5e6        # C2 per grad-step
BYTES_PER_STEP = 7.03e6


def host_cores() -> int:
    """CPU threads this process may really use: the cgroup quota when there is one (a GPU box
    exposes every core in the affinity mask but grants a share), else the affinity count."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 32)


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json, written by tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE
    passes, KB units, FETCH_SIZE calibrated on this kernel's known byte count as
    MI355X_MICROARCH.md section HBM prescribes for non-16-B/lane access).  None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        rec = json.load(open(path))["kernels"][kernel.split("[")[0]]
        return int(rec["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(budget_s: float = 12.0):
    """Time the eager CPU restatement (reference op granularity) on a bounded sample."""
    import torch
    from oracle import mlp as o_mlp, torch_eager, sgld as o_sgld
    from bayesian_inference_for_nn_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    spec = o_mlp.MLPSpec(DIMS, ("relu", "softmax"), "scce")
    x, y = synth.mnist_like(8192)
    xt, yt = torch.as_tensor(x), torch.as_tensor(y.astype(np.int64))
    eager = torch_eager.EagerSGLD(spec, synth.glorot_uniform(DIMS), o_sgld.lr_schedule(10_000, LR_UPPER, LR_LOWER, LR_GAMMA),
                                  generator=torch.Generator().manual_seed(0))
    rng = np.random.default_rng(0)

    def one():
        idx = torch.as_tensor(rng.permutation(8192)[:BATCH])
        eager.step(xt[idx], yt[idx])

    for _ in range(5):
        one()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s or n < 20:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} eager torch-CPU SGLD steps (fp32, batch {BATCH}, 784->200->10) in {dt:.1f} s; "
                      "TensorFlow is not installed, so the reference's eager step is timed through oracle/torch_eager.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the isolated-kernel timing (used for the PMC passes)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bayesian_inference_for_nn_amd import engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU over RCCL ("nccl"); PYZ_BENCH_BACKEND=gloo rehearses the multi-rank logic
    # with several ranks sharing one GPU (a 1-GPU test box)
    backend = os.environ.get("PYZ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1) if world > 1 else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    dev = torch.device("cuda", torch.cuda.current_device())

    spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
    D = spec.n_params
    plan = engine.MLPPlan(spec, max_batch=BATCH)
    x_h, y_h = synth.mnist_like(N_ROWS)
    x = torch.as_tensor(x_h).to(dev)
    y = torch.as_tensor(y_h).to(dev)
    theta = torch.as_tensor(synth.glorot_uniform(DIMS)).to(dev)
    mean = torch.zeros(D, device=dev)
    sq_mean = torch.zeros(D, device=dev)
    total = args.warmup + args.steps
    idx_h, sizes = synth.batch_plan(N_ROWS, BATCH, total, seed=1236 + 1000 * rank)
    idx = torch.as_tensor(idx_h).to(dev)
    lrs = synth.sgld_lr_table(total, LR_UPPER, LR_LOWER, LR_GAMMA, 0, total)
    losses = torch.zeros(total, device=dev)
    stream = torch.cuda.Stream()
    use_graph = not args.no_graph

    def run(s0, n):
        with torch.cuda.stream(stream):
            plan.sgld_run(theta, mean, sq_mean, x, y, idx, sizes[s0:s0 + n], lrs[s0:s0 + n], s0, SEED + rank,
                          losses, use_graph=use_graph, slot0=s0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # One-time setup, like compilation: the library captures its 32-step hipGraph the first time a run is long
    # enough.  Prime it here on the real buffers (their addresses are baked into the graph) and put the chain
    # state back, so that neither the warm-up nor the timed region contains the capture when W < 32.
    if use_graph and total >= GRAPH_STEPS:
        saved = (theta.clone(), mean.clone(), sq_mean.clone())
        run(0, GRAPH_STEPS)
        torch.cuda.synchronize()
        for dst, src in zip((theta, mean, sq_mean), saved):
            dst.copy_(src)
        del saved
    if args.warmup > 0:
        run(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_loss = float(losses[total - 1].item())
    if not np.isfinite(last_loss):
        raise SystemExit(f"bench.py: non-finite loss {last_loss} on rank {rank}")

    roof = None
    if rank == 0 and not args.no_roofline:
        # The step is three kernels (k_dense_fwd, k_head_rows, k_wgrad_all).  Their in-pipeline durations are
        # measured live with HIP events recorded on the bench stream around every kernel of 256 further
        # steps of the same chain (eager launches: an event cannot sit inside a graph node sequence).
        n_prof = 256
        pidx_h, psizes = synth.batch_plan(N_ROWS, BATCH, n_prof, seed=977)
        pidx = torch.as_tensor(pidx_h).to(dev)
        plr = synth.sgld_lr_table(total + n_prof, LR_UPPER, LR_LOWER, LR_GAMMA, total, n_prof)
        plosses = torch.zeros(n_prof, device=dev)
        with torch.cuda.stream(stream):
            plan.sgld_profile(theta, mean, sq_mean, x, y, pidx, psizes[:16], plr[:16], total, SEED + rank, plosses)   # warm
            us = plan.sgld_profile(theta, mean, sq_mean, x, y, pidx, psizes, plr, total + 16, SEED + rank, plosses)
        names = ["k_dense_fwd", "k_head_rows", "k_wgrad_all"]
        us = list(us)
        # An event record between two kernels costs queue time of its own (the instrumented step is
        # slower than the timed region's).  The kernels tile the step, so the per-record overhead is
        # (sum of the instrumented durations - step time of the timed region) / kernels; it is removed from each.
        step_us = dt / args.steps * 1e6
        raw_us = list(us)
        overhead = max(0.0, (sum(us) - step_us) / len(us))
        us = [v - overhead for v in us]
        # algorithmic FLOP per launch: forward of layer 0 = 2 B (K+1) N; head = last layer forward
        # + its data gradient; k_wgrad_all = [dW; db] of both layers
        flops = [2.0 * BATCH * (DIMS[0] + 1) * DIMS[1],
                 2.0 * BATCH * (DIMS[1] + 1) * DIMS[2] + 2.0 * BATCH * DIMS[1] * DIMS[2],
                 2.0 * BATCH * (DIMS[0] + 1) * DIMS[1] + 2.0 * BATCH * (DIMS[1] + 1) * DIMS[2]]
        k = int(np.argmax(us))
        achieved = flops[k] / (us[k] * 1e-6) / 1e12
        roof = {"bound": "mfma", "kernel": names[k], "kernel_us": round(us[k], 3), "flop_per_launch": flops[k],
                "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(names[k]),
                "kernels_us": {n: round(v, 3) for n, v in zip(names, us)},
                "kernels_us_with_event_overhead": {n: round(v, 3) for n, v in zip(names, raw_us)},
                "whole_step": {"flop": FLOP_PER_STEP, "bytes": BYTES_PER_STEP, "us": round(step_us, 3),
                               "tflops": round(FLOP_PER_STEP / (step_us * 1e-6) / 1e12, 3),
                               "frac_mfma": round(FLOP_PER_STEP / (step_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                               "gbps": round(BYTES_PER_STEP / (step_us * 1e-6) / 1e9, 1),
                               "frac_hbm": round(BYTES_PER_STEP / (step_us * 1e-6) / 8.0e12, 5)}}

    if rank == 0:
        out = {
            "metric": "posterior samples/sec (grad-steps/sec) on MLP 784->200->10, batch 1024",
            "value": round(args.gpus * args.steps / dt, 2),
            "unit": "grad-steps/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "SGLD, MLP 784->200->10 (D=159010), synthetic MNIST-shaped 48000x784 fp32 resident in HBM, "
                                   "batch 1024 (last batch of an epoch 896), 1 chain per GPU, hipGraph replay"
                                   if use_graph else "SGLD C2, eager launches",
                       "lr": [LR_UPPER, LR_LOWER, LR_GAMMA], "seed": SEED, "final_loss": round(last_loss, 6),
                       "parallelism": f"independent chains x{args.gpus} (replicas only, no data-path collective)"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
