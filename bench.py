#!/usr/bin/env python3
"""Headline benchmark: SGLD grad-steps/s on the MLP 784->200->10, batch 1024
(BASELINE.json configs[1]), one chain per GPU.

    python bench.py --gpus N --steps K --warmup W            (the driver's contract)
    python bench.py --method svgd --gpus N ...               (BASELINE.json configs[4]: 64 particles sharded over N GPUs)

A step = one pass of the hot path over one batch: forward, loss, backward, fused
noise + parameter + moment update (Pyesian/optimizers/SGLD.py:46-95).  Inputs
(data set, row-index plan, weights, moments) are resident in HBM before the timed
region; the timed region is K steps bracketed by barrier + synchronize; the time
is the max over ranks and `value` the whole-job steps/s.

Everything one-time (graph capture for every chunk length a run of K or W steps uses) happens
before the warm-up; `config.path` says what the timed region really executed (read back from the
library: steps inside replayed hipGraphs / eager steps / graph launches).

Extra objects on the JSON line:
  roofline      the dominant kernel of the step (the slowest of k_dense_fwd, k_head_rows,
                k_wgrad_all): algorithmic FLOP per launch / its average duration.  Durations are
                measured live, after the timed region and independently of it: 256 further steps of
                the same chain are launched with a start / stop event pair on every kernel
                (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, what
                rocprofv3 --kernel-trace reports; the committed profiles/ summary of the same
                command must agree).
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
N_ROWS = 48_000
LR_UPPER, LR_LOWER, LR_GAMMA = 0.01, 0.003, 0.99      # reference tests/unittest2.py:73
SEED = 2024
FLOP_PER_STEP = 654.5e6        # C2 per grad-step
BYTES_PER_STEP = 7.03e6
SVGD_M, SVGD_LR = 64, 0.01     # BASELINE.json configs[4]; SVGD_mnist.py:11
SVGD_FLOP_PER_STEP = 45.2e9    # 41.9 gradients + 1.95 kernel + 1.3 repulsion
PMC_TRAFFIC = "r03_pmc_traffic.json"


def cpu_model() -> str:
    """The host CPU's model string (BASELINE.md section 3 asks for it beside the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


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
    """HBM bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC passes (profiles/r0x_pmc_traffic.json, written
    by tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes, KB units, FETCH_SIZE calibrated on this kernel's
    known byte count as MI355X_MICROARCH.md section HBM prescribes) -- a lookup, not a counter read of THIS run (a
    process cannot read the PMC counters of its own kernels).  Returns (bytes, detail) or (None, None):
    bytes = counters + a modelled term (the half of a wide stream FETCH_SIZE does not see); detail names both."""
    for name in (PMC_TRAFFIC, "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"][kernel.split("<")[0]]
            detail = {"source": "profiles/" + name + " (committed PMC passes of the same command; not read in this run)",
                      "counter_bytes": rec.get("counter_bytes_per_launch"),
                      "modelled_correction_bytes": rec.get("modelled_correction_bytes")}
            return int(rec["hbm_bytes_per_launch"]), detail
        except Exception:
            continue
    return None, None


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
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{n} eager torch-CPU SGLD steps (fp32, batch {BATCH}, 784->200->10) in {dt:.1f} s; "
                      "TensorFlow is not installed, so the reference's eager step is timed through oracle/torch_eager.py"}


def init_ranks(args):
    import torch
    import torch.distributed as dist
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
    return rank, world, backend, torch.device("cuda", torch.cuda.current_device())


def timed_region(run_warm, run_timed, world, backend, dev):
    """W warm-up steps, then the K timed steps between barrier + synchronize; max over ranks."""
    import torch
    import torch.distributed as dist

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_warm()
    fence()
    t0 = time.perf_counter()
    run_timed()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def whole_step(flop, nbytes, step_us):
    tf = flop / (step_us * 1e-6) / 1e12
    return {"flop": flop, "bytes": nbytes, "us": round(step_us, 3), "tflops": round(tf, 3),
            "frac_mfma": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "gbps": round(nbytes / (step_us * 1e-6) / 1e9, 1),
            "frac_hbm": round(nbytes / (step_us * 1e-6) / 8.0e12, 5)}


# ---------------------------------------------------------------------------------------------- SGLD (headline)
def bench_sgld(args, rank, world, backend, dev):
    import torch
    from bayesian_inference_for_nn_amd import engine, synth

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
    slots = total
    idx_h, sizes = synth.batch_plan(N_ROWS, BATCH, slots, seed=1236 + 1000 * rank)
    idx = torch.as_tensor(idx_h).to(dev)
    lrs = synth.sgld_lr_table(total, LR_UPPER, LR_LOWER, LR_GAMMA, 0, slots)
    losses = torch.zeros(slots, device=dev)
    stream = torch.cuda.Stream()
    use_graph = not args.no_graph

    def run(s0, n):
        if n <= 0:
            return
        with torch.cuda.stream(stream):
            plan.sgld_run(theta, mean, sq_mean, x, y, idx, sizes[s0:s0 + n], lrs[s0:s0 + n], s0, SEED + rank,
                          losses, use_graph=use_graph, slot0=s0)

    # One-time setup, like compilation: the library captures one hipGraph per run length it meets (32-step chunks
    # + one graph for the remainder).  Rehearse the exact call pair of the measurement on the real buffers (their
    # addresses are baked into the graphs) and put the chain state back: neither the warm-up nor the timed region
    # contains a capture.
    if use_graph:
        saved = (theta.clone(), mean.clone(), sq_mean.clone())
        run(0, args.warmup)
        run(args.warmup, args.steps)
        torch.cuda.synchronize()
        for dst, src in zip((theta, mean, sq_mean), saved):
            dst.copy_(src)
        del saved
        torch.cuda.synchronize()
    dt = timed_region(lambda: run(0, args.warmup), lambda: run(args.warmup, args.steps), world, backend, dev)
    kind, n_ran = plan.last_run_path()
    assert n_ran == args.steps
    path = {"kind": kind, "steps_in_graphs": args.steps if kind == "graph" else 0,
            "graph_launches": plan.last_run_graph_launches(),
            "steps_per_graph": "32 + one graph of the remainder's exact length" if kind == "graph" else None}
    last_loss = float(losses[total - 1].item())
    plan.check_finite()                                # PYZ_E_NAN if any step of the run produced a NaN / Inf loss
    if not np.isfinite(last_loss):
        raise SystemExit(f"bench.py: non-finite loss {last_loss} on rank {rank}")

    roof = None
    step_us = dt / args.steps * 1e6
    if rank == 0 and not args.no_roofline:
        # Per-kernel durations, measured independently of the timed region: 256 further steps of the same chain,
        # launched eagerly back to back, every kernel with its own start / stop event pair.
        n_prof = 256
        pidx_h, psizes = synth.batch_plan(N_ROWS, BATCH, n_prof + 16, seed=977)
        pidx = torch.as_tensor(pidx_h).to(dev)
        plr = synth.sgld_lr_table(total + n_prof + 16, LR_UPPER, LR_LOWER, LR_GAMMA, total, n_prof + 16)
        plosses = torch.zeros(n_prof + 16, device=dev)
        with torch.cuda.stream(stream):
            plan.sgld_run(theta, mean, sq_mean, x, y, pidx, psizes[:16], plr[:16], total, SEED + rank, plosses, use_graph=False)
            with engine.KernelProbe(4 * n_prof) as kp:
                plan.sgld_run(theta, mean, sq_mean, x, y, pidx, psizes[16:], plr[16:], total + 16, SEED + rank, plosses,
                              use_graph=True, slot0=16)   # (a probe forces eager launches)
        per = kp.by_kernel()
        # algorithmic FLOP per launch: forward of layer 0 = 2 B (K+1) N; head = last layer forward
        # + its data gradient; k_wgrad_all = [dW; db] of both layers
        f_fwd = 2.0 * BATCH * (DIMS[0] + 1) * DIMS[1]
        f_head = 2.0 * BATCH * (DIMS[1] + 1) * DIMS[2] + 2.0 * BATCH * DIMS[1] * DIMS[2]
        flop_of = {"k_dense_fwd": f_fwd, "k_head_rows": f_head,
                   "k_wgrad_all": f_fwd + 2.0 * BATCH * (DIMS[1] + 1) * DIMS[2]}
        kern = {}
        for name, (count, us) in per.items():
            base = name.split("<")[0]
            if base in flop_of and count >= n_prof:
                kern[name] = (us, flop_of[base])
        if kern:
            k = max(kern, key=lambda n: kern[n][0])
            us_k, fl = kern[k]
            achieved = fl / (us_k * 1e-6) / 1e12
            traffic, traffic_detail = pmc_traffic(k)
            roof = {"bound": "mfma", "kernel": k, "kernel_us": round(us_k, 3), "flop_per_launch": fl,
                    "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_detail": traffic_detail,
                    "kernels_us": {n: round(v[0], 3) for n, v in kern.items()},
                    "timing": "start/stop event pair per launch (hipExtLaunchKernelGGL), 256 eager steps after the timed region",
                    "whole_step": whole_step(FLOP_PER_STEP, BYTES_PER_STEP, step_us)}

    out = None
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
                                   "batch 1024 (last batch of an epoch 896), 1 chain per GPU, "
                                   + ("device-resident run replayed from hipGraphs" if kind == "graph" else f"{kind} launches"),
                       "path": path, "lr": [LR_UPPER, LR_LOWER, LR_GAMMA], "seed": SEED, "final_loss": round(last_loss, 6),
                       "parallelism": f"independent chains x{args.gpus} (replicas only, no data-path collective)"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline()
    return out


# ---------------------------------------------------------------------------------------------- SVGD (configs[4])
def cpu_baseline_svgd(budget_s: float = 12.0):
    """The reference's particle loop (SVGD.py:100-123: one tape per particle, float64 kernel row, legacy Adam) in its
    row-wise CPU restatement (oracle/torch_eager.py + oracle/svgd.py), timed on a bounded sample of the 64 particles."""
    import torch
    from oracle import mlp as o_mlp, svgd as o_svgd, torch_eager
    from bayesian_inference_for_nn_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    spec = o_mlp.MLPSpec(DIMS, ("relu", "softmax"), "scce")
    M, D = SVGD_M, spec.n_params
    x, y = synth.mnist_like(BATCH)
    rng = np.random.default_rng(0)
    parts = rng.normal(size=(M, D))
    m, v = np.zeros((M, D), np.float32), np.zeros((M, D), np.float32)

    def one(i, t):
        _, g = torch_eager.flat_grad(spec, parts[i].astype(np.float32), x, y.astype(np.int64), dtype=torch.float32)
        k_row, rep = o_svgd.rbf_row(parts, i, 1.0)
        phi = ((k_row.sum() * np.asarray(g, np.float64) + rep) / M).astype(np.float32)
        parts[i], m[i], v[i] = o_svgd.adam_update(parts[i].astype(np.float32), phi, m[i], v[i], t, SVGD_LR, np.float32)

    one(0, 1)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s or n < 8:
        one(n % M, 1 + n // M)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "particle-grad-steps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{n} particle updates of the reference's Gauss-Seidel loop (eager torch-CPU fp32 tape per particle + float64 "
                      f"kernel row over all 64 particles + legacy Adam) in {dt:.1f} s; TensorFlow is not installed"}


def bench_svgd(args, rank, world, backend, dev):
    """64 particles of 784->200->10 sharded over the ranks, batch 1024 replicated; one all-gather of the particle
    matrix per step (RCCL), Jacobi sweep: the kernel matrix of the gathered snapshot on a second stream beside the
    gradient pass, then the combine.  Total work is fixed: strong scaling."""
    import torch
    import torch.distributed as dist
    from bayesian_inference_for_nn_amd import _lib, engine, parallel, synth

    spec = engine.MLPSpec(DIMS, ("relu", "softmax"), "scce")
    D, M = spec.n_params, SVGD_M
    row0, n_local = parallel.shard_range(M, world, rank)
    plan = engine.MLPPlan(spec, max_batch=BATCH, max_particles=n_local)
    x_h, y_h = synth.mnist_like(N_ROWS)
    x, y = torch.as_tensor(x_h).to(dev), torch.as_tensor(y_h).to(dev)
    total = args.warmup + args.steps
    idx_h, sizes = synth.batch_plan(N_ROWS, BATCH, total + 24, seed=1236)       # the same batches on every rank
    idx = torch.as_tensor(idx_h).to(dev)
    allp = torch.empty((M, D), device=dev)
    engine.fill_normal(allp, SEED, _lib.STREAM_INIT, 0, 0.0, 1.0)               # prior N(0, 1) samples, same on every rank
    sweep = "jacobi" if (world > 1 or args.sweep == "jacobi") else "gauss_seidel"
    sharded = world > 1
    local = allp[row0:row0 + n_local].clone() if sharded else allp
    buf = {"all": allp, "next": torch.empty_like(allp) if (not sharded and sweep == "jacobi") else None, "local": local}
    am, av = torch.zeros((n_local, D), device=dev), torch.zeros((n_local, D), device=dev)
    loss = torch.zeros(1, device=dev)
    state = {"t": 0}
    aux = torch.cuda.Stream()
    split = sweep == "jacobi" and plan.svgd_tile_shape(n_local, M, row0) and os.environ.get("PYZ_SVGD_OVERLAP_KM", "1") == "1"
    overlap_gather = os.environ.get("PYZ_SVGD_OVERLAP_GATHER", "0") == "1"      # async all-gather: opt-in (never run on a node yet)
    # PYZ_SVGD_GATHER=p2p: every rank writes its rows straight into its peers' matrices (parallel.PeerGather) instead of the
    # RCCL all-gather -- opt-in, exercised with two processes on one GPU only
    peer = parallel.PeerGather(M, D, device=dev) if (sharded and os.environ.get("PYZ_SVGD_GATHER", "rccl").lower() == "p2p") else None
    xseq = {"n": 0}

    def one_step(s, ev=None):
        """ev: optional dict of lists that receives (start, end) event pairs per phase of this step."""
        state["t"] += 1
        main = torch.cuda.current_stream()

        def mark(stream=None):
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream if stream is not None else main)
            return e
        t_a = mark() if ev is not None else None
        work = None
        if peer is not None:
            xseq["n"] += 1
            snapshot, target, cur = peer.post(buf["local"], row0, xseq["n"]), buf["local"], buf["local"]
        elif sharded:
            work = parallel.all_gather_rows(buf["local"], buf["all"], async_op=overlap_gather)
            snapshot, target, cur = buf["all"], buf["local"], buf["local"]
        elif sweep == "jacobi":
            snapshot, target, cur = buf["all"], buf["next"], buf["all"]
        else:
            snapshot = target = cur = buf["all"]
        t_b = mark() if ev is not None else None          # (synchronous gather: it sits on the main stream between t_a and t_b)
        if split:
            if peer is not None:
                peer.wait(xseq["n"], stream=aux)
            elif work is not None:
                with torch.cuda.stream(aux):
                    work.wait()
            else:
                aux.wait_stream(main)
            k0 = mark(aux) if ev is not None else None
            plan.svgd_kernel_matrix(snapshot, row0, n_local, 1.0, stream=aux)
            done = aux.record_event()
            k1 = mark(aux) if ev is not None else None
            plan.svgd_gradients(cur, x, y, batch=sizes[s], row_idx=idx[s])
            t_c = mark() if ev is not None else None
            main.wait_event(done)
            if peer is not None:
                main.wait_event(peer.copied)
            if work is not None:
                work.wait()
            plan.svgd_combine(target, snapshot, row0, am, av, SVGD_LR, 1.0, state["t"], loss)
        else:
            k0 = k1 = None
            plan.svgd_gradients(cur, x, y, batch=sizes[s], row_idx=idx[s])
            t_c = mark() if ev is not None else None
            if peer is not None:
                peer.wait(xseq["n"])
                main.wait_event(peer.copied)
            if work is not None:
                work.wait()
            plan.svgd_sweep(target, snapshot, row0, am, av, SVGD_LR, 1.0, state["t"], loss, sweep=sweep)
        t_d = mark() if ev is not None else None
        if buf["next"] is not None:
            buf["all"], buf["next"] = buf["next"], buf["all"]
        if ev is not None:
            ev["gather"].append((t_a, t_b))
            ev["gradients"].append((t_b, t_c))
            ev["tail"].append((t_c, t_d))
            ev["step"].append((t_a, t_d))
            if k0 is not None:
                ev["kernel_matrix"].append((k0, k1))

    def steps(s0, n):
        for s in range(s0, s0 + n):
            one_step(s)

    dt = timed_region(lambda: steps(0, args.warmup), lambda: steps(args.warmup, args.steps), world, backend, dev)
    plan.check_finite()
    if peer is not None:
        peer.check()
    total_loss = loss.clone()
    parallel.sum_over_ranks(total_loss)
    step_us = dt / args.steps * 1e6

    # ---- after the timed region, independently of it: where a step's time goes on THIS rank (HIP events per phase; the
    #      gather's share shows which all-gather algorithm RCCL picked: ~33 us direct, ~233 us ring for 5 MB per rank over
    #      7 x ~153 GB/s xGMI links), and the kernels of one step with their own begin / end timestamps
    ev = {k: [] for k in ("gather", "gradients", "kernel_matrix", "tail", "step")}
    for s in range(total, total + 12):
        one_step(s, ev)
    torch.cuda.synchronize()
    phases = {k: round(float(np.median([a.elapsed_time(b) * 1e3 for a, b in v])), 1) for k, v in ev.items() if v}
    with engine.KernelProbe(512) as kp:
        for s in range(total + 12, total + 16):
            one_step(s)
    per = {n: (c / 4.0, us) for n, (c, us) in kp.by_kernel().items()}
    B, K1, N1, N2 = BATCH, DIMS[0], DIMS[1], DIMS[2]
    f_fwd = 2.0 * B * (K1 + 1) * N1 * n_local
    flop_of = {"k_dense_fwd": f_fwd, "k_dense_fwd_lds": f_fwd, "k_dense_fwd_ring": f_fwd,
               "k_wgrad_all": f_fwd + 2.0 * B * (N1 + 1) * N2 * n_local,
               "k_head_rows": (2.0 * B * (N1 + 1) * N2 + 2.0 * B * N1 * N2) * n_local}
    kern = {n: (c, us, flop_of[n.split("<")[0]]) for n, (c, us) in per.items() if n.split("<")[0] in flop_of}
    roof = {"bound": "mfma", "whole_step": whole_step(SVGD_FLOP_PER_STEP, 247e6, step_us),
            "phases_us_rank0": phases,
            "kernels_us_per_step": {n: [round(c, 2), round(us, 2)] for n, (c, us) in sorted(per.items(), key=lambda kv: -kv[1][0] * kv[1][1])}}
    if kern:
        k = max(kern, key=lambda n: kern[n][0] * kern[n][1])
        c, us_k, fl = kern[k]
        achieved = fl / (us_k * 1e-6) / 1e12
        roof.update({"kernel": k, "kernel_us": round(us_k, 3), "flop_per_launch": fl, "achieved": round(achieved, 3),
                     "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                     "traffic": None,
                     "timing": "start/stop event pair per launch (hipExtLaunchKernelGGL), 4 eager steps after the timed region"})
    all_phases = [phases]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, phases)
        all_phases = gathered
    if rank != 0:
        return None
    out = {
        "metric": "SVGD particle-grad-steps/sec, 64 particles, MLP 784->200->10, batch 1024",
        "value": round(M * args.steps / dt, 2),
        "unit": "particle-grad-steps/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 6),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 (kernel matrix f64)",
        "data": "synthetic",
        "config": {"workload": f"SVGD, {M} particles of MLP 784->200->10 (D=159010), batch 1024 replicated, {sweep} sweep, "
                               "gamma 1, prior N(0,1) start, eager launches",
                   "final_loss": round(float(total_loss.item()), 6),
                   "kernel_matrix_on_second_stream": bool(split), "async_gather": bool(overlap_gather and sharded), "exchange": "peer writes + flags (parallel.PeerGather)" if peer is not None else ("RCCL all-gather" if sharded else None),
                   "rccl_knobs_to_try": "NCCL_ALGO=Tree|Ring, NCCL_PROTO=Simple|LL|LL128, NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS: "
                                        "compare phases_us_per_rank.gather across runs",
                   "parallelism": (f"particles sharded x{args.gpus} ({n_local} per GPU), one all-gather of the (64, D) matrix "
                                   "per step" if world > 1 else "1 GPU, no collective")},
        "roofline": roof,
        "phases_us_per_rank": all_phases,
    }
    if not args.no_cpu_baseline and args.gpus == 1:
        out["cpu_baseline"] = cpu_baseline_svgd()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--method", choices=["sgld", "svgd"], default="sgld")
    ap.add_argument("--sweep", choices=["gauss_seidel", "jacobi"], default="gauss_seidel", help="svgd on one GPU")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel timing (used for the PMC passes)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 4000 if args.method == "sgld" else 100
    if args.warmup is None:
        args.warmup = 400 if args.method == "sgld" else 10

    import torch.distributed as dist
    rank, world, backend, dev = init_ranks(args)
    out = (bench_sgld if args.method == "sgld" else bench_svgd)(args, rank, world, backend, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
