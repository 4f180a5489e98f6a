"""The sharded / split forms of the SVGD step (SVGD.py:54-68,100-129; include/pyz.h: pyz_svgd_kernel_matrix, pyz_svgd_combine):

  * kernel matrix + combine == pyz_svgd_sweep bit for bit, on one stream and with the kernel matrix built on a second
    stream WHILE the gradient pass runs (what SVGD.step and bench.py --method svgd do);
  * the plan refuses to combine what it does not hold (another entry point used its buffers in between);
  * BASELINE.json configs[4] as one rank of eight sees it: rows [56, 64) of 64 particles at D = 159 010, batch 1024,
    against the same rows of the whole-matrix step (the 8-particle gradient pass runs other kernels than the
    64-particle one -- k_dense_fwd_ring / four-wave weight gradients -- so the comparison is to float32 rounding of phi,
    not bit for bit) and against the float64 oracle;
  * the sharded branch of SVGD.step under backend "nccl" (RCCL) in a world of ONE rank: the collective, its stream and the
    handle's wait are the real ones, synchronous and asynchronous (overlap_gather=True)."""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mlp as o_mlp
from oracle import svgd as o_svgd

from bayesian_inference_for_nn_amd import synth
from svgd_checks import lr_t, strict_particle_check

MNIST = o_mlp.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
WIDE3 = o_mlp.MLPSpec((64, 40, 24, 10), ("relu", "relu", "softmax"), "scce")


def close(gpu, ref, rel=1e-4, what=""):
    gpu = np.asarray(gpu.detach().cpu().numpy() if hasattr(gpu, "detach") else gpu, dtype=np.float64)
    ref = np.asarray(ref.detach().cpu().numpy() if hasattr(ref, "detach") else ref, dtype=np.float64)
    assert gpu.shape == ref.shape, (what, gpu.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(gpu - ref).max()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e})"


@pytest.fixture(scope="module")
def eng(gpu_device):
    from bayesian_inference_for_nn_amd import engine
    return engine


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def espec(eng, spec):
    return eng.MLPSpec(spec.dims, spec.acts, spec.loss)


def _wide3_case(M=16, n=130, seed=91):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, 64)).astype(np.float32)
    y = rng.integers(0, 10, size=n).astype(np.int32)
    parts = (rng.normal(size=(M, WIDE3.n_params)) * 0.015).astype(np.float32)
    return x, y, parts


@pytest.mark.parametrize("gamma", [1.0, "median"])
def test_kernel_matrix_plus_combine_equals_sweep(eng, gamma):
    x, y, parts = _wide3_case()
    M, D, n = parts.shape[0], WIDE3.n_params, len(x)
    plan = eng.MLPPlan(espec(eng, WIDE3), max_batch=n, max_particles=8)
    xd, yd = dev(x), dev(y, torch.int32)
    snap = dev(parts)
    row0, nl, lr, t = 8, 8, 1e-3, 3
    outs = []
    for form in ("sweep", "split", "two_streams"):
        local = snap[row0:row0 + nl].clone()
        am, av = torch.full((nl, D), 0.01, device="cuda"), torch.full((nl, D), 0.02, device="cuda")
        loss = torch.zeros(1, device="cuda")
        if form == "sweep":
            plan.svgd_gradients(local, xd, yd)
            plan.svgd_sweep(local, snap, row0, am, av, lr, gamma, t, loss, sweep="jacobi")
        elif form == "split":
            plan.svgd_gradients(local, xd, yd)
            plan.svgd_kernel_matrix(snap, row0, nl, gamma)
            plan.svgd_combine(local, snap, row0, am, av, lr, gamma, t, loss)
        else:
            main, aux = torch.cuda.current_stream(), torch.cuda.Stream()
            aux.wait_stream(main)
            plan.svgd_kernel_matrix(snap, row0, nl, gamma, stream=aux)
            done = aux.record_event()
            plan.svgd_gradients(local, xd, yd)
            main.wait_event(done)
            out = torch.empty_like(local)            # (the combine only writes its output rows)
            plan.svgd_combine(out, snap, row0, am, av, lr, gamma, t, loss)
            local = out
        torch.cuda.synchronize()
        outs.append((local.clone(), am.clone(), av.clone(), loss.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    plan.close()


def test_combine_refuses_what_the_plan_does_not_hold(eng):
    from bayesian_inference_for_nn_amd._lib import PyzError
    x, y, parts = _wide3_case(M=8)
    D, n = WIDE3.n_params, len(x)
    plan = eng.MLPPlan(espec(eng, WIDE3), max_batch=n, max_particles=8)
    xd, yd, snap = dev(x), dev(y, torch.int32), dev(parts)
    local, am, av, loss = snap.clone(), torch.zeros((8, D), device="cuda"), torch.zeros((8, D), device="cuda"), torch.zeros(1, device="cuda")
    with pytest.raises(PyzError):                      # no gradients in the plan at all
        plan.svgd_sweep(local, snap, 0, am, av, 1e-3, 1.0, 1, loss, sweep="jacobi")
    plan.svgd_gradients(local, xd, yd)
    with pytest.raises(PyzError):                      # no kernel matrix of this snapshot
        plan.svgd_combine(local, snap, 0, am, av, 1e-3, 1.0, 1, loss)
    plan.svgd_kernel_matrix(snap, 0, 8, 1.0)
    with pytest.raises(PyzError):                      # ... of another bandwidth
        plan.svgd_combine(local, snap, 0, am, av, 1e-3, 2.0, 1, loss)
    plan.svgd_kernel_matrix(snap, 0, 8, 1.0)
    # an HMC proposal on the same plan takes the gradient buffer: the sweep must not consume its contents
    q, stats = torch.zeros((8, D), device="cuda"), torch.zeros((8, 8), device="cuda")
    plan.hmc_step(q, xd, yd, 1, 0.01, 1.0, 0.0, 1.0, np.full(8, 0.5), 1, 3, stats)
    with pytest.raises(PyzError):
        plan.svgd_combine(local, snap, 0, am, av, 1e-3, 1.0, 1, loss)
    with pytest.raises(PyzError):                      # shapes the all-rows-at-once kernels do not take
        plan.svgd_kernel_matrix(snap, 0, 6, 1.0)
    plan.svgd_gradients(local, xd, yd)                 # and the regular order still works
    plan.svgd_kernel_matrix(snap, 0, 8, 1.0)
    plan.svgd_combine(local, snap, 0, am, av, 1e-3, 1.0, 1, loss)
    torch.cuda.synchronize()
    plan.close()


def test_c5_one_rank_of_eight(eng):
    """Rows [56, 64) of the 64 particles of 784 -> 200 -> 10 (D = 159 010), batch 1024 through row indices: the shard's step
    (kernel matrix on a second stream beside the gradient pass, then the combine) against the same rows of the
    whole-matrix Jacobi step and against the oracle."""
    spec, M, B, lr, row0, nl = MNIST, 64, 1024, 0.01, 56, 8
    D = spec.n_params
    x, y = synth.mnist_like(2048)
    rng = np.random.default_rng(73)
    idx = rng.permutation(2048)[:B].astype(np.int32)
    parts = (synth.glorot_uniform(spec.dims)[None, :] + 1e-3 * rng.normal(size=(M, D))).astype(np.float32)   # K_ij ~ 0.7
    xd, yd, idxd = dev(x), dev(y, torch.int32), dev(idx, torch.int32)
    snap = dev(parts)
    # the whole matrix on one GPU
    whole_plan = eng.MLPPlan(espec(eng, spec), max_batch=B, max_particles=M)
    wp, wm, wv = torch.empty_like(snap), torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    wl = torch.zeros(1, device="cuda")
    whole_plan.svgd_step(wp, snap, 0, wm, wv, xd, yd, lr, 1.0, 1, wl, sweep="jacobi", batch=B, row_idx=idxd)
    # one rank of eight
    plan = eng.MLPPlan(espec(eng, spec), max_batch=B, max_particles=nl)
    local = snap[row0:row0 + nl].clone()
    am, av, loss = torch.zeros((nl, D), device="cuda"), torch.zeros((nl, D), device="cuda"), torch.zeros(1, device="cuda")
    main, aux = torch.cuda.current_stream(), torch.cuda.Stream()
    aux.wait_stream(main)
    plan.svgd_kernel_matrix(snap, row0, nl, 1.0, stream=aux)
    done = aux.record_event()
    with eng.KernelProbe(16) as kp:
        plan.svgd_gradients(local, xd, yd, batch=B, row_idx=idxd)
    assert any(n.startswith("k_dense_fwd_ring") for n, _ in kp.launches), kp.launches
    main.wait_event(done)
    plan.svgd_combine(local, snap, row0, am, av, lr, 1.0, 1, loss)
    torch.cuda.synchronize()
    # m = 0.1 phi, v = 0.001 phi^2: phi itself, element-wise, against the whole-matrix step (float32 rounding of the
    # gradients: other summation orders) and against the oracle
    close(am, wm[row0:row0 + nl], rel=2e-5, what="adam m: shard vs whole")
    close(av, wv[row0:row0 + nl], rel=4e-5, what="adam v: shard vs whole")
    st = o_svgd.SVGDState(parts)
    out = o_svgd.svgd_step(st, x[idx], y[idx], spec, lr, 1.0, sweep="jacobi")
    close(am, st.m[row0:row0 + nl], rel=1e-4, what="adam m vs oracle")
    close(loss, [out["losses"][row0:row0 + nl].sum() / M], rel=1e-4, what="this rank's share of the loss")
    sub = o_svgd.SVGDState(parts[row0:row0 + nl])      # (a state holding the shard's rows, for the particle check)
    sub.particles = st.particles[row0:row0 + nl]
    strict_particle_check(local, sub, [out["phi"][row0:row0 + nl]], [lr_t(lr, 1)], "shard vs oracle")
    whole_plan.close()
    plan.close()


@pytest.mark.parametrize("case", ["c5_rank7_of_8", "small_gamma", "small_median", "small_pairwise"])
def test_kernel_matrix_from_group_sums_equals_the_local_pass(eng, case, monkeypatch):
    """The distance pass split over the ELEMENTS (pyz_svgd_gram_groups on slices of the groups, as ranks of a world of 2, 4
    or 8 would, then pyz_svgd_kernel_matrix_groups) against pyz_svgd_kernel_matrix on the same snapshot: bit for bit through
    the combine, whatever the split of the groups."""
    if case == "c5_rank7_of_8":
        spec, M, row0, nl, gamma = MNIST, 64, 56, 8, 1.0
        x, y = synth.mnist_like(256)
        rng = np.random.default_rng(5)
        parts = (synth.glorot_uniform(spec.dims)[None, :] + 1e-3 * rng.normal(size=(M, spec.n_params))).astype(np.float32)
        splits = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8)]
    else:
        spec, row0, nl = WIDE3, 4, 8
        x, y, parts = _wide3_case(M=20)
        M = 20
        gamma = "median" if case == "small_median" else 1.0
        splits = [(0, 4), (4, 8)] if case != "small_pairwise" else [(0, 2), (2, 4), (4, 6), (6, 8)]
        if case == "small_pairwise":
            monkeypatch.setenv("PYZ_SVGD_GRAM", "0")
    D, n = spec.n_params, len(x)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=nl)
    xd, yd, snap = dev(x), dev(y, torch.int32), dev(parts)
    outs = []
    for form in ("local", "groups"):
        local = snap[row0:row0 + nl].clone()
        am, av = torch.full((nl, D), 0.01, device="cuda"), torch.full((nl, D), 0.02, device="cuda")
        loss = torch.zeros(1, device="cuda")
        plan.svgd_gradients(local, xd, yd)
        if form == "local":
            plan.svgd_kernel_matrix(snap, row0, nl, gamma)
        else:
            groups = torch.full((eng.SVGD_GROUPS, 64 * 64), float("nan"), dtype=torch.float64, device="cuda")
            for g_lo, g_hi in reversed(splits):          # (any order: every call fills its own groups)
                plan.svgd_gram_groups(snap, g_lo, g_hi, groups)
            torch.cuda.synchronize()
            assert torch.isfinite(groups.view(8, 64, 64)[:, :M, :M]).all()
            plan.svgd_kernel_matrix_groups(groups, snap, row0, nl, gamma)
        plan.svgd_combine(local, snap, row0, am, av, 1e-3, gamma, 2, loss)
        torch.cuda.synchronize()
        outs.append((local.clone(), am.clone(), av.clone(), loss.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    plan.close()


# ------------------------------------------------------------------ RCCL in a world of one rank
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_world_of_one(rank, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from bayesian_inference_for_nn_amd import synth as sy
        from bayesian_inference_for_nn_amd.datasets import Dataset
        from bayesian_inference_for_nn_amd.distributions import GaussianPrior
        from bayesian_inference_for_nn_amd.losses import SparseCategoricalCrossentropy
        from bayesian_inference_for_nn_amd.nn import sequential_json
        from bayesian_inference_for_nn_amd.optimizers import SVGD
        from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters
        cfg = sequential_json(2, [16, 2], ["relu", "softmax"])
        x, y = sy.moons(500, seed=42)
        ds = Dataset((x, y), SparseCategoricalCrossentropy, "Classification", seed=5)
        runs = {}
        for name, kw in (("whole", dict(shard=False, sweep="jacobi")),
                         ("gather_sync", dict(_force_sharded=True, overlap_gather=False)),
                         ("gather_async", dict(_force_sharded=True, overlap_gather=True)),
                         ("gather_sync_sharded_gram", dict(_force_sharded=True, overlap_gather=False, shard_gram=True)),
                         ("gather_async_one_stream", dict(_force_sharded=True, overlap_gather=True, overlap_kernel_matrix=False))):
            opt = SVGD()
            opt.compile(HyperParameters(lr=0.05, M=8, batch_size=100), cfg, ds, verbose=False, prior=GaussianPrior(0.0, 0.3),
                        seed=77, **kw)
            for _ in range(12):                      # (crosses a recording step: the loss collective of SVGD.py:137-139)
                opt.step()
            ens, tl, _ = opt.result()
            runs[name] = (np.stack([m.weights_flat for m in ens]), [float(v) for v in tl], opt._sharded)
        ret["sharded_flags"] = [runs[k][2] for k in runs]
        ret["diff"] = {k: float(np.abs(runs[k][0] - runs["whole"][0]).max()) for k in runs}
        ret["loss_diff"] = {k: float(np.abs(np.asarray(runs[k][1]) - np.asarray(runs["whole"][1])).max()) for k in runs}
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_sharded_step_over_rccl_in_a_world_of_one(gpu_device):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_nccl_world_of_one, args=(_free_port(), ret), nprocs=1, join=True)
    assert ret["backend"] == "nccl" and ret["sharded_flags"] == [False, True, True, True, True]
    # gather (synchronous, or asynchronous on RCCL's stream with the wait in front of its first reader), kernel matrix on a
    # second stream, combine: the particles of the unsharded Jacobi run, bit for bit
    assert all(v == 0.0 for v in ret["diff"].values()), dict(ret["diff"])
    assert all(v == 0.0 for v in ret["loss_diff"].values()), dict(ret["loss_diff"])
