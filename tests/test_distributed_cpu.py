"""world_size-2 gloo tests (CPU) of the N>1 paths: shard bookkeeping, the SVGD all-gather protocol
(Jacobi sweep on sharded rows == whole-matrix Jacobi step; arithmetic by the CPU oracle) and the
end-of-run merges of independent chains."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn):
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _svgd_sharded(rank, world):
    from bayesian_inference_for_nn_amd import parallel
    from oracle import mlp, svgd
    spec = mlp.MLPSpec((3, 4, 2), ("relu", "softmax"), "scce")
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=(10, 3)), rng.integers(0, 2, size=10)
    M, D = 4, spec.n_params
    parts = rng.normal(size=(M, D)) * 0.2
    row0, n_local = parallel.shard_range(M, world, rank)
    local = torch.as_tensor(parts[row0:row0 + n_local].copy())
    gathered = torch.zeros((M, D), dtype=torch.float64)
    m, v = np.zeros((n_local, D)), np.zeros((n_local, D))
    for t in (1, 2, 3):
        parallel.all_gather_rows(local, gathered)                       # the one exchange step
        snap = gathered.numpy().copy()
        for il in range(n_local):
            i = row0 + il
            _, g, _ = mlp.loss_and_grad(snap[i], x, y, spec)
            k, rep = svgd.rbf_row(snap, i)
            phi = (k.sum() * g + rep) / M
            new, m[il], v[il] = svgd.adam_update(snap[i], phi, m[il], v[il], t, 0.05, np.float64)
            local[il] = torch.as_tensor(new)
    parallel.all_gather_rows(local, gathered)
    # reference: the whole matrix on one rank, Jacobi sweep
    st = svgd.SVGDState(parts)
    for _ in range(3):
        svgd.svgd_step(st, x, y, spec, 0.05, sweep="jacobi")
    return float(np.abs(gathered.numpy() - st.particles).max())


def test_svgd_particle_sharding_protocol_equals_whole_matrix_jacobi():
    errs = run2(_svgd_sharded)
    assert max(errs) < 1e-12


def _merges(rank, world):
    from bayesian_inference_for_nn_amd import parallel
    assert parallel.world_info() == (rank, world)
    assert parallel.shard_range(64, world, rank) == (32 * rank, 32)
    try:
        parallel.shard_range(7, world, rank)
        bad = False
    except ValueError:
        bad = True
    rng = np.random.default_rng(rank)
    n = 10 + 5 * rank
    thetas = rng.normal(size=(n, 6))
    mean, sq, tot = parallel.merge_moment_chains(torch.as_tensor(thetas.mean(0)), torch.as_tensor((thetas ** 2).mean(0)), n)
    allt = np.concatenate([np.random.default_rng(r).normal(size=(10 + 5 * r, 6)) for r in range(world)])
    ok = np.allclose(mean.numpy(), allt.mean(0)) and np.allclose(sq.numpy(), (allt ** 2).mean(0)) and tot == len(allt)
    s, f = parallel.merge_sampled_chains([np.full(3, rank)], [rank + 1])
    tmax = parallel.max_over_ranks(1.0 + rank)
    return bool(bad and ok and [int(a[0]) for a in s] == [0, 1] and f == [1, 2] and tmax == 2.0)


def test_chain_merges_and_shard_ranges():
    assert all(run2(_merges))


def _seeds(rank, world):
    from bayesian_inference_for_nn_amd import parallel
    from bayesian_inference_for_nn_amd.datasets import Dataset
    from bayesian_inference_for_nn_amd.losses import MeanSquaredError
    fresh = parallel.shared_seed(None)            # rank 0's entropy, the same on every rank
    given = parallel.shared_seed(7 + rank)        # rank 0's argument wins
    x = np.arange(200.0).reshape(100, 2)
    ds = Dataset((x, x[:, :1]), MeanSquaredError, "Regression")     # seed=None: one split for all ranks
    return fresh, given, ds.train_data.x[:5].tolist()


def test_shared_seed_and_dataset_split_agree_across_ranks():
    a, b = run2(_seeds)
    assert a[0] == b[0] and a[1] == b[1] == 7 and a[2] == b[2]
