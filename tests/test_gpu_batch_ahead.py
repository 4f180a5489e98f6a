"""Chained (device-resident) runs assemble every batch one step ahead of its step (idle workgroups of the
weight-gradient launch; csrc/pyz_fused.h PrepArgs) and the forward pass reads the contiguous copy without waiting
for the step scalars.  These tests pin that path against the one that does neither: the same steps as single
eager `*_step` calls (in-kernel gather, Pyesian/optimizers/SGLD.py:46-95 / SGD.py:42-89 per step).  Both run the
same kernels on the same operand values, so with whole batches the results must agree BIT FOR BIT -- for every run
length around the graph-chunk boundaries, input widths that take the 16-byte and the scalar copy loop, integer labels
and float targets.  With a ragged last batch per epoch the launch geometry differs (a chained run launches for the
largest batch of the run, an eager step for its own: another split of the batch reduction over the waves of a
workgroup, i.e. another float32 summation order), so those runs are compared at 2e-6 of the largest magnitude."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mlp as o_mlp
from oracle import sgld as o_sgld


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.fixture(scope="module")
def eng(gpu_device):
    from bayesian_inference_for_nn_amd import engine
    return engine


def same(a, b, name, exact):
    if exact:
        assert np.array_equal(a, b), f"{name}: chained run and eager steps differ (max {np.abs(a - b).max():.3e})"
    else:
        scale = max(float(np.abs(b).max()), 1e-30)
        assert np.abs(a.astype(np.float64) - b).max() <= 2e-6 * scale, f"{name}: {np.abs(a - b).max():.3e} vs scale {scale:.3e}"


def batches(rng, N, B, n_steps):
    """Row table (n_steps, B) + sizes: one permutation per epoch, ragged last batch of every epoch."""
    idx = np.zeros((n_steps, B), dtype=np.int32)
    bs, s = [], 0
    while s < n_steps:
        perm = rng.permutation(N)
        for o in range(0, N, B):
            if s == n_steps:
                break
            chunk = perm[o:o + B]
            idx[s, :len(chunk)] = chunk
            bs.append(len(chunk))
            s += 1
    return idx, bs


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("k_in", [24, 23])                       # 16-byte copy loop / scalar copy loop
@pytest.mark.parametrize("n_steps", [1, 2, 3, 31, 32, 33, 67])   # around the 32-step graph chunks (+ remainder graphs)
def test_sgld_run_equals_eager_steps(eng, k_in, n_steps, ragged):
    spec = o_mlp.MLPSpec((k_in, 16, 4), ("relu", "softmax"), "scce")
    rng = np.random.default_rng(100 * k_in + n_steps)
    N, B = (150 if ragged else 192), 64                          # batches 64, 64, 22 per epoch / 64, 64, 64
    x = rng.normal(size=(N, k_in)).astype(np.float32)
    y = rng.integers(0, 4, size=N).astype(np.int32)
    theta = (rng.normal(size=spec.n_params) * 0.3).astype(np.float32)
    idx, bs = batches(rng, N, B, n_steps)
    lr_fn = o_sgld.lr_schedule(max(n_steps, 2), 0.01, 0.003, 0.99)
    lrs = [float(np.float32(lr_fn(s))) for s in range(n_steps)]
    D = spec.n_params
    plan = eng.MLPPlan(eng.MLPSpec(spec.dims, spec.acts, spec.loss), max_batch=B)
    xd, yd, idxd = dev(x), dev(y, torch.int32), dev(idx, torch.int32)
    out = []
    for chained in (True, False):
        th, mean, sq = dev(theta), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        losses = torch.zeros(n_steps, device="cuda")
        if chained:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                # two calls: the second run starts on a fresh first batch with the tables of a new run
                h = n_steps // 2
                if h:
                    plan.sgld_run(th, mean, sq, xd, yd, idxd, bs[:h], lrs[:h], 0, 7, losses, use_graph=True)
                plan.sgld_run(th, mean, sq, xd, yd, idxd, bs[h:], lrs[h:], h, 7, losses, use_graph=True, slot0=h)
            stream.synchronize()
        else:
            one = torch.zeros(1, device="cuda")
            for s in range(n_steps):
                plan.sgld_step(th, mean, sq, xd, yd, lrs[s], s, 7, one, batch=bs[s], row_idx=idxd[s, :bs[s]])
                losses[s] = one[0]
            torch.cuda.synchronize()
        out.append([t.cpu().numpy().copy() for t in (th, mean, sq, losses)])
    plan.check_finite()
    for a, b, name in zip(out[0], out[1], ("theta", "mean", "sq_mean", "losses")):
        same(a, b, name, exact=not ragged)
    plan.close()


@pytest.mark.parametrize("n_steps", [1, 5, 40])
def test_sgd_run_with_float_targets_equals_eager_steps(eng, n_steps):
    """MSE: no integer labels travel with the rows; the head gathers the targets through the row table."""
    spec = o_mlp.MLPSpec((6, 8, 2), ("tanh", "linear"), "mse")
    rng = np.random.default_rng(n_steps)
    N, B = 64, 32                                                # whole batches
    x = rng.normal(size=(N, 6)).astype(np.float32)
    y = rng.normal(size=(N, 2)).astype(np.float32)
    theta = (rng.normal(size=spec.n_params) * 0.3).astype(np.float32)
    idx, bs = batches(rng, N, B, n_steps)
    lrs = [0.01] * n_steps
    plan = eng.MLPPlan(eng.MLPSpec(spec.dims, spec.acts, spec.loss), max_batch=B)
    xd, yd, idxd = dev(x), dev(y), dev(idx, torch.int32)
    out = []
    for chained in (True, False):
        th = dev(theta)
        losses = torch.zeros(n_steps, device="cuda")
        if chained:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                plan.sgd_run(th, xd, yd, idxd, bs, lrs, losses, use_graph=True)
            stream.synchronize()
        else:
            one = torch.zeros(1, device="cuda")
            for s in range(n_steps):
                plan.sgd_step(th, xd, yd, lrs[s], one, batch=bs[s], row_idx=idxd[s, :bs[s]])
                losses[s] = one[0]
            torch.cuda.synchronize()
        out.append([t.cpu().numpy().copy() for t in (th, losses)])
    for a, b, name in zip(out[0], out[1], ("theta", "losses")):
        assert np.array_equal(a, b), f"{name}: chained run and eager steps differ (max {np.abs(a - b).max():.3e})"
    plan.close()
