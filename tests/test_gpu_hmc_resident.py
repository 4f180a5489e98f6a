"""k_hmc_resident (one launch per proposal, the row-slice workgroups of a chain exchanging their partial gradients through
an arrival counter) against k_hmc_multi (one launch per gradient evaluation): the same sums in the same order, so q,
the energies and the acceptance must agree bit for bit -- HMC.py:74-104.  And the exit condition: a workgroup that
never sees the others gives up, marks the proposal and leaves q alone."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mlp as o_mlp
from bayesian_inference_for_nn_amd import synth


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.fixture(scope="module")
def eng(gpu_device):
    from bayesian_inference_for_nn_amd import engine
    return engine


CASES = [
    # dims, acts, loss, rows, chains, L
    ((2, 50, 2), ("relu", "softmax"), "scce", 1600, 1, 20),     # C3
    ((2, 50, 2), ("relu", "softmax"), "scce", 1600, 8, 20),
    ((2, 50, 2), ("relu", "softmax"), "scce", 1600, 16, 3),
    ((2, 50, 2), ("tanh", "softmax"), "scce", 700, 3, 0),
    ((2, 50, 2), ("relu", "softmax"), "scce", 700, 3, 1),
    ((3, 5, 2), ("tanh", "linear"), "mse", 301, 2, 4),
    ((4, 30, 3), ("sigmoid", "softmax"), "scce", 1000, 5, 6),
]


def run(eng, monkeypatch, resident, case, burning, on_graph):
    dims, acts, loss, n, P, L = case
    monkeypatch.setenv("PYZ_HMC_RESIDENT", str(resident))
    spec = eng.MLPSpec(dims, acts, loss)
    D = spec.n_params
    rng = np.random.default_rng(7)
    x = rng.normal(size=(n, dims[0])).astype(np.float32)
    y = rng.normal(size=(n, dims[-1])).astype(np.float32) if loss == "mse" else rng.integers(0, dims[-1], size=n).astype(np.int32)
    qs = (rng.normal(size=(P, D)) * 0.2).astype(np.float32)
    plan = eng.MLPPlan(spec, max_batch=n, max_particles=P)
    q, stats = dev(qs), torch.zeros((P, 8), device="cuda")
    xd, yd = dev(x), dev(y, torch.float32 if loss == "mse" else torch.int32)
    us = list(rng.random(P))
    out = []
    st = torch.cuda.Stream() if on_graph else torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for step in range(3):      # three proposals in a row: the device's own Philox momentum, graph replays
            plan.hmc_step(q, xd, yd, L, 0.004, 0.5, 0.0, 1.0, us, step, 11, stats, burning=burning)
            out.append((q.clone(), stats.clone()))
    st.synchronize()
    plan.close()
    return out


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("burning", [False, True])
def test_resident_equals_one_launch_per_gradient(eng, monkeypatch, case, burning):
    a = run(eng, monkeypatch, 1, case, burning, True)
    b = run(eng, monkeypatch, 0, case, burning, True)
    for k, ((qa, sa), (qb, sb)) in enumerate(zip(a, b)):
        assert (sa[:, 7] == 0).all()
        assert torch.equal(qa, qb), f"proposal {k}: q differs (max {float((qa - qb).abs().max()):.3e})"
        assert torch.equal(sa, sb), f"proposal {k}: stats differ\n{sa}\n{sb}"
    if not burning:
        assert any(float(s[:, 0].sum()) > 0 for _, s in a), "no proposal accepted: the comparison would not see q move"


def test_resident_eager_launch_equals_graph_replay(eng, monkeypatch):
    a = run(eng, monkeypatch, 1, CASES[1], False, True)
    monkeypatch.setenv("PYZ_HMC_GRAPH", "0")
    b = run(eng, monkeypatch, 1, CASES[1], False, True)
    for (qa, sa), (qb, sb) in zip(a, b):
        assert torch.equal(qa, qb) and torch.equal(sa, sb)


def test_a_workgroup_that_never_sees_the_others_gives_up(eng, monkeypatch):
    """PYZ_HMC_SPIN_LIMIT = -1: the first poll that does not find every slice gives up.  At most one workgroup per
    evaluation (the last to arrive) gets through, so the proposal cannot complete: it is marked (stats[7] = -1), q
    stays, and the launch returns."""
    monkeypatch.setenv("PYZ_HMC_SPIN_LIMIT", "-1")
    dims, acts, loss, n, P, L = CASES[0]
    spec = eng.MLPSpec(dims, acts, loss)
    xm, ym = synth.moons(2000)
    plan = eng.MLPPlan(spec, max_batch=n, max_particles=P)
    q0 = (np.random.default_rng(1).normal(size=(P, spec.n_params)) * 0.2).astype(np.float32)
    q, stats = dev(q0), torch.zeros((P, 8), device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        plan.hmc_step(q, dev(xm[:n]), dev(ym[:n], torch.int32), L, 0.005, 0.5, 0.0, 1.0, [0.5], 0, 3, stats, burning=True)
    st.synchronize()
    assert float(stats[0, 7]) == -1.0
    assert np.array_equal(q.cpu().numpy(), q0)
    plan.close()
