"""The Gauss-Seidel sweep (the reference's order, SVGD.py:100-123) as ONE resident launch (k_svgd_gs_resident: the matrix in
registers for the whole sweep, per particle a two-hop exchange of tagged granules) against one launch per particle
(k_svgd_gs, rows ascending): bit for bit -- particles, Adam moments, loss -- over several steps, on shapes whose workgroups
reduce several columns each (D small) and at the BASELINE C5 shape; a sweep whose workgroups cannot meet reports itself.
Both resident kernels: k_svgd_gs_resident (mode 1, the default) and k_svgd_gs_resident2 (mode 2: the partials of row i + 2 through
the reducers one step early, the one distance that depends on the update of row i summed by every workgroup in one hop)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mlp as o_mlp

from bayesian_inference_for_nn_amd import synth

MNIST = o_mlp.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
WIDE3 = o_mlp.MLPSpec((64, 40, 24, 10), ("relu", "relu", "softmax"), "scce")     # D = 3 834: five workgroups, ragged last one
TINY = o_mlp.MLPSpec((5, 7, 3), ("tanh", "softmax"), "scce")                     # D = 66: one workgroup reduces every column


@pytest.fixture(scope="module")
def eng(gpu_device):
    from bayesian_inference_for_nn_amd import engine
    return engine


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def _run(eng, monkeypatch, resident, spec, M, parts, x, y, steps, lr, batch=None, idx=None):
    monkeypatch.setenv("PYZ_SVGD_GS_FUSED", "1")
    monkeypatch.setenv("PYZ_SVGD_GS_ZIGZAG", "0")
    monkeypatch.setenv("PYZ_SVGD_GS_RESIDENT", str(int(resident)))   # 0: a launch per particle, 1 / 2: the resident kernels
    D = spec.n_params
    plan = eng.MLPPlan(eng.MLPSpec(spec.dims, spec.acts, spec.loss), max_batch=len(x) if batch is None else batch, max_particles=M)
    p, am, av = dev(parts), torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    loss, xd, yd = torch.zeros(1, device="cuda"), dev(x), dev(y, torch.int32)
    losses, names = [], set()
    for t in range(1, steps + 1):
        with eng.KernelProbe(80) as kp:
            if idx is None:
                plan.svgd_step(p, p, 0, am, av, xd, yd, lr, 1.0, t, loss, sweep="gauss_seidel")
            else:
                plan.svgd_step(p, p, 0, am, av, xd, yd, lr, 1.0, t, loss, sweep="gauss_seidel", batch=batch, row_idx=dev(idx, torch.int32))
        names |= {n for n, _ in kp.launches}
        losses.append(float(loss))
    torch.cuda.synchronize()
    plan.check_finite()
    plan.close()
    return p.cpu().numpy(), am.cpu().numpy(), av.cpu().numpy(), losses, names


def _resident_kernel(names, mode):
    """The resident kernel of `mode` ran (1: k_svgd_gs_resident, 2: k_svgd_gs_resident2, distances one step early)."""
    want = "k_svgd_gs_resident2" if mode == 2 else "k_svgd_gs_resident"
    return any(n.split("(")[0].strip() == want for n in names)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case", ["tiny_m1", "tiny_m2", "tiny_m5", "wide3_m7", "wide3_m9", "wide3_m20", "wide3_m64"])
def test_resident_sweep_equals_one_launch_per_particle(eng, monkeypatch, case, mode):
    spec, M = {"tiny_m1": (TINY, 1), "tiny_m2": (TINY, 2), "tiny_m5": (TINY, 5), "wide3_m7": (WIDE3, 7), "wide3_m9": (WIDE3, 9),
               "wide3_m20": (WIDE3, 20), "wide3_m64": (WIDE3, 64)}[case]
    rng = np.random.default_rng(len(case))
    n = 90
    x = rng.normal(size=(n, spec.dims[0])).astype(np.float32)
    y = rng.integers(0, spec.dims[-1], size=n).astype(np.int32)
    parts = (rng.normal(size=(M, spec.n_params)) * (0.015 if spec is WIDE3 else 0.1)).astype(np.float32)   # K_ij well away from 0 and 1
    a = _run(eng, monkeypatch, mode, spec, M, parts, x, y, 3, 1e-3)
    b = _run(eng, monkeypatch, 0, spec, M, parts, x, y, 3, 1e-3)
    assert _resident_kernel(a[4], mode), a[4]
    assert not any(n.startswith("k_svgd_gs_resident") for n in b[4]) and any(n.startswith("k_svgd_gs") for n in b[4]), b[4]
    for u, v, what in zip(a[:3], b[:3], ("particles", "adam m", "adam v")):
        assert np.array_equal(u, v), (case, what, float(np.abs(u - v).max()))
    assert a[3] == b[3]
    assert not np.array_equal(a[0], parts)


@pytest.mark.parametrize("mode", [1, 2])
def test_resident_sweep_at_c5(eng, monkeypatch, mode):
    spec, M, B = MNIST, 64, 1024
    x, y = synth.mnist_like(2048)
    rng = np.random.default_rng(17)
    idx = rng.permutation(2048)[:B].astype(np.int32)
    parts = (synth.glorot_uniform(spec.dims)[None, :] + 1e-3 * rng.normal(size=(M, spec.n_params))).astype(np.float32)   # K_ij ~ 0.7
    a = _run(eng, monkeypatch, mode, spec, M, parts, x, y, 2, 0.01, batch=B, idx=idx)
    b = _run(eng, monkeypatch, 0, spec, M, parts, x, y, 2, 0.01, batch=B, idx=idx)
    assert _resident_kernel(a[4], mode), a[4]
    for u, v, what in zip(a[:3], b[:3], ("particles", "adam m", "adam v")):
        assert np.array_equal(u, v), (what, float(np.abs(u - v).max()))
    assert a[3] == b[3]


@pytest.mark.parametrize("mode", [1, 2])
def test_a_sweep_that_cannot_meet_reports_itself(eng, monkeypatch, mode):
    """A poll limit of -1: every poll that does not find its granules at once gives up -- the exit path of a grid that is not
    resident together.  The grid drains, the step's loss is NaN and the plan's sentinel counts it."""
    from bayesian_inference_for_nn_amd._lib import PyzError
    monkeypatch.setenv("PYZ_SVGD_GS_FUSED", "1")
    monkeypatch.setenv("PYZ_SVGD_GS_RESIDENT", str(mode))
    monkeypatch.setenv("PYZ_SVGD_GS_SPIN_LIMIT", "-1")
    spec, M = WIDE3, 16
    rng = np.random.default_rng(3)
    x = rng.normal(size=(50, 64)).astype(np.float32)
    y = rng.integers(0, 10, size=50).astype(np.int32)
    D = spec.n_params
    plan = eng.MLPPlan(eng.MLPSpec(spec.dims, spec.acts, spec.loss), max_batch=50, max_particles=M)
    p = dev((rng.normal(size=(M, D)) * 0.015).astype(np.float32))
    am, av, loss = torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda"), torch.zeros(1, device="cuda")
    plan.svgd_step(p, p, 0, am, av, dev(x), dev(y, torch.int32), 1e-3, 1.0, 1, loss, sweep="gauss_seidel")
    torch.cuda.synchronize()
    assert np.isnan(float(loss))
    with pytest.raises(PyzError):
        plan.check_finite()
    # and the next sweep, with the limit back, is a regular one (tags of the abandoned sweep are stale)
    monkeypatch.setenv("PYZ_SVGD_GS_SPIN_LIMIT", str(1 << 20))
    p2 = dev((rng.normal(size=(M, D)) * 0.015).astype(np.float32))
    am.zero_(); av.zero_()
    plan.svgd_step(p2, p2, 0, am, av, dev(x), dev(y, torch.int32), 1e-3, 1.0, 1, loss, sweep="gauss_seidel")
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    plan.check_finite()
    plan.close()
