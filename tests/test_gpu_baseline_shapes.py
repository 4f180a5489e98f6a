"""The BASELINE.json configurations at their FULL shapes against the float64 oracle (through the C-ABI):

  C1  SGD linreg 1 -> 1, the whole training split as one batch (480 of 600 rows) and batch 600
  C2  SGLD 784 -> 200 -> 10, batch 1024 with the ragged 896: single steps (fused epilogue + device Philox at
      D = 159 010) and a 70-step device-resident run replayed as hipGraphs
  C3  HMC moons 2 -> 50 -> 2, N = 1600, L = 20, 8 chains, on the sliced and the one-workgroup-per-chain paths
  C4  BBB 784 -> 400 -> 400 -> 10, batch 1024, plus the validation forward over 6 000 rows
  C5  SVGD with M = 64 particles: both sweeps and every kernel path on a model whose D spans several
      workgroups with particles close enough to couple (K != I), and one step at the real D = 159 010

Tolerances: float32 kernels vs the float64 oracle.  Losses / costs / energies 1e-4 relative (north_star);
element-wise state after ONE step 1e-5 of its max norm; after many steps 1e-4."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import bbb as o_bbb
from oracle import hmc as o_hmc
from oracle import mlp as o_mlp
from oracle import philox as o_philox
from oracle import sgd as o_sgd
from oracle import sgld as o_sgld
from oracle import svgd as o_svgd

from bayesian_inference_for_nn_amd import synth


def close(gpu, ref, rel=1e-4, what=""):
    gpu = np.asarray(gpu.detach().cpu().numpy() if hasattr(gpu, "detach") else gpu, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
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


MNIST = o_mlp.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
MNIST_BBB = o_mlp.MLPSpec((784, 400, 400, 10), ("relu", "relu", "softmax"), "scce")
MOONS = o_mlp.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce")
WIDE3 = o_mlp.MLPSpec((64, 40, 24, 10), ("relu", "relu", "softmax"), "scce")


# ------------------------------------------------------------------ C1
@pytest.mark.parametrize("batch", [480, 600])
def test_c1_sgd_whole_set_batches(eng, batch):
    """simple_regression_example.py: y = 2x + 2; BASELINE.json quotes 'batch 600' (the whole set), the script's
    training split is 480 rows."""
    spec = o_mlp.MLPSpec((1, 1), ("linear",), "mse")
    x, y = synth.linreg(600)
    st = o_sgd.SGDState(np.array([0.3, -0.1], dtype=np.float32))
    plan = eng.MLPPlan(espec(eng, spec), max_batch=600)
    th, loss = dev(st.theta), torch.zeros(1, device="cuda")
    xd, yd = dev(x), dev(y)
    for s in range(40):
        plan.sgd_step(th, xd, yd, 1e-3, loss, batch=batch)
        rl, _ = o_sgd.sgd_step(st, x[:batch], y[:batch], spec, 1e-3)
        close(loss, [rl], what=f"loss step {s}")
        if s == 0:
            close(th, st.theta, rel=1e-5, what="theta after one step")
    close(th, st.theta, what="theta")
    plan.close()


# ------------------------------------------------------------------ C2
def _c2_data(n_rows):
    x, y = synth.mnist_like(n_rows)
    return x, y, synth.glorot_uniform(MNIST.dims)


def test_c2_sgld_single_steps_full_shape(eng):
    """Three pyz_sgld_step calls at 784 -> 200 -> 10: batch 1024, the ragged 896, 1024 again; the update runs in
    the weight-gradient epilogue with noise from the device Philox stream (159 010 elements)."""
    x, y, theta0 = _c2_data(4096)
    D = MNIST.n_params
    lr = o_sgld.lr_schedule(10_000, 0.01, 0.003, 0.99)
    plan = eng.MLPPlan(espec(eng, MNIST), max_batch=1024)
    st = o_sgld.SGLDState(theta0)
    th, mean, sq = dev(theta0), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    loss = torch.zeros(1, device="cuda")
    xd, yd = dev(x), dev(y, torch.int32)
    rng = np.random.default_rng(5)
    for s, b in enumerate((1024, 896, 1024)):
        idx = rng.permutation(4096)[:b].astype(np.int32)
        plan.sgld_step(th, mean, sq, xd, yd, float(np.float32(lr(s))), s, 2024, loss, batch=b, row_idx=dev(idx, torch.int32))
        rl, _ = o_sgld.sgld_step(st, x[idx], y[idx], MNIST, float(np.float32(lr(s))), o_philox.normal(2024, 0, s, D))
        close(loss, [rl], what=f"loss {s}")
        tol = 1e-5 if s == 0 else 1e-4
        close(th, st.theta, rel=tol, what=f"theta {s}")
        close(mean, st.mean, rel=tol, what=f"mean {s}")
        close(sq, st.sq_mean, rel=tol, what=f"sq_mean {s}")
    plan.close()


@pytest.mark.parametrize("n_steps,ksplit", [(70, 0), (20, 0), (70, 8), (20, 4)])
def test_c2_sgld_graph_run_full_shape(eng, monkeypatch, n_steps, ksplit):
    """pyz_sgld_run(use_graph=1) at the bench shape: 2 944 rows -> batches 1024, 1024, 896 per epoch.  70 steps =
    two replays of the 32-step graph + a 6-step remainder graph; 20 steps = shorter than one chunk (the driver's
    bench invocation): one graph of exactly 20 steps, its tables carried by the launch that sets the scalars.
    ksplit > 0: the opt-in split-reduction forward (k_dense_fwd_ring cut into `ksplit` ranges of slabs per row block, the
    head adds the partial sums; measured slower at this shape and off by default -- same results to float32 rounding)."""
    monkeypatch.setenv("PYZ_FWD_KSPLIT", str(ksplit))
    n_rows = 2944
    x, y, theta0 = _c2_data(n_rows)
    D = MNIST.n_params
    idx, sizes = synth.batch_plan(n_rows, 1024, n_steps, seed=77)
    assert 896 in sizes
    lrs = synth.sgld_lr_table(n_steps, 0.01, 0.003, 0.99, 0, n_steps)
    plan = eng.MLPPlan(espec(eng, MNIST), max_batch=1024)
    th, mean, sq = dev(theta0), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    losses = torch.zeros(n_steps, device="cuda")
    stream = torch.cuda.Stream()
    xd, yd, idxd = dev(x), dev(y, torch.int32), dev(idx, torch.int32)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        plan.sgld_run(th, mean, sq, xd, yd, idxd, sizes, list(lrs), 0, 99, losses, use_graph=True)
    stream.synchronize()
    st = o_sgld.SGLDState(theta0)
    ref = []
    for s in range(n_steps):
        rows = idx[s, :sizes[s]]
        rl, _ = o_sgld.sgld_step(st, x[rows], y[rows], MNIST, float(lrs[s]), o_philox.normal(99, 0, s, D))
        ref.append(rl)
    close(losses, ref, what="losses")
    close(th, st.theta, what="theta")
    close(mean, st.mean, what="mean")
    close(sq, st.sq_mean, what="sq_mean")
    assert plan.last_run_path() == ("graph", n_steps)       # every step ran inside a replayed graph
    assert plan.last_run_graph_launches() == (3 if n_steps == 70 else 1)
    # which forward the run's steps launch (a probe makes the same run eager)
    with torch.cuda.stream(stream):
        with eng.KernelProbe(32) as kp:
            plan.sgld_run(th, mean, sq, xd, yd, idxd, sizes[:2], list(lrs[:2]), 0, 99, losses, use_graph=True)
    stream.synchronize()
    assert any(n.startswith("k_dense_fwd_ring") for n, _ in kp.launches) == (ksplit > 0), kp.launches
    plan.close()


# ------------------------------------------------------------------ C3
@pytest.mark.parametrize("multi", [1, 0])
def test_c3_hmc_moons_full_shape(eng, monkeypatch, multi):
    """N = 1600 rows, L = 20, 8 chains, epsilon 0.005, m 0.5 (HMC_classification.py:135-136): multi = 1 row slices
    over 16 workgroups per chain and one launch per gradient evaluation (k_hmc_multi), multi = 0 one workgroup per
    chain with the whole data set in LDS (k_hmc_fused)."""
    monkeypatch.setenv("PYZ_HMC_MULTI", str(multi))
    xm, ym = synth.moons(2000)
    x, y = xm[:1600], ym[:1600]
    D, P, L, eps_, m = MOONS.n_params, 8, 20, 0.005, 0.5
    rng = np.random.default_rng(43)
    qs = (rng.normal(size=(P, D)) * 0.2).astype(np.float32)
    zs = rng.normal(size=(P, D)).astype(np.float32)
    refs = [o_hmc.hmc_step(qs[c], zs[c], x, y, MOONS, 0.0, 1.0, L, eps_, m, u=0.5) for c in range(P)]
    us = []   # uniforms away from the acceptance threshold: float32 rounding cannot flip the decision
    for c, r in enumerate(refs):
        ratio = np.exp(min(r["log_ratio"], 50.0))
        us.append(0.5 * ratio if c % 2 == 0 else min(2.0 * ratio + 0.1, 1e30))
    refs = [o_hmc.hmc_step(qs[c], zs[c], x, y, MOONS, 0.0, 1.0, L, eps_, m, u=us[c]) for c in range(P)]
    plan = eng.MLPPlan(espec(eng, MOONS), max_batch=1600, max_particles=P)
    q, stats = dev(qs), torch.zeros((P, 8), device="cuda")
    side = torch.cuda.Stream()
    xd, yd, zd = dev(x), dev(y, torch.int32), dev(zs)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        plan.hmc_step(q, xd, yd, L, eps_, m, 0.0, 1.0, us, 0, 1, stats, unit_p=zd)
    side.synchronize()
    s = stats.cpu().numpy()
    for c, r in enumerate(refs):
        assert bool(s[c, 0]) == r["accepted"], (c, s[c], r["log_ratio"])
        close(q[c], r["q"], what=f"q[{c}]")
        for k, key in ((2, "U0"), (3, "K0"), (4, "U1"), (5, "K1")):
            assert abs(s[c, k] - r[key]) <= 1e-4 * abs(r[key]) + 1e-4, (key, s[c, k], r[key])
        assert abs(s[c, 6] - r["log_ratio"]) <= 2e-4 * max(abs(r["U0"]), abs(r["K0"]), abs(r["U1"]), abs(r["K1"]), 1.0)
    # the proposals themselves (burning accepts every chain)
    q2 = dev(qs)
    with torch.cuda.stream(side):
        plan.hmc_step(q2, xd, yd, L, eps_, m, 0.0, 1.0, [0.0] * P, 0, 1, stats, unit_p=zd, burning=True)
    side.synchronize()
    for c, r in enumerate(refs):
        close(q2[c], r["q_proposed"], what=f"proposal[{c}]")
    plan.close()


# ------------------------------------------------------------------ C4
@pytest.mark.parametrize("init", ["trained_like", "as_compiled"])
def test_c4_bbb_full_shape(eng, init):
    """Two pyz_bbb_step calls at 784 -> 400 -> 400 -> 10 (D = 478 410), batch 1024 then 896, then the validation
    forward of BBB.py:203-209 over 6 000 rows with the sampled weights.  'as_compiled' starts where
    BBB.compile_extra_components leaves the posterior for BBB_mnist.py's prior: mu = 0, rho = 1."""
    spec = MNIST_BBB
    D = spec.n_params
    x, y = synth.mnist_like(8192)
    xv, yv = synth.mnist_like(6000, seed=1235)
    rng = np.random.default_rng(17)
    if init == "trained_like":
        mu0 = synth.glorot_uniform(spec.dims)
        rho0 = (rng.normal(size=D) * 0.3 - 4.0).astype(np.float32)
    else:
        mu0, rho0 = np.zeros(D, np.float32), np.ones(D, np.float32)
    pm, pr = o_bbb.mix_prior(0.0, 1.0)
    lr, alpha = 5e-4, 0.3
    plan = eng.MLPPlan(espec(eng, spec), max_batch=6000)
    mu, rho, w, cost = dev(mu0), dev(rho0), torch.zeros(D, device="cuda"), torch.zeros(4, device="cuda")
    xd, yd = dev(x), dev(y, torch.int32)
    rmu, rrho = mu0.astype(np.float64), rho0.astype(np.float64)
    for s, b in ((1, 1024), (2, 896)):
        idx = rng.permutation(8192)[:b].astype(np.int32)
        eps = o_philox.normal(2024, 1, s, D)
        plan.bbb_step(mu, rho, w, xd, yd, lr, alpha, pm, pr, s, 2024, cost, batch=b, row_idx=dev(idx, torch.int32))
        out = o_bbb.bbb_step(rmu, rrho, eps, x[idx], y[idx], spec, lr, alpha, pm, pr)
        rmu, rrho = out["mu"], out["rho"]
        tol = 1e-5 if s == 1 else 1e-4
        close(w, out["w"], rel=tol, what=f"w {s}")
        c = cost.cpu().numpy()
        assert abs(c[0] - out["cost"]) <= 1e-4 * abs(out["cost"]), (c, out["cost"])
        assert abs(c[1] - out["loss"]) <= 1e-4 * abs(out["loss"]), (c, out["loss"])
        close(mu, rmu, rel=tol, what=f"mu {s}")
        close(rho, rrho, rel=tol, what=f"rho {s}")
    vloss, _ = plan.loss_grad(w, dev(xv), dev(yv, torch.int32), want_grad=False)
    rv = o_bbb.validation_loss(out["w"], xv, yv, spec)
    close(vloss, [rv], what="validation loss")
    plan.close()


# ------------------------------------------------------------------ C5
from svgd_checks import lr_t as _lr_t, strict_particle_check as _strict_particle_check  # noqa: E402


@pytest.mark.parametrize("sweep,path", [("gauss_seidel", "resident"), ("gauss_seidel", "fused"), ("gauss_seidel", "rows"), ("jacobi", "gram"),
                                        ("jacobi", "pairwise"), ("jacobi", "rows")])
def test_c5_svgd_64_particles_coupled(eng, monkeypatch, sweep, path):
    """M = 64 on 64 -> 40 -> 24 -> 10 (D = 3 834: five k_svgd_gs workgroups with a ragged last one, the prefetch
    window of 32 rows wraps, all ten 16 x 16 blocks of the Gram kernel) with K_ij ~ 0.2: two steps against the
    oracle, every kernel path; under the Jacobi sweep rows [32, 64) as a shard == the same rows of the whole."""
    monkeypatch.setenv("PYZ_SVGD_GS_FUSED", "1" if path in ("fused", "resident") else "0")
    monkeypatch.setenv("PYZ_SVGD_GS_RESIDENT", "1" if path == "resident" else "0")
    monkeypatch.setenv("PYZ_SVGD_GRAM", "1" if path == "gram" else "0")
    monkeypatch.setenv("PYZ_SVGD_TILES", "0" if (sweep == "jacobi" and path == "rows") else "1")
    spec, n, M, lr = WIDE3, 130, 64, 1e-3
    rng = np.random.default_rng(61)
    x = rng.normal(size=(n, 64)).astype(np.float32)
    y = rng.integers(0, 10, size=n).astype(np.int32)
    D = spec.n_params
    parts = (rng.normal(size=(M, D)) * 0.015).astype(np.float32)     # |x_i - x_j|^2 ~ 2 D 0.015^2 = 1.7
    st = o_svgd.SVGDState(parts)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=M)
    p, am, av = dev(parts), torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    loss, xd, yd = torch.zeros(1, device="cuda"), dev(x), dev(y, torch.int32)
    phis, lr_ts = [], []
    for t in (1, 2):
        if sweep == "jacobi" and t == 2:     # rows [32, 64) of step 2 also as a shard against the gathered matrix
            shard, sm_, sv_ = p[32:].clone(), am[32:].clone(), av[32:].clone()
            plan.svgd_step(shard, p.clone(), 32, sm_, sv_, xd, yd, lr, 1.0, t, loss, sweep="jacobi")
        snap = p.clone() if sweep == "jacobi" else p
        plan.svgd_step(p, snap, 0, am, av, xd, yd, lr, 1.0, t, loss, sweep=sweep)
        out = o_svgd.svgd_step(st, x, y, spec, lr, 1.0, sweep=sweep)
        close(loss, [out["loss"]], what=f"loss {t}")
        phis.append(out["phi"])
        lr_ts.append(_lr_t(lr, t))
        if t == 1:     # m = 0.1 phi, v = 0.001 phi^2: phi itself, element-wise
            close(am, st.m, rel=1e-5, what="adam m after one step")
            close(av, st.v, rel=2e-5, what="adam v after one step")
    close(am, st.m, rel=2e-4, what="adam m")
    close(av, st.v, rel=2e-4, what="adam v")
    _strict_particle_check(p, st, phis, lr_ts, f"{sweep}/{path}")
    if sweep == "jacobi":
        assert torch.equal(shard, p[32:]) and torch.equal(sm_, am[32:]) and torch.equal(sv_, av[32:])
    plan.close()


@pytest.mark.parametrize("sweep", ["gauss_seidel", "gauss_seidel_per_launch", "jacobi"])
def test_c5_svgd_full_shape(eng, sweep, monkeypatch):
    """One pyz_svgd_step at the real C5 shape: 64 particles of 784 -> 200 -> 10 (D = 159 010), batch 1024.
    The particles sit 1e-3 around a Glorot point (|x_i - x_j|^2 ~ 0.3, K_ij ~ 0.7) so that the kernel matrix and
    the repulsion matter at this size; with the reference's N(0, 1) start K underflows to exactly I (second
    case below: phi_i = g_i / M)."""
    monkeypatch.setenv("PYZ_SVGD_GS_RESIDENT", "0" if sweep == "gauss_seidel_per_launch" else "1")
    sweep = sweep.replace("_per_launch", "")
    spec, M, B, lr = MNIST, 64, 1024, 0.01
    D = spec.n_params
    x, y = synth.mnist_like(2048)
    rng = np.random.default_rng(71)
    idx = rng.permutation(2048)[:B].astype(np.int32)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=B, max_particles=M)
    xd, yd, idxd = dev(x), dev(y, torch.int32), dev(idx, torch.int32)
    for case in ("coupled", "prior_start"):
        if case == "coupled":
            parts = (synth.glorot_uniform(spec.dims)[None, :] + 1e-3 * rng.normal(size=(M, D))).astype(np.float32)
        else:
            parts = rng.normal(size=(M, D)).astype(np.float32)
        st = o_svgd.SVGDState(parts)
        p, am, av = dev(parts), torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
        loss = torch.zeros(1, device="cuda")
        snap = p.clone() if sweep == "jacobi" else p
        plan.svgd_step(p, snap, 0, am, av, xd, yd, lr, 1.0, 1, loss, sweep=sweep, batch=B, row_idx=idxd)
        out = o_svgd.svgd_step(st, x[idx], y[idx], spec, lr, 1.0, sweep=sweep)
        close(loss, [out["loss"]], what=f"{case} loss")
        close(am, st.m, rel=1e-4, what=f"{case} adam m")
        close(av, st.v, rel=2e-4, what=f"{case} adam v")
        _strict_particle_check(p, st, [out["phi"]], [_lr_t(lr, 1)], f"{case}/{sweep}")
    plan.close()


# ------------------------------------------------------------------ V5: median-heuristic bandwidth
@pytest.mark.parametrize("M,gram", [(8, 1), (8, 0), (64, 1), (64, 0)])
def test_svgd_median_heuristic_kernel(eng, monkeypatch, M, gram):
    """SVGD.baseline__kernel (SVGD.py:165-181; dead code in the reference, opt-in here): bandwidth from the median
    of all M^2 squared distances, found on the device (k_svgd_kmat distances -> k_svgd_median -> kernel rows).
    Two Jacobi steps against oracle/svgd.py:median_kernel; the second half of the rows as a shard (every rank
    derives the same bandwidth from the gathered matrix) == the same rows of the whole."""
    monkeypatch.setenv("PYZ_SVGD_GRAM", str(gram))
    spec, n, lr = WIDE3, 130, 1e-3
    rng = np.random.default_rng(81 + M)
    x = rng.normal(size=(n, 64)).astype(np.float32)
    y = rng.integers(0, 10, size=n).astype(np.int32)
    D = spec.n_params
    parts = (rng.normal(size=(M, D)) * 0.3).astype(np.float32)       # far apart for gamma = 1: the heuristic rescales
    st = o_svgd.SVGDState(parts)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=M)
    p, am, av = dev(parts), torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    loss, xd, yd = torch.zeros(1, device="cuda"), dev(x), dev(y, torch.int32)
    phis, lr_ts, h = [], [], M // 2
    for t in (1, 2):
        if t == 2:
            shard, sm_, sv_ = p[h:].clone(), am[h:].clone(), av[h:].clone()
            plan.svgd_step(shard, p.clone(), h, sm_, sv_, xd, yd, lr, "median", t, loss, sweep="jacobi")
        plan.svgd_step(p, p.clone(), 0, am, av, xd, yd, lr, None, t, loss, sweep="jacobi")
        out = o_svgd.svgd_step(st, x, y, spec, lr, "median", sweep="jacobi")
        close(loss, [out["loss"]], what=f"loss {t}")
        phis.append(out["phi"])
        lr_ts.append(_lr_t(lr, t))
        if t == 1:
            close(am, st.m, rel=1e-5, what="adam m after one step")
    close(am, st.m, rel=2e-4, what="adam m")
    close(av, st.v, rel=2e-4, what="adam v")
    _strict_particle_check(p, st, phis, lr_ts, f"median M={M}")
    assert torch.equal(shard, p[h:]) and torch.equal(sm_, am[h:])
    # with gamma = 1 these particles do not interact at all (K = I): the heuristic is what couples them
    K, _, _ = o_svgd.median_kernel(parts.astype(np.float64))
    # (at the median distance the heuristic gives K = exp(-log(M + 1)) = 1 / (M + 1))
    assert np.exp(-1.0 * ((parts[0] - parts[1]).astype(np.float64) ** 2).sum()) < 1e-100 and K[0, 1] > 0.3 / (M + 1)
    from bayesian_inference_for_nn_amd._lib import PyzError
    with pytest.raises(PyzError):      # defined on a snapshot only
        plan.svgd_step(p, p, 0, am, av, xd, yd, lr, "median", 3, loss, sweep="gauss_seidel")
    plan.close()

