"""Parity of the HIP path (through the C-ABI) against the CPU oracle on identical
seeded inputs.  Tolerances: float32 kernels vs the float64 oracle, 1e-4 relative
to the largest reference magnitude (BASELINE.json north_star); integer labels
bit-exact."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import bbb as o_bbb
from oracle import hmc as o_hmc
from oracle import mlp as o_mlp
from oracle import philox as o_philox
from oracle import predict as o_predict
from oracle import sgd as o_sgd
from oracle import sgld as o_sgld
from oracle import svgd as o_svgd
from oracle import swag as o_swag


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


SPECS = {
    "tiny_cls": (o_mlp.MLPSpec((5, 7, 3), ("relu", "softmax"), "scce"), 11),
    "moons": (o_mlp.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce"), 200),
    "linreg": (o_mlp.MLPSpec((1, 1), ("linear",), "mse"), 64),
    "reg3": (o_mlp.MLPSpec((4, 6, 6, 2), ("tanh", "sigmoid", "linear"), "mse"), 37),
    "mnist_small_batch": (o_mlp.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce"), 96),
    "wide3": (o_mlp.MLPSpec((64, 40, 24, 10), ("relu", "relu", "softmax"), "scce"), 130),
    # last layer wider than one 32-column tile: the unfused path (separate loss / gradient / update kernels)
    "many_classes": (o_mlp.MLPSpec((12, 20, 40), ("tanh", "softmax"), "scce"), 70),
    "wide_regression": (o_mlp.MLPSpec((9, 16, 48), ("relu", "linear"), "mse"), 33),
    "single_row": (o_mlp.MLPSpec((8, 8, 3), ("sigmoid", "softmax"), "scce"), 1),
    "small_reg2": (o_mlp.MLPSpec((3, 5, 2), ("tanh", "linear"), "mse"), 45),      # 2 layers + MSE: fused HMC kernel
    "hmc_regression": (o_mlp.MLPSpec((1, 1, 1), ("linear", "linear"), "mse"), 60),  # HMC_regression.py:36-39
    # head kernels: one wave per row (k_head_rows) with MSE / several units per lane, and the MFMA head
    # (k_head) that serves last layers too large for lane-resident operands
    "reg_rows": (o_mlp.MLPSpec((6, 8, 2), ("tanh", "linear"), "mse"), 29),
    "rows_ut4": (o_mlp.MLPSpec((10, 100, 5), ("sigmoid", "softmax"), "scce"), 41),
    "mfma_head": (o_mlp.MLPSpec((8, 300, 32), ("relu", "softmax"), "scce"), 50),
    # sliced HMC (k_hmc_multi): 7 and 3 row slices per chain
    "moons_700": (o_mlp.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce"), 700),
    "hmc_multi_mse": (o_mlp.MLPSpec((3, 5, 2), ("tanh", "linear"), "mse"), 301),
}


def make(spec, n, seed=0, scale=0.3):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, spec.dims[0])).astype(np.float32)
    if spec.loss == "scce":
        y = rng.integers(0, spec.dims[-1], size=n).astype(np.int32)
    else:
        y = rng.normal(size=(n, spec.dims[-1])).astype(np.float32)
    theta = (rng.normal(size=spec.n_params) * scale).astype(np.float32)
    return x, y, theta


def espec(eng, spec):
    return eng.MLPSpec(spec.dims, spec.acts, spec.loss)


def ydev(spec, y):
    return dev(y, torch.int32 if spec.loss == "scce" else torch.float32)


# ------------------------------------------------------------------ RNG
def test_device_philox_matches_oracle(eng):
    from bayesian_inference_for_nn_amd import _lib
    for n, seed, stream, step in [(1003, 2024, 0, 0), (4096, 7, 3, 12345), (5, (1 << 40) + 9, 1, 2)]:
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        eng.fill_normal(out, seed, stream, step, 0.0, 1.0)
        ref = o_philox.normal(seed, stream, step, n)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=4e-6)
    out = torch.empty(200_000, dtype=torch.float32, device="cuda")
    eng.fill_normal(out, 1, _lib.STREAM_INIT, 0, 0.5, 2.0)
    z = out.cpu().numpy()
    assert abs(z.mean() - 0.5) < 0.02 and abs(z.std() - 2.0) < 0.02


# ------------------------------------------------------------------ forward / gradient
@pytest.mark.parametrize("name", list(SPECS))
def test_loss_grad_and_forward(eng, name):
    spec, n = SPECS[name]
    x, y, theta = make(spec, n, seed=sum(map(ord, name)))
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n + 5, max_particles=1)
    loss, grad = plan.loss_grad(dev(theta), dev(x), ydev(spec, y))
    rl, rg, rout = o_mlp.loss_and_grad(theta, x, y, spec)
    close(loss, [rl], what="loss")
    close(grad[0], rg, what="grad")
    out = plan.forward(dev(theta), dev(x))
    close(out[0], rout, what="forward")
    if spec.loss == "scce":   # integer class labels bit-exact
        assert np.array_equal(out[0].argmax(1).cpu().numpy(), rout.argmax(1))
    loss_only, none = plan.loss_grad(dev(theta), dev(x), ydev(spec, y), want_grad=False)
    assert none is None
    close(loss_only, [rl], what="loss only")
    plan.close()


def test_particles_and_row_gather(eng):
    spec, n = SPECS["wide3"]
    x, y, _ = make(spec, 400, seed=3)
    rng = np.random.default_rng(5)
    thetas = (rng.normal(size=(3, spec.n_params)) * 0.3).astype(np.float32)
    idx = rng.permutation(400)[:n].astype(np.int32)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=256, max_particles=4)
    loss, grad = plan.loss_grad(dev(thetas), dev(x), ydev(spec, y), batch=n, row_idx=dev(idx, torch.int32))
    for p in range(3):
        rl, rg, _ = o_mlp.loss_and_grad(thetas[p], x[idx], y[idx], spec)
        close(loss[p:p + 1], [rl], what=f"loss[{p}]")
        close(grad[p], rg, what=f"grad[{p}]")
    plan.close()


@pytest.mark.parametrize("lds", [1, 0])
def test_wide_layers_many_particles(eng, monkeypatch, lds):
    """64 particles on 128 -> 224 -> 10 with a gathered, ragged batch (250 of 256 rows): the launch fills the
    chip with one tile per wave, so the forward pass of the wide layer runs as the LDS-tiled kernel
    (k_dense_fwd_lds, 128 x 224 per workgroup) -- or, with the switch off, as the one-wave-per-tile kernel.
    Losses and gradients of every particle against the oracle."""
    monkeypatch.setenv("PYZ_FWD_LDS", str(lds))
    monkeypatch.setenv("PYZ_FWD_LDS_MINWG", "64")       # (the default asks for 256 workgroups of 128 rows; here there are 128)
    spec = o_mlp.MLPSpec((128, 224, 10), ("relu", "softmax"), "scce")
    rng = np.random.default_rng(77)
    x = rng.normal(size=(400, 128)).astype(np.float32)
    y = rng.integers(0, 10, size=400).astype(np.int32)
    P, b = 64, 250
    thetas = (rng.normal(size=(P, spec.n_params)) * 0.1).astype(np.float32)
    idx = rng.permutation(400)[:b].astype(np.int32)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=256, max_particles=P)
    loss, grad = plan.loss_grad(dev(thetas), dev(x), dev(y, torch.int32), batch=b, row_idx=dev(idx, torch.int32))
    for p in (0, 1, 17, 63):
        rl, rg, _ = o_mlp.loss_and_grad(thetas[p], x[idx], y[idx], spec)
        close(loss[p:p + 1], [rl], what=f"loss[{p}]")
        close(grad[p], rg, what=f"grad[{p}]")
    plan.close()


def test_many_rows_take_four_rows_per_head_wave(eng):
    """64 particles x 1 003 gathered rows (> 32 768 rows in the launch): the head kernel gives every wave FOUR
    batch rows and loads its share of the last layer once for them (k_head_rows<.., 4>); the ragged end (1 003 is not
    a multiple of four) and the loss partials past the batch are what this covers.  Losses and gradients against
    the oracle."""
    spec = o_mlp.MLPSpec((48, 72, 10), ("tanh", "softmax"), "scce")
    rng = np.random.default_rng(78)
    x = rng.normal(size=(1200, 48)).astype(np.float32)
    y = rng.integers(0, 10, size=1200).astype(np.int32)
    P, b = 64, 1003
    thetas = (rng.normal(size=(P, spec.n_params)) * 0.2).astype(np.float32)
    idx = rng.permutation(1200)[:b].astype(np.int32)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=1024, max_particles=P)
    with eng.KernelProbe(16) as kp:
        loss, grad = plan.loss_grad(dev(thetas), dev(x), dev(y, torch.int32), batch=b, row_idx=dev(idx, torch.int32))
    assert any(name.startswith("k_head_rows") and name.rstrip(">").endswith(", 4") for name, _ in kp.launches), kp.launches
    for p in (0, 5, 33, 63):
        rl, rg, _ = o_mlp.loss_and_grad(thetas[p], x[idx], y[idx], spec)
        close(loss[p:p + 1], [rl], what=f"loss[{p}]")
        close(grad[p], rg, what=f"grad[{p}]")
    plan.close()


def test_full_size_mnist_gradient(eng):
    """BASELINE config 2 shapes: 784 -> 200 -> 10, batch 1024 (and the ragged 896)."""
    spec = o_mlp.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")
    rng = np.random.default_rng(1234)
    x = rng.random((2048, 784), dtype=np.float32)
    y = rng.integers(0, 10, size=2048).astype(np.int32)
    theta = o_mlp.glorot_uniform(spec, np.random.default_rng(99))
    plan = eng.MLPPlan(espec(eng, spec), max_batch=1024)
    for b in (1024, 896):
        idx = rng.permutation(2048)[:b].astype(np.int32)
        loss, grad = plan.loss_grad(dev(theta), dev(x), dev(y, torch.int32), batch=b, row_idx=dev(idx, torch.int32))
        rl, rg, _ = o_mlp.loss_and_grad(theta, x[idx], y[idx], spec)
        close(loss, [rl], what="loss")
        close(grad[0], rg, what="grad")
    plan.close()


def test_shape_errors_are_reported_not_faulted(eng):
    from bayesian_inference_for_nn_amd._lib import PyzError
    spec, n = SPECS["tiny_cls"]
    x, y, theta = make(spec, 40)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=16)
    with pytest.raises((PyzError, ValueError)):
        plan.loss_grad(dev(theta), dev(x), ydev(spec, y))     # 40 rows > max_batch 16
    bad = eng.MLPSpec((5, 7, 3), ("relu", "relu"), "scce")
    p2 = eng.MLPPlan(bad, max_batch=16)
    with pytest.raises(PyzError):
        p2.loss_grad(dev(theta), dev(x[:8]), ydev(spec, y[:8]))  # SCCE without softmax


# ------------------------------------------------------------------ SGD
def test_sgd_steps_match_oracle(eng):
    spec = o_mlp.MLPSpec((1, 1), ("linear",), "mse")
    rng = np.random.default_rng(7)
    x = (1 + 19 * rng.random((600, 1))).astype(np.float32)
    y = (2 * x + 2).astype(np.float32)
    theta0 = np.array([0.3, -0.1], dtype=np.float32)
    st = o_sgd.SGDState(theta0)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=64)
    th = dev(theta0)
    loss = torch.zeros(1, device="cuda")
    xd, yd = dev(x), dev(y)
    for s in range(25):
        idx = rng.permutation(480)[:64].astype(np.int32)
        plan.sgd_step(th, xd, yd, 1e-3, loss, batch=64, row_idx=dev(idx, torch.int32))
        rl, _ = o_sgd.sgd_step(st, x[idx], y[idx], spec, 1e-3)
        close(loss, [rl], what=f"loss step {s}")
    close(th, st.theta, what="theta")


def test_sgd_step_unfused_path(eng):
    spec, n = SPECS["wide_regression"]
    x, y, theta = make(spec, n, seed=77)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n)
    th, loss = dev(theta), torch.zeros(1, device="cuda")
    st = o_sgd.SGDState(theta)
    for _ in range(3):
        plan.sgd_step(th, dev(x), ydev(spec, y), 0.05, loss)
        rl, _ = o_sgd.sgd_step(st, x, y, spec, 0.05)
        close(loss, [rl], what="loss")
    close(th, st.theta, what="theta")


# ------------------------------------------------------------------ SWAG
@pytest.mark.parametrize("name", ["tiny_cls", "wide3", "wide_regression"])
def test_swag_step_matches_oracle(eng, name):
    """SGD update + gated moments + the deviation rows (append until k, then replace the last)."""
    spec, n = SPECS[name]
    x, y, theta = make(spec, n, seed=21)
    D, k, freq = spec.n_params, 3, 2
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n)
    st = o_swag.SWAGState(theta, k)
    th, mean, sq = dev(theta), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    rows, n_cols, loss = torch.zeros((k, D), device="cuda"), 0, torch.zeros(1, device="cuda")
    for s in range(11):
        upd = s % freq == 0
        col = min(n_cols, k - 1)
        plan.swag_step(th, mean, sq, rows[col] if upd else None, dev(x), ydev(spec, y), 0.05, s, upd, loss)
        n_cols += 1 if (upd and n_cols < k) else 0
        rl = o_swag.swag_step(st, x, y, spec, 0.05, freq)
        close(loss, [rl], what=f"loss {s}")
    assert n_cols == st.dev.shape[0] == k
    close(th, st.theta, what="theta")
    close(mean, st.mean, what="mean")
    close(sq, st.sq_mean, what="sq_mean")
    close(rows, st.dev, what="deviation rows")


# ------------------------------------------------------------------ SGLD
@pytest.mark.parametrize("name", ["tiny_cls", "wide3", "many_classes", "wide_regression"])
def test_sgld_step_injected_and_device_noise(eng, name):
    spec, n = SPECS[name]
    x, y, theta = make(spec, n, seed=11)
    D = spec.n_params
    lr = o_sgld.lr_schedule(100, 0.01, 0.003, 0.99)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n)
    for device_noise in (False, True):
        st = o_sgld.SGLDState(theta)
        th, mean, sq = dev(theta), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        loss = torch.zeros(1, device="cuda")
        for s in range(6):
            z = o_philox.normal(2024, 0, s, D)
            plan.sgld_step(th, mean, sq, dev(x), ydev(spec, y), float(lr(s)), s, 2024, loss,
                           unit_noise=None if device_noise else dev(z))
            rl, _ = o_sgld.sgld_step(st, x, y, spec, lr(s), z)
            close(loss, [rl], what="loss")
        close(th, st.theta, what="theta")
        close(mean, st.mean, what="mean")
        close(sq, st.sq_mean, what="sq_mean")
    plan.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_sgld_run_matches_stepwise_oracle(eng, use_graph):
    spec = o_mlp.MLPSpec((24, 16, 4), ("relu", "softmax"), "scce")
    rng = np.random.default_rng(21)
    # 150 rows, batch 64 -> batches 64, 64, 22 per epoch; with the graph: two 32-step replays + 6 eager steps
    N, B, n_steps = 150, 64, (70 if use_graph else 19)
    x = rng.normal(size=(N, 24)).astype(np.float32)
    y = rng.integers(0, 4, size=N).astype(np.int32)
    theta = (rng.normal(size=spec.n_params) * 0.3).astype(np.float32)
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
    lr_fn = o_sgld.lr_schedule(n_steps, 0.01, 0.003, 0.99)
    lrs = [float(np.float32(lr_fn(s))) for s in range(n_steps)]
    D = spec.n_params
    plan = eng.MLPPlan(espec(eng, spec), max_batch=B)
    th, mean, sq = dev(theta), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    losses = torch.zeros(n_steps, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        plan.sgld_run(th, mean, sq, dev(x), dev(y, torch.int32), dev(idx, torch.int32), bs, lrs, 0, 99, losses,
                      use_graph=use_graph)
    stream.synchronize()
    st = o_sgld.SGLDState(theta)
    ref_losses = []
    for s in range(n_steps):
        rows = idx[s, :bs[s]]
        rl, _ = o_sgld.sgld_step(st, x[rows], y[rows], spec, lrs[s], o_philox.normal(99, 0, s, D))
        ref_losses.append(rl)
    close(losses, ref_losses, what="losses")
    close(th, st.theta, what="theta")
    close(mean, st.mean, what="mean")
    close(sq, st.sq_mean, what="sq_mean")
    plan.close()


# ------------------------------------------------------------------ BBB
@pytest.mark.parametrize("name", ["tiny_cls", "reg3", "wide3", "many_classes"])
def test_bbb_step_matches_oracle(eng, name):
    spec, n = SPECS[name]
    x, y, mu0 = make(spec, n, seed=31)
    D = spec.n_params
    rng = np.random.default_rng(32)
    rho0 = (rng.normal(size=D) * 0.3 - 0.5).astype(np.float32)
    pm, pr = o_bbb.mix_prior(0.1, 0.8, 0.0, 0.5, 0.7)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n)
    for device_eps in (False, True):
        mu, rho, w = dev(mu0), dev(rho0), torch.zeros(D, device="cuda")
        cost = torch.zeros(4, device="cuda")
        rmu, rrho = mu0.astype(np.float64), rho0.astype(np.float64)
        for s in range(1, 5):
            eps = o_philox.normal(5, 1, s, D)
            plan.bbb_step(mu, rho, w, dev(x), ydev(spec, y), 1e-2, 0.3, pm, pr, s, 5, cost,
                          eps=None if device_eps else dev(eps))
            out = o_bbb.bbb_step(rmu, rrho, eps, x, y, spec, 1e-2, 0.3, pm, pr)
            rmu, rrho = out["mu"], out["rho"]
            close(w, out["w"], what="w")
            c = cost.cpu().numpy()
            assert abs(c[0] - out["cost"]) <= 1e-4 * abs(out["cost"]) + 1e-5, (c, out["cost"])
            assert abs(c[1] - out["loss"]) <= 1e-4 * abs(out["loss"]) + 1e-6
        close(mu, rmu, what="mu")
        close(rho, rrho, what="rho")
    plan.close()


def test_list_valued_priors_bbb_and_hmc(eng, monkeypatch):
    """Per-layer GaussianPrior lists (GaussianPrior.py:50-69) reach the kernels as per-element vectors."""
    spec, n = SPECS["tiny_cls"]
    x, y, mu0 = make(spec, n, seed=91)
    D = spec.n_params
    sl = spec.layer_slices()
    pm = np.zeros(D, np.float32); pr = np.zeros(D, np.float32)
    pm[sl[0]], pr[sl[0]] = 0.2, 0.7
    pm[sl[1]], pr[sl[1]] = -0.1, 1.3
    rho0 = (np.random.default_rng(92).normal(size=D) * 0.2 - 0.5).astype(np.float32)
    eps = o_philox.normal(3, 1, 1, D)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=2)
    mu, rho, w, cost = dev(mu0), dev(rho0), torch.zeros(D, device="cuda"), torch.zeros(4, device="cuda")
    plan.bbb_step(mu, rho, w, dev(x), ydev(spec, y), 1e-2, 0.3, 0.0, 1.0, 1, 3, cost, prior_mean_vec=dev(pm), prior_rho_vec=dev(pr))
    out = o_bbb.bbb_step(mu0, rho0, eps, x, y, spec, 1e-2, 0.3, pm, pr)
    close(mu, out["mu"], what="mu"); close(rho, out["rho"], what="rho")
    assert abs(float(cost[0]) - out["cost"]) <= 1e-4 * abs(out["cost"]) + 1e-5
    z = np.random.default_rng(93).normal(size=(1, D)).astype(np.float32)
    q = dev(mu0.reshape(1, -1)); stats = torch.zeros((1, 8), device="cuda")
    plan.hmc_step(q, dev(x), ydev(spec, y), 4, 0.002, 0.5, 0.0, 1.0, [0.0], 0, 1, stats, unit_p=dev(z), burning=True,
                  prior_mean_vec=dev(pm), prior_sigma_vec=dev(pr))
    r = o_hmc.hmc_step(mu0, z[0], x, y, spec, pm, pr, 4, 0.002, 0.5, u=0.0, burning=True)
    close(q[0], r["q_proposed"], what="q")
    s = stats.cpu().numpy()[0]
    assert abs(s[2] - r["U0"]) <= 1e-4 * abs(r["U0"]) and abs(s[5] - r["K1"]) <= 1e-4 * abs(r["K1"])
    plan.close()


# ------------------------------------------------------------------ HMC
@pytest.mark.parametrize("name,L,fused,multi", [("moons", 5, 1, 1), ("moons", 5, 1, 0), ("moons", 5, 0, 0), ("moons_700", 4, 1, 1),
                                                ("moons_700", 0, 1, 1), ("moons_700", 1, 1, 1), ("hmc_multi_mse", 3, 1, 1),
                                                ("linreg", 3, 1, 1), ("tiny_cls", 0, 1, 1), ("tiny_cls", 2, 0, 0), ("reg3", 2, 1, 1),
                                                ("small_reg2", 3, 1, 1), ("small_reg2", 3, 0, 0), ("hmc_regression", 4, 1, 1)])
def test_hmc_step_matches_oracle(eng, name, L, fused, multi, monkeypatch):
    """fused = 1: small 2-layer models run inside workgroups that keep the chain state in LDS -- one per
    chain (pyz_hmc_fused.h), or, with multi = 1 and at least 192 rows, NW row slices per chain and one
    launch per gradient evaluation (pyz_hmc_multi.h); fused = 0 forces the generic multi-launch path."""
    monkeypatch.setenv("PYZ_HMC_FUSED", str(fused))
    monkeypatch.setenv("PYZ_HMC_MULTI", str(multi))
    spec, n = SPECS[name]
    x, y, q0 = make(spec, n, seed=41, scale=0.2)
    D = spec.n_params
    rng = np.random.default_rng(42)
    P = 3
    qs = np.stack([q0 + 0.01 * k for k in range(P)]).astype(np.float32)
    zs = rng.normal(size=(P, D)).astype(np.float32)
    eps_, m = 0.002, 0.5
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=P)
    refs = [o_hmc.hmc_step(qs[c], zs[c], x, y, spec, 0.0, 1.0, L, eps_, m, u=0.5) for c in range(P)]
    # uniforms chosen away from the acceptance threshold so that fp32 rounding cannot flip the decision
    us = []
    for c, r in enumerate(refs):
        ratio = np.exp(min(r["log_ratio"], 50.0))
        us.append(0.5 * ratio if c % 2 == 0 else min(2.0 * ratio + 0.1, 1e30))
    refs = [o_hmc.hmc_step(qs[c], zs[c], x, y, spec, 0.0, 1.0, L, eps_, m, u=us[c]) for c in range(P)]
    q = dev(qs)
    stats = torch.zeros((P, 8), device="cuda")
    plan.hmc_step(q, dev(x), ydev(spec, y), L, eps_, m, 0.0, 1.0, us, 0, 1, stats, unit_p=dev(zs))
    s = stats.cpu().numpy()
    if multi:   # on a side stream the sliced path replays a captured graph: same bits, twice
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(2):
            qg, sg = dev(qs), torch.zeros((P, 8), device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                plan.hmc_step(qg, dev(x), ydev(spec, y), L, eps_, m, 0.0, 1.0, us, 0, 1, sg, unit_p=dev(zs))
            side.synchronize()
            assert torch.equal(qg, q) and torch.equal(sg, stats)
    for c, r in enumerate(refs):
        assert bool(s[c, 0]) == r["accepted"], (c, s[c], r["log_ratio"])
        close(q[c], r["q"], what=f"q[{c}]")
        for k, key in ((2, "U0"), (3, "K0"), (4, "U1"), (5, "K1")):
            assert abs(s[c, k] - r[key]) <= 1e-4 * abs(r[key]) + 1e-4, (key, s[c, k], r[key])
        assert abs(s[c, 6] - r["log_ratio"]) <= 2e-4 * max(abs(r["U0"]), abs(r["K0"]), 1.0)
        assert abs(s[c, 1] - r["loss"]) <= 1e-4 * abs(r["loss"]) + 1e-6
    # negative rho: NaN potential -> rejected unless burning; burning accepts
    q2 = dev(qs)
    plan.hmc_step(q2, dev(x), ydev(spec, y), L, eps_, m, 0.0, -1.0, [0.0] * P, 0, 1, stats, unit_p=dev(zs))
    assert (stats[:, 0] == 0).all() and torch.equal(q2, dev(qs))
    plan.hmc_step(q2, dev(x), ydev(spec, y), L, eps_, m, 0.0, -1.0, [0.0] * P, 0, 1, stats, unit_p=dev(zs), burning=True)
    assert (stats[:, 0] == 1).all()
    for c, r in enumerate(refs):
        close(q2[c], r["q_proposed"], what="burning proposal")
    plan.close()


# ------------------------------------------------------------------ SVGD
@pytest.mark.parametrize("sweep", ["gauss_seidel", "jacobi"])
def test_svgd_step_matches_oracle(eng, sweep):
    spec, n = SPECS["tiny_cls"]
    x, y, _ = make(spec, n, seed=51)
    D = spec.n_params
    rng = np.random.default_rng(52)
    M = 5
    parts = (rng.normal(size=(M, D)) * 0.15).astype(np.float32)   # close together: K far from I
    st = o_svgd.SVGDState(parts)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=M)
    p = dev(parts)
    am, av = torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    loss = torch.zeros(1, device="cuda")
    for t in range(1, 4):
        snapshot = p.clone() if sweep == "jacobi" else p
        plan.svgd_step(p, snapshot, 0, am, av, dev(x), ydev(spec, y), 0.05, 1.0, t, loss, sweep=sweep)
        out = o_svgd.svgd_step(st, x, y, spec, 0.05, 1.0, sweep=sweep)
        close(loss, [out["loss"]], what="loss")
    close(p, st.particles, what="particles", rel=2e-4)
    close(am, st.m, what="adam m", rel=2e-4)
    plan.close()


@pytest.mark.parametrize("fused", [2, 1, 0])
def test_svgd_gauss_seidel_paths(eng, monkeypatch, fused):
    """The reference-order sweep as one launch per particle (k_svgd_gs: the matrix in registers, partial
    distances handed from launch to launch) and as the two per-row kernels: both against the oracle, on a
    model whose parameters span several workgroups (D = 3 834 > 768, ragged last one) with particles close
    enough for the kernel matrix to couple them (K_ij ~ 0.05).
    Adam's first steps move every element by ~lr * sign(phi): where phi is ~0 (dead units) float32 and float64
    may disagree on the sign, so the update direction is checked through Adam's m and v (linear / quadratic in
    phi), and the particles on all but a handful of elements."""
    monkeypatch.setenv("PYZ_SVGD_GS_FUSED", str(min(fused, 1)))
    monkeypatch.setenv("PYZ_SVGD_GS_RESIDENT", "1" if fused == 2 else "0")     # 2: the whole sweep as one resident launch
    spec, n = SPECS["wide3"]
    x, y, _ = make(spec, n, seed=57)
    D = spec.n_params
    M, lr = 7, 1e-3
    parts = (np.random.default_rng(58).normal(size=(M, D)) * 0.02).astype(np.float32)
    st = o_svgd.SVGDState(parts)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=M)
    p = dev(parts)
    am, av = torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    loss = torch.zeros(1, device="cuda")
    from svgd_checks import lr_t, strict_particle_check
    phis, lr_ts = [], []
    for t in range(1, 3):
        plan.svgd_step(p, p, 0, am, av, dev(x), ydev(spec, y), lr, 1.0, t, loss, sweep="gauss_seidel")
        out = o_svgd.svgd_step(st, x, y, spec, lr, 1.0, sweep="gauss_seidel")
        close(loss, [out["loss"]], what="loss")
        phis.append(out["phi"])
        lr_ts.append(lr_t(lr, t))
    close(am, st.m, what="adam m", rel=5e-4)
    close(av, st.v, what="adam v", rel=5e-4)
    # elements with a definite phi agree to 2e-4 of the particle scale; only where phi is numerically zero may Adam's
    # lr * sign(phi) steps differ (bounded by 2 sum lr_t) -- the check of tests/test_gpu_baseline_shapes.py
    strict_particle_check(p, st, phis, lr_ts, f"gauss_seidel fused={fused}")
    plan.close()


def test_svgd_jacobi_shard_equals_whole(eng):
    """Rows [2,5) updated as a shard against the gathered matrix == the same rows of a whole-matrix Jacobi step."""
    spec, n = SPECS["tiny_cls"]
    x, y, _ = make(spec, n, seed=53)
    D = spec.n_params
    parts = (np.random.default_rng(54).normal(size=(5, D)) * 0.15).astype(np.float32)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=5)
    loss = torch.zeros(1, device="cuda")
    whole = dev(parts)
    plan.svgd_step(whole, whole.clone(), 0, torch.zeros((5, D), device="cuda"), torch.zeros((5, D), device="cuda"),
                   dev(x), ydev(spec, y), 0.05, 1.0, 1, loss, sweep="jacobi")
    shard = dev(parts[2:5])
    plan.svgd_step(shard, dev(parts), 2, torch.zeros((3, D), device="cuda"), torch.zeros((3, D), device="cuda"),
                   dev(x), ydev(spec, y), 0.05, 1.0, 1, loss, sweep="jacobi")
    close(shard, whole[2:5].cpu().numpy(), rel=1e-6, what="shard rows")
    plan.close()


@pytest.mark.parametrize("gram", [1, 0])
def test_svgd_jacobi_tiled_sweep(eng, monkeypatch, gram):
    """Local rows in multiples of 4 run the all-rows-at-once kernels (k_svgd_gram_tile on the float64 matrix
    cores, or the pairwise k_svgd_dist_tile; kmat; update_tile): 8 particles against the oracle, and rows
    [4, 8) as a shard == the same rows of the whole-matrix step."""
    monkeypatch.setenv("PYZ_SVGD_GRAM", str(gram))
    spec, n = SPECS["wide3"]
    x, y, _ = make(spec, n, seed=55)
    D = spec.n_params
    M = 8
    parts = (np.random.default_rng(56).normal(size=(M, D)) * 0.05).astype(np.float32)   # close together: K far from I
    st = o_svgd.SVGDState(parts)
    plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=M)
    p = dev(parts)
    am, av = torch.zeros((M, D), device="cuda"), torch.zeros((M, D), device="cuda")
    loss = torch.zeros(1, device="cuda")
    for t in range(1, 4):
        if t == 3:   # the third step also as a shard of rows [4, 8)
            shard, sm_, sv_ = p[4:].clone(), am[4:].clone(), av[4:].clone()
            plan.svgd_step(shard, p.clone(), 4, sm_, sv_, dev(x), ydev(spec, y), 0.05, 1.0, t, loss, sweep="jacobi")
        plan.svgd_step(p, p.clone(), 0, am, av, dev(x), ydev(spec, y), 0.05, 1.0, t, loss, sweep="jacobi")
        out = o_svgd.svgd_step(st, x, y, spec, 0.05, 1.0, sweep="jacobi")
        close(loss, [out["loss"]], what="loss")
    close(p, st.particles, what="particles", rel=2e-4)
    close(am, st.m, what="adam m", rel=2e-4)
    assert torch.equal(shard, p[4:]) and torch.equal(sm_, am[4:])       # same kernels, same order: same bits
    plan.close()


# ------------------------------------------------------------------ predict
def test_predict_matches_oracle(eng):
    for name in ("wide3", "reg3"):
        spec, n = SPECS[name]
        x, _, _ = make(spec, n, seed=61)
        W = (np.random.default_rng(62).normal(size=(7, spec.n_params)) * 0.3).astype(np.float32)
        W[2, 1] = np.nan
        plan = eng.MLPPlan(espec(eng, spec), max_batch=n, max_particles=3)   # forces chunking over samples
        samples, mean = plan.predict(dev(W), dev(x))
        rs, rm = o_predict.predict(W, x, spec)
        close(samples, rs, what="samples")
        close(mean, rm, what="mean")
        if spec.loss == "scce":
            assert np.array_equal(mean.argmax(1).cpu().numpy(), rm.argmax(1))
        plan.close()
