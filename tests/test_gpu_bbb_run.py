"""pyz_bbb_run: the BBB train loop (BBB.py:128-211 inside Optimizer.py:121-134) as one device-resident, graph-replayed run
-- batches assembled one step ahead, step scalars on the device, the validation forward of BBB.py:203-209 inside the run
(its launches return at once on every tenth step) -- against the same steps as single pyz_bbb_step calls plus
pyz_mlp_loss_grad on the validation plan.  Whole batches: bit for bit (mu, rho, the sampled weights, every cost triple,
every validation loss).  A ragged last batch per epoch changes the launch geometry of the chained run (it launches for
the largest batch), i.e. a float32 summation order: 2e-6 of the largest magnitude there.
And through the drop-in surface: BBB.train(verbose=False) == the per-step loop."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import bbb as o_bbb
from oracle import mlp as o_mlp


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.fixture(scope="module")
def eng(gpu_device):
    from bayesian_inference_for_nn_amd import engine
    return engine


def same(a, b, name, exact):
    a, b = np.asarray(a), np.asarray(b)
    if exact:
        assert np.array_equal(a, b), f"{name}: run and eager steps differ (max {np.abs(a - b).max():.3e})"
    else:
        scale = max(float(np.abs(b).max()), 1e-30)
        assert np.abs(a.astype(np.float64) - b).max() <= 2e-6 * scale, f"{name}: {np.abs(a - b).max():.3e} vs scale {scale:.3e}"


def batches(rng, N, B, n_steps):
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
@pytest.mark.parametrize("n_steps,step0", [(1, 1), (10, 1), (33, 8), (45, 20)])
@pytest.mark.parametrize("with_val", [True, False])
def test_bbb_run_equals_eager_steps(eng, n_steps, step0, ragged, with_val):
    spec = o_mlp.MLPSpec((24, 32, 16, 5), ("relu", "tanh", "softmax"), "scce")
    es = eng.MLPSpec(spec.dims, spec.acts, spec.loss)
    D, B = spec.n_params, 64
    N = 500 if ragged else 512
    rng = np.random.default_rng(100 + n_steps)
    x = rng.normal(size=(N, 24)).astype(np.float32)
    y = rng.integers(0, 5, size=N).astype(np.int32)
    xv = rng.normal(size=(130, 24)).astype(np.float32)
    yv = rng.integers(0, 5, size=130).astype(np.int32)
    idx_h, sizes = batches(rng, N, B, n_steps)
    lr, alpha, pm, pr, seed = 2e-3, 0.05, 0.0, -2.0, 77
    plan, vplan = eng.MLPPlan(es, max_batch=B), eng.MLPPlan(es, max_batch=130)
    xd, yd, xvd, yvd, idx = dev(x), dev(y, torch.int32), dev(xv), dev(yv, torch.int32), dev(idx_h, torch.int32)
    mu0 = (rng.normal(size=D) * 0.2).astype(np.float32)
    rho0 = np.full(D, -2.0, np.float32)

    # (a) single eager steps
    mu, rho, w = dev(mu0), dev(rho0), torch.zeros(D, device="cuda")
    cost = torch.zeros(4, device="cuda")
    costs_e, vals_e = [], {}
    for i in range(n_steps):
        plan.bbb_step(mu, rho, w, xd, yd, lr, alpha, pm, pr, step0 + i, seed, cost, batch=sizes[i], row_idx=idx[i])
        costs_e.append(cost[:3].cpu().numpy().copy())
        if with_val and (step0 + i) % 10:
            vl, _ = vplan.loss_grad(w, xvd, yvd, want_grad=False)
            vals_e[i] = float(vl[0])
    ref = (mu.cpu().numpy(), rho.cpu().numpy(), w.cpu().numpy())

    # (b) one device-resident run (twice: the second call replays the captured graphs)
    for rep in range(2):
        mu, rho, w = dev(mu0), dev(rho0), torch.zeros(D, device="cuda")
        costs = torch.zeros((n_steps + 3, 4), device="cuda")
        vals = torch.full((n_steps + 3,), -1.0, device="cuda")
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            plan.bbb_run(mu, rho, w, xd, yd, idx, sizes, [lr] * n_steps, alpha, pm, pr, step0, seed, costs,
                         val_plan=vplan if with_val else None, val_x=xvd if with_val else None, val_y=yvd if with_val else None,
                         val_losses_out=vals if with_val else None)
        st.synchronize()
        exact = not ragged
        for name, a, b in zip(("mu", "rho", "w"), (mu, rho, w), ref):
            same(a.cpu().numpy(), b, f"{name} (rep {rep})", exact)
        same(costs[:n_steps, :3].cpu().numpy(), np.stack(costs_e), "cost triples", exact)
        v = vals.cpu().numpy()
        for i in range(n_steps):
            if with_val and (step0 + i) % 10:
                same(v[i:i + 1], np.asarray([vals_e[i]], np.float32), f"validation loss of step {step0 + i}", exact)
            else:
                assert v[i] == -1.0, f"step {step0 + i} must not validate"
    kind, n = plan.last_run_path()
    assert kind == "graph" and n == n_steps
    plan.close()
    vplan.close()


def test_bbb_run_first_steps_match_the_oracle(eng):
    """The run against the float64 oracle with the device's own Philox noise (three steps, whole batches)."""
    from oracle import philox as o_philox
    from bayesian_inference_for_nn_amd import _lib
    spec = o_mlp.MLPSpec((12, 20, 4), ("relu", "softmax"), "scce")
    es = eng.MLPSpec(spec.dims, spec.acts, spec.loss)
    D, B, N = spec.n_params, 32, 96
    rng = np.random.default_rng(5)
    x = rng.normal(size=(N, 12)).astype(np.float32)
    y = rng.integers(0, 4, size=N).astype(np.int32)
    idx_h, sizes = batches(rng, N, B, 3)
    mu0, rho0 = (rng.normal(size=D) * 0.2).astype(np.float32), np.full(D, -1.5, np.float32)
    plan = eng.MLPPlan(es, max_batch=B)
    mu, rho, w = dev(mu0), dev(rho0), torch.zeros(D, device="cuda")
    costs = torch.zeros((3, 4), device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.bbb_run(mu, rho, w, dev(x), dev(y, torch.int32), dev(idx_h, torch.int32), sizes, [1e-2] * 3, 0.1, 0.0, 1.0, 4, 21, costs)
    st.synchronize()
    m, r = mu0.astype(np.float64), rho0.astype(np.float64)
    for i in range(3):
        eps = o_philox.normal(21, _lib.STREAM_BBB, 4 + i, D)
        rows = idx_h[i, :sizes[i]]
        out = o_bbb.bbb_step(m, r, eps, x[rows], y[rows], spec, 1e-2, 0.1, 0.0, 1.0)
        m, r = out["mu"], out["rho"]
        np.testing.assert_allclose(float(costs[i, 0]), out["cost"], rtol=1e-4)
    np.testing.assert_allclose(mu.cpu().numpy(), m, rtol=0, atol=1e-5 * np.abs(m).max())
    np.testing.assert_allclose(rho.cpu().numpy(), r, rtol=0, atol=1e-5 * np.abs(r).max())
    plan.close()


def test_bbb_train_quiet_equals_the_step_loop(gpu_device):
    """BBB.train(verbose=False) takes the device-resident run; the same optimizer driven step by step (verbose=True path,
    output swallowed) must end with the same posterior, the same train / validation loss lists."""
    import contextlib
    import io
    from bayesian_inference_for_nn_amd.datasets import Dataset
    from bayesian_inference_for_nn_amd.distributions import GaussianPrior
    from bayesian_inference_for_nn_amd.losses import SparseCategoricalCrossentropy
    from bayesian_inference_for_nn_amd.nn import sequential_json
    from bayesian_inference_for_nn_amd.optimizers import BBB
    from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters
    rng = np.random.default_rng(3)
    x = rng.normal(size=(800, 10)).astype(np.float32)          # 640 training rows: ten whole batches of 64 per epoch
    y = (x[:, 0] + x[:, 1] > 0).astype(np.int64)
    cfg = sequential_json(10, [24, 2], ["relu", "softmax"])
    outs = []
    for quiet in (True, False):
        ds = Dataset((x, y), SparseCategoricalCrossentropy, "Classification", seed=11)
        opt = BBB()
        opt.compile(HyperParameters(lr=5e-3, alpha=0.01, batch_size=64), cfg, ds, verbose=not quiet, prior=GaussianPrior(0.0, -2.0), seed=13)
        if quiet:
            opt.train(27)
            opt.train(8)                                       # a second call continues the chain (step counts, epochs)
        else:
            with contextlib.redirect_stdout(io.StringIO()):
                opt.train(35)
        outs.append((opt._mu.cpu().numpy(), opt._rho.cpu().numpy(), [float(v) for v in opt.train_losses],
                     [float(v) for v in opt.val_losses], opt._step))
    a, b = outs
    assert a[4] == b[4] == 35 and len(a[2]) == len(b[2]) == 32 and len(a[3]) == len(b[3]) == 32
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and a[3] == b[3]
