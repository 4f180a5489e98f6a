"""Known-answer tests that pin the oracle (the reference ships no golden vectors,
SURVEY.md 8c): analytic answers + an independent torch-autograd restatement."""

import math

import numpy as np
import pytest
import scipy.stats

from oracle import bbb, hmc, mlp, philox, predict, sgd, sgld, svgd, torch_eager

RNG = np.random.default_rng(0)


def _spec_cls():
    return mlp.MLPSpec((5, 7, 3), ("relu", "softmax"), "scce")


def _spec_reg():
    return mlp.MLPSpec((4, 6, 6, 2), ("tanh", "sigmoid", "linear"), "mse")


def _data(spec, n=11, rng=RNG):
    x = rng.normal(size=(n, spec.dims[0]))
    if spec.loss == "scce":
        y = rng.integers(0, spec.dims[-1], size=n)
    else:
        y = rng.normal(size=(n, spec.dims[-1]))
    theta = rng.normal(size=spec.n_params) * 0.5
    return x, y, theta


# ---------------------------------------------------------------- Philox KATs
def test_philox_random123_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kats = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, exp in kats:
        out = philox.philox4x32(*[[c] for c in ctr], key[0], key[1])
        assert tuple(int(o[0]) for o in out) == exp


def test_philox_normal_moments():
    z = philox.normal(2024, 3, 5, 400_000)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 4).mean() - 3) < 0.05
    assert scipy.stats.kstest(z[:20000], "norm").pvalue > 1e-3
    # distinct steps/streams are distinct streams
    assert not np.allclose(z[:16], philox.normal(2024, 3, 6, 16))
    assert not np.allclose(z[:16], philox.normal(2024, 4, 5, 16))


# ---------------------------------------------------------------- MLP gradients
@pytest.mark.parametrize("spec_fn", [_spec_cls, _spec_reg])
def test_gradient_matches_torch_autograd(spec_fn):
    spec = spec_fn()
    x, y, theta = _data(spec)
    loss, g, _ = mlp.loss_and_grad(theta, x, y, spec)
    tl, tg = torch_eager.flat_grad(spec, theta, x, y)
    assert abs(loss - tl) < 1e-12
    np.testing.assert_allclose(g, tg, rtol=1e-10, atol=1e-13)


def test_gradient_finite_difference():
    spec = _spec_cls()
    x, y, theta = _data(spec)
    _, g, _ = mlp.loss_and_grad(theta, x, y, spec)
    for k in RNG.choice(spec.n_params, 10, replace=False):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += 1e-6
        tm[k] -= 1e-6
        fd = (mlp.loss_and_grad(tp, x, y, spec)[0] - mlp.loss_and_grad(tm, x, y, spec)[0]) / 2e-6
        assert abs(fd - g[k]) < 1e-7


def test_flat_order_kernel_then_bias():
    spec = mlp.MLPSpec((2, 3, 1), ("relu", "linear"), "mse")
    theta = np.arange(spec.n_params, dtype=float)
    (w0, b0), (w1, b1) = mlp.unpack(theta, spec)
    assert w0.shape == (2, 3) and w0[0, 1] == 1 and w0[1, 0] == 3
    assert list(b0) == [6, 7, 8] and w1.shape == (3, 1) and b1[0] == 12
    assert spec.offsets() == [(0, 6), (9, 12)] and spec.n_params == 13


def test_scce_uniform_logits_is_log_c():
    spec = mlp.MLPSpec((3, 4), ("softmax",), "scce")
    loss, _, out = mlp.loss_and_grad(np.zeros(spec.n_params), RNG.normal(size=(5, 3)), np.array([0, 1, 2, 3, 0]), spec)
    assert abs(loss - math.log(4)) < 1e-15 and np.allclose(out, 0.25)


# ---------------------------------------------------------------- SGD
def test_sgd_linear_regression_known_answer():
    """y = 2x + 2 (simple_regression_example.py:11-12): one step equals the closed-form
    least-squares gradient and the iteration converges to (w, b) = (2, 2)."""
    spec = mlp.MLPSpec((1, 1), ("linear",), "mse")
    rng = np.random.default_rng(7)
    x = 1 + 19 * rng.random((600, 1))
    y = 2 * x + 2
    st = sgd.SGDState(np.array([0.3, -0.1]))
    w, b = st.theta
    r = w * x + b - y
    g_closed = np.array([2 * (r * x).mean(), 2 * r.mean()])
    sgd.sgd_step(st, x, y, spec, lr=1e-3)
    np.testing.assert_allclose(st.theta, np.array([0.3, -0.1]) - 1e-3 * g_closed, rtol=1e-12)
    np.testing.assert_allclose(st.mean, st.theta)
    for _ in range(60000):
        sgd.sgd_step(st, x, y, spec, lr=4e-3)
    np.testing.assert_allclose(st.theta, [2.0, 2.0], atol=2e-3)


# ---------------------------------------------------------------- SGLD
def test_sgld_lr_schedule_endpoints():
    lr = sgld.lr_schedule(10000, 0.01, 0.003, 0.99)
    assert abs(lr(0) - 0.01) < 1e-12 and abs(lr(10000) - 0.003) < 1e-12
    assert lr(1) < lr(0) and lr(5000) > lr(10000)


def test_sgld_step_and_moments_match_eager_restatement():
    import torch
    spec = _spec_cls()
    x, y, theta = _data(spec, n=16)
    lr = sgld.lr_schedule(50, 0.01, 0.003, 0.99)
    st = sgld.SGLDState(theta)
    eager = torch_eager.EagerSGLD(spec, theta, lr)
    hist = []
    for n in range(5):
        z = philox.normal(1, 0, n, spec.n_params)
        hist.append(st.theta.copy())
        _, ret = sgld.sgld_step(st, x, y, spec, lr(n), z)
        ret_e = eager.step(torch.as_tensor(x, dtype=torch.float32), torch.as_tensor(y), torch.as_tensor(z, dtype=torch.float32))
        assert abs(ret - float(ret_e)) < 1e-5
    np.testing.assert_allclose(st.theta, eager.model.flat(), rtol=2e-5, atol=2e-6)
    # running moments are the plain averages of the visited thetas
    visited = np.stack(hist[1:] + [st.theta])
    np.testing.assert_allclose(st.mean, visited.mean(0), rtol=1e-12)
    np.testing.assert_allclose(st.sq_mean, (visited ** 2).mean(0), rtol=1e-12)
    off = 0
    for l, sl in enumerate(spec.layer_slices()):
        np.testing.assert_allclose(st.mean[sl], eager.mean[l].numpy().ravel(), rtol=2e-5, atol=2e-6)


def test_sgld_noise_scale_is_lr_squared():
    spec = mlp.MLPSpec((1, 1), ("linear",), "mse")
    st = sgld.SGLDState(np.zeros(2))
    x, y = np.zeros((4, 1)), np.zeros((4, 1))       # zero gradient
    sgld.sgld_step(st, x, y, spec, 0.1, np.array([1.0, -2.0]))
    np.testing.assert_allclose(st.theta, [-0.01, 0.02])


# ---------------------------------------------------------------- HMC
def test_normal_log_prob_matches_scipy_and_nan_for_negative_scale():
    x = RNG.normal(size=20)
    np.testing.assert_allclose(hmc.normal_log_prob(x, 0.3, 1.7), scipy.stats.norm.logpdf(x, 0.3, 1.7), rtol=1e-12)
    assert np.isnan(hmc.normal_log_prob(x, 0.0, -1.0)).all()


def test_hmc_potential_gradient_matches_autograd():
    spec = _spec_cls()
    x, y, q = _data(spec, n=13)
    U, loss, g = hmc.potential_energy(q, x, y, spec, 0.1, 1.3, 13)
    tU, tl, tg = torch_eager.hmc_potential_autograd(spec, q, x, y, 0.1, 1.3, 13)
    assert abs(U - tU) < 1e-10 and abs(loss - tl) < 1e-12
    np.testing.assert_allclose(g, tg, rtol=1e-10, atol=1e-12)
    # negative rho: U is NaN, the gradient is finite and equal to the |rho| one
    Un, _, gn = hmc.potential_energy(q, x, y, spec, 0.1, -1.3, 13)
    assert np.isnan(Un)
    np.testing.assert_allclose(gn, g)


def test_hmc_total_kick_and_energy_error():
    """No data term (constant model) -> U is the Gaussian prior: closed-form leapfrog.
    Checks the (L+1)*eps momentum kick of HMC.py:82-87 and O(eps^2) energy error."""
    spec = mlp.MLPSpec((1, 1), ("linear",), "mse")
    X, y = np.zeros((3, 1)), np.zeros((3, 1))
    q0, z = np.array([0.7, -0.4]), np.array([0.3, 1.1])

    def run(eps, L):
        return hmc.hmc_step(q0, z, X, y, spec, 0.0, 1.0, L, eps, 1.0, u=2.0, n_train=0)

    # with n_train = 0 and q frozen by L = 0: p1 = p0 - eps * q0 (one full kick)
    r = run(0.01, 0)
    np.testing.assert_allclose(r["p_final"], z - 0.01 * q0, rtol=1e-12)
    e1 = abs(run(0.02, 10)["log_ratio"])
    e2 = abs(run(0.01, 20)["log_ratio"])
    assert e2 < e1
    assert not run(0.01, 5)["accepted"]                 # u = 2 > any ratio near 1
    assert hmc.hmc_step(q0, z, X, y, spec, 0.0, 1.0, 5, 0.01, 1.0, u=2.0, burning=True, n_train=0)["accepted"]
    # NaN potential (negative rho): rejected unless burning
    rn = hmc.hmc_step(q0, z, X, y, spec, 0.0, -1.0, 5, 0.01, 1.0, u=0.0, n_train=0)
    assert not rn["accepted"] and np.array_equal(rn["q"], q0)


def test_hmc_chain_bookkeeping():
    ch = hmc.HMCChain(np.zeros(2))
    ch.record(dict(q=np.ones(2), accepted=True))
    ch.record(dict(q=np.ones(2), accepted=False))
    ch.record(dict(q=2 * np.ones(2), accepted=True))
    assert ch.frequency == [1, 2, 1] and len(ch.samples) == 3 and ch.samples[0][0] == 0


# ---------------------------------------------------------------- BBB
def test_bbb_closed_form_gradients_match_autograd():
    spec = _spec_cls()
    x, y, mu = _data(spec, n=9)
    rho = RNG.normal(size=spec.n_params) * 0.3 - 0.5
    eps = RNG.normal(size=spec.n_params)
    alpha, lr = 0.3, 1e-2
    pm, pr = bbb.mix_prior(0.1, 0.8, 0.0, 0.5, 0.7)
    out = bbb.bbb_step(mu, rho, eps, x, y, spec, lr, alpha, pm, pr)
    cost, g_mu, g_rho, g_w = torch_eager.bbb_autograd(spec, mu, rho, eps, pm, pr, x, y, alpha)
    assert abs(out["cost"] - cost) < 1e-9
    np.testing.assert_allclose(out["mu"], mu - lr * (g_mu + g_w), rtol=1e-9, atol=1e-12)
    sd = eps / (1 + np.exp(-rho)) * g_w + g_rho
    np.testing.assert_allclose(out["rho"], rho - lr * sd, rtol=1e-9, atol=1e-12)
    assert abs(bbb.cost_function(out["w"], mu, rho, pm, pr, x, y, spec, alpha) - cost) < 1e-9


def test_bbb_prior_mixing():
    assert bbb.mix_prior(0.0, 1.0) == (0.0, 1.0)
    m, r = bbb.mix_prior(1.0, -2.0, 3.0, 4.0, 0.25)
    assert abs(m - 2.5) < 1e-15 and abs(r + math.sqrt(0.25 + 9.0)) < 1e-15
    with pytest.raises(ZeroDivisionError):
        bbb.mix_prior(0.0, 0.0)


# ---------------------------------------------------------------- SVGD
def test_svgd_row_formula_matches_literal_broadcast_autograd():
    P = RNG.normal(size=(4, 6)) * 0.4
    g = RNG.normal(size=6)
    for i in range(4):
        k, rep = svgd.rbf_row(P, i)
        phi = (k.sum() * g + rep) / 4
        np.testing.assert_allclose(phi, torch_eager.svgd_phi_autograd(P, i, g), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(svgd.literal_svgd_gradients(P, g)[2], (svgd.rbf_row(P, 2)[0].sum() * g + svgd.rbf_row(P, 2)[1]) / 4,
                               rtol=1e-5, atol=1e-7)


def test_adam_first_step_closed_form():
    g = np.array([0.5, -2.0, 1e-3])
    th, m, v = svgd.adam_update(np.zeros(3), g, np.zeros(3), np.zeros(3), 1, 0.01, np.float64)
    expect = -0.01 * g / (np.abs(g) + 1e-7 / math.sqrt(1 - 0.999))
    np.testing.assert_allclose(th, expect, rtol=1e-9)


def test_svgd_single_particle_is_adam_and_far_particles_decouple():
    spec = _spec_cls()
    x, y, theta = _data(spec, n=8)
    st = svgd.SVGDState(theta[None, :])
    svgd.svgd_step(st, x, y, spec, 0.01)
    _, g, _ = mlp.loss_and_grad(theta, x, y, spec)
    th, _, _ = svgd.adam_update(theta, g, 0, 0, 1, 0.01, np.float64)
    np.testing.assert_allclose(st.particles[0], th, rtol=1e-12)
    # far apart (K = I): each particle does Adam on g / M
    P = np.stack([theta, theta + 100.0])
    st2 = svgd.SVGDState(P)
    svgd.svgd_step(st2, x, y, spec, 0.01)
    th0, _, _ = svgd.adam_update(theta, g / 2, 0, 0, 1, 0.01, np.float64)
    np.testing.assert_allclose(st2.particles[0], th0, rtol=1e-12)


def test_svgd_gauss_seidel_differs_from_jacobi_when_coupled():
    spec = mlp.MLPSpec((2, 2), ("softmax",), "scce")
    x, y, _ = _data(spec, n=6)
    P = RNG.normal(size=(3, spec.n_params)) * 0.1
    a, b = svgd.SVGDState(P), svgd.SVGDState(P)
    svgd.svgd_step(a, x, y, spec, 0.05, sweep="gauss_seidel")
    svgd.svgd_step(b, x, y, spec, 0.05, sweep="jacobi")
    np.testing.assert_allclose(a.particles[0], b.particles[0], rtol=1e-13)   # first row sees the same snapshot
    assert np.abs(a.particles[2] - b.particles[2]).max() > 1e-6


def test_median_kernel_shapes_and_symmetry():
    P = RNG.normal(size=(5, 3))
    K, dx, h = svgd.median_kernel(P)
    assert K.shape == (5, 5) and dx.shape == (5, 3) and h > 0
    np.testing.assert_allclose(K, K.T)
    np.testing.assert_allclose(np.diag(K), 1.0)


# ---------------------------------------------------------------- predict
def test_predict_mean_nan_to_zero_and_labels():
    spec = _spec_cls()
    x, _, _ = _data(spec, n=6)
    W = RNG.normal(size=(4, spec.n_params))
    W[1, 0] = np.nan
    outs, mean = predict.predict(W, x, spec)
    assert outs.shape == (4, 6, 3) and not np.isnan(mean).any()
    np.testing.assert_allclose(mean, np.nan_to_num(outs).sum(0) / 4)
    assert predict.sampled_index([1, 3, 4], 1) == 0 and predict.sampled_index([1, 3, 4], 2) == 1
    assert predict.sampled_index([1, 3, 4], 4) == 2


def test_swag_moments_are_running_averages_and_deviation_rule():
    """frequency=1: mean / sq_mean are the plain averages of the iterates; the deviation matrix fills
    to k columns and then only its last column is replaced (SWAG.py:84-89)."""
    from oracle import swag
    spec = mlp.MLPSpec((3, 4, 2), ("relu", "softmax"), "scce")
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=(9, 3)), rng.integers(0, 2, size=9)
    st = swag.SWAGState(rng.normal(size=spec.n_params) * 0.3, k=3)
    thetas, firsts = [], None
    for s in range(7):
        swag.swag_step(st, x, y, spec, 0.1, 1)
        thetas.append(st.theta.copy())
        if s == 2:
            firsts = st.dev[:2].copy()
    np.testing.assert_allclose(st.mean, np.mean(thetas, 0), rtol=1e-12)
    np.testing.assert_allclose(st.sq_mean, np.mean(np.square(thetas), 0), rtol=1e-12)
    assert st.dev.shape == (3, spec.n_params)
    np.testing.assert_array_equal(st.dev[:2], firsts)                    # kept forever
    np.testing.assert_allclose(st.dev[2], st.theta - st.mean, rtol=1e-12)
    mean, diag, D = swag.result_distribution(st, scale=2.0)
    assert D.shape == (spec.n_params, 3) and np.all(diag > -1e-12)
    np.testing.assert_allclose(D, st.dev.T, rtol=1e-12)                  # sqrt(scale/(k-1)) = 1
    z1, z2 = rng.normal(size=spec.n_params), rng.normal(size=3)
    np.testing.assert_allclose(swag.lowrank_sample(mean, diag, D, z1, z2), mean + diag * z1 + D @ z2 * 0.5, rtol=1e-12)
