"""Committed vectors: the oracle must reproduce them (CPU) and the HIP path must hit them (GPU)."""

import os

import numpy as np
import pytest

from oracle import bbb, hmc, mlp, philox, predict, sgd, sgld, svgd, swag

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz"))
SPEC = mlp.MLPSpec(tuple(int(d) for d in G["dims"]), ("relu", "softmax"), "scce")
D = SPEC.n_params


def test_oracle_reproduces_the_committed_vectors():
    x, y, theta = G["x"], G["y"], G["theta"]
    loss, grad, probs = mlp.loss_and_grad(theta, x, y, SPEC)
    np.testing.assert_allclose(loss, G["loss"], rtol=1e-13)
    np.testing.assert_allclose(grad, G["grad"], rtol=1e-11, atol=1e-15)
    assert np.array_equal(probs.argmax(1), G["labels"])
    st = sgld.SGLDState(theta)
    for s in range(3):
        sgld.sgld_step(st, x, y, SPEC, G["sgld_lr"][s], philox.normal(11, 0, s, D))
    np.testing.assert_allclose(st.theta, G["sgld_theta"], rtol=1e-11)
    np.testing.assert_allclose(st.sq_mean, G["sgld_sq_mean"], rtol=1e-11)
    r = bbb.bbb_step(theta, G["bbb_rho0"], philox.normal(12, 1, 1, D), x, y, SPEC, 0.01, 0.3, 0.0, 1.0)
    np.testing.assert_allclose(r["mu"], G["bbb_mu"], rtol=1e-11)
    np.testing.assert_allclose(r["cost"], G["bbb_cost"], rtol=1e-12)
    r = hmc.hmc_step(theta, G["hmc_z"], x, y, SPEC, 0.0, 1.0, 6, 0.002, 0.5, u=0.5)
    np.testing.assert_allclose(r["q_proposed"], G["hmc_q"], rtol=1e-11)
    np.testing.assert_allclose(r["log_ratio"], G["hmc_log_ratio"], rtol=1e-8, atol=1e-10)
    st = svgd.SVGDState(G["svgd_p0"])
    for _ in range(2):
        svgd.svgd_step(st, x, y, SPEC, 0.05)
    np.testing.assert_allclose(st.particles, G["svgd_p"], rtol=1e-11)
    s, m = predict.predict(G["pred_W"], x, SPEC)
    np.testing.assert_allclose(m, G["pred_mean"], rtol=1e-12)
    st = swag.SWAGState(theta, 2)
    for _ in range(5):
        swag.swag_step(st, x, y, SPEC, 0.05, 2)
    np.testing.assert_allclose(st.mean, G["swag_mean"], rtol=1e-11)
    np.testing.assert_allclose(st.dev, G["swag_dev"], rtol=1e-9, atol=1e-14)


@pytest.mark.gpu
def test_hip_path_hits_the_committed_vectors(gpu_device):
    import torch
    from bayesian_inference_for_nn_amd import engine

    def dev(a, dt=torch.float32):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda()

    def close(got, ref, rel=1e-4):
        got = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
        assert np.abs(got - ref).max() <= rel * max(np.abs(ref).max(), 1e-30)

    x, y, theta = dev(G["x"]), dev(G["y"], torch.int32), dev(G["theta"])
    plan = engine.MLPPlan(engine.MLPSpec(SPEC.dims, SPEC.acts, "scce"), max_batch=len(G["x"]), max_particles=4)
    loss, grad = plan.loss_grad(theta, x, y)
    close(loss, np.array([G["loss"]]))
    close(grad[0], G["grad"])
    out = plan.forward(theta, x)
    assert np.array_equal(out[0].argmax(1).cpu().numpy(), G["labels"])       # integer labels bit-exact
    th, mean, sq, l1 = theta.clone(), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(1, device="cuda")
    for s in range(3):
        plan.sgld_step(th, mean, sq, x, y, float(G["sgld_lr"][s]), s, 11, l1)
    close(th, G["sgld_theta"]); close(mean, G["sgld_mean"]); close(sq, G["sgld_sq_mean"])
    th = theta.clone()
    plan.sgd_step(th, x, y, 0.05, l1)
    close(th, G["sgd_theta"])
    mu, rho, w, cost = theta.clone(), dev(G["bbb_rho0"]), torch.zeros(D, device="cuda"), torch.zeros(4, device="cuda")
    plan.bbb_step(mu, rho, w, x, y, 0.01, 0.3, 0.0, 1.0, 1, 12, cost)
    close(mu, G["bbb_mu"]); close(rho, G["bbb_rho"]); close(w, G["bbb_w"])
    assert abs(float(cost[0]) - G["bbb_cost"]) <= 1e-4 * abs(G["bbb_cost"])
    q, stats = theta.clone().reshape(1, -1), torch.zeros((1, 8), device="cuda")
    plan.hmc_step(q, x, y, 6, 0.002, 0.5, 0.0, 1.0, [0.0], 0, 1, stats, unit_p=dev(G["hmc_z"]).reshape(1, -1), burning=True)
    close(q[0], G["hmc_q"])
    s = stats.cpu().numpy()[0]
    for k, key in ((2, "hmc_U0"), (3, "hmc_K0"), (4, "hmc_U1"), (5, "hmc_K1")):
        assert abs(s[k] - G[key]) <= 1e-4 * abs(G[key])
    p, am, av = dev(G["svgd_p0"]), torch.zeros((3, D), device="cuda"), torch.zeros((3, D), device="cuda")
    for t in (1, 2):
        plan.svgd_step(p, p, 0, am, av, x, y, 0.05, 1.0, t, l1)
    close(p, G["svgd_p"], rel=2e-4)
    assert abs(float(l1) - G["svgd_loss"]) <= 1e-4 * abs(G["svgd_loss"])
    samples, mean = plan.predict(dev(G["pred_W"]), x)
    close(samples, G["pred_samples"]); close(mean, G["pred_mean"])
    th, mean, sq, rows, cols = theta.clone(), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), \
        torch.zeros((2, D), device="cuda"), 0
    for s in range(5):
        upd = s % 2 == 0
        plan.swag_step(th, mean, sq, rows[min(cols, 1)] if upd else None, x, y, 0.05, s, upd, l1)
        cols += 1 if (upd and cols < 2) else 0
    close(th, G["swag_theta"]); close(mean, G["swag_mean"]); close(sq, G["swag_sq_mean"]); close(rows, G["swag_dev"])
