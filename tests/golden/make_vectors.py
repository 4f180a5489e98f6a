#!/usr/bin/env python3
"""Writes tests/golden/oracle_vectors.npz: seeded inputs and float64 outputs of the CPU oracle
for one step of every method on small shapes.  The reference cannot be executed here
(TensorFlow absent), so these are ORACLE outputs: they freeze the oracle and give the GPU
tests a fixed target.  Run from the repo root:  python tests/golden/make_vectors.py"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bbb, hmc, mlp, philox, predict, sgd, sgld, svgd, swag  # noqa: E402

DIMS, ACTS = (6, 9, 4), ("relu", "softmax")


def main():
    spec = mlp.MLPSpec(DIMS, ACTS, "scce")
    rng = np.random.default_rng(20260101)
    n = 24
    x = rng.normal(size=(n, DIMS[0])).astype(np.float32)
    y = rng.integers(0, DIMS[-1], size=n).astype(np.int32)
    theta = (rng.normal(size=spec.n_params) * 0.4).astype(np.float32)
    D = spec.n_params
    out = dict(dims=np.array(DIMS), x=x, y=y, theta=theta)
    loss, grad, probs = mlp.loss_and_grad(theta, x, y, spec)
    out.update(loss=loss, grad=grad, probs=probs, labels=probs.argmax(1))
    # SGD
    st = sgd.SGDState(theta)
    sgd.sgd_step(st, x, y, spec, 0.05)
    out["sgd_theta"] = st.theta
    # SGLD, 3 steps, library noise stream (seed 11)
    st = sgld.SGLDState(theta)
    lr = sgld.lr_schedule(10, 0.01, 0.003, 0.99)
    for s in range(3):
        sgld.sgld_step(st, x, y, spec, lr(s), philox.normal(11, 0, s, D))
    out.update(sgld_theta=st.theta, sgld_mean=st.mean, sgld_sq_mean=st.sq_mean, sgld_lr=np.array([lr(s) for s in range(3)]))
    # BBB, 1 step
    rho = (rng.normal(size=D) * 0.2 - 1.0).astype(np.float32)
    eps = philox.normal(12, 1, 1, D)
    r = bbb.bbb_step(theta, rho, eps, x, y, spec, 0.01, 0.3, 0.0, 1.0)
    out.update(bbb_rho0=rho, bbb_mu=r["mu"], bbb_rho=r["rho"], bbb_w=r["w"], bbb_cost=r["cost"], bbb_loss=r["loss"])
    # HMC, 1 proposal
    z = rng.normal(size=D).astype(np.float32)
    r = hmc.hmc_step(theta, z, x, y, spec, 0.0, 1.0, 6, 0.002, 0.5, u=0.5)
    out.update(hmc_z=z, hmc_q=r["q_proposed"], hmc_U0=r["U0"], hmc_K0=r["K0"], hmc_U1=r["U1"], hmc_K1=r["K1"],
               hmc_log_ratio=r["log_ratio"])
    # SVGD, 2 steps, 3 particles
    parts = (rng.normal(size=(3, D)) * 0.2).astype(np.float32)
    st = svgd.SVGDState(parts)
    for _ in range(2):
        res = svgd.svgd_step(st, x, y, spec, 0.05)
    out.update(svgd_p0=parts, svgd_p=st.particles, svgd_loss=res["loss"])
    # predict
    W = (rng.normal(size=(4, D)) * 0.4).astype(np.float32)
    s, m = predict.predict(W, x, spec)
    out.update(pred_W=W, pred_samples=s, pred_mean=m)
    # SWAG, 5 steps, k = 2, frequency 2 (moments at steps 0, 2, 4; the third update replaces column 1)
    st = swag.SWAGState(theta, 2)
    for _ in range(5):
        swag.swag_step(st, x, y, spec, 0.05, 2)
    out.update(swag_theta=st.theta, swag_mean=st.mean, swag_sq_mean=st.sq_mean, swag_dev=st.dev)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
