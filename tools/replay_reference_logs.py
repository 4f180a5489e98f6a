#!/usr/bin/env python3
"""Replays rows of the reference's held outputs (`logs/*.txt`, copied as numbers into
tests/golden/reference_logs.json) and prints what this repo gets for them.

The logs are the ONLY outputs the reference holds for the hot path.  They are grid searches of unseeded runs
(test accuracy / MSE of one run per row), so they pin behaviour statistically, not bit-wise:

  HMC_classification_FULL.txt   GaussianPrior(0.0, -1.0), 100 sampling steps after the 10 burn-in steps.  Under
                                SURVEY.md Appendix A2 (tfp Normal.log_prob does not validate its scale: log(-1) is
                                NaN, every Metropolis test fails) the posterior is the single state the 10 forced
                                burn-in trajectories reach from q = 0, so accuracy grows with epsilon * L / m
                                (52.5 % at 0.001, m = 2, L = 10 ... 98 % at 0.005, m = 0.5, L = 30).
  SVGD_regression_FULL.txt      y = 2x + 2, 1 -> 1 -> 1 linear, 500 steps: MSE <= 1e-6 for lr = 0.1, ~1e-3..1e-1 for
                                lr = 0.01, > 1 for lr = 0.001: the per-particle Keras legacy Adam (Appendix A3).
  BBB_classification_FULL.txt   moons, 600 steps, prior (0.0, -1.0): alpha = 0 learns (80 - 98.5 %), alpha >= 0.1
                                stays at chance (28 - 59 %): the alpha-weighted KL term of BBB.py:107-124,152-201.

    python tools/replay_reference_logs.py --backend oracle|gpu [--what hmc,svgd,bbb] [--seeds 3]

backend oracle = the float64 NumPy restatement (oracle/, CPU); gpu = the drop-in surface of this package
(Optimizer.compile / train / result / BayesianModel.predict) on the HIP kernels."""

from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_logs.json")


def golden():
    return json.load(open(GOLDEN))


def make_moons(n, noise, rng):
    """sklearn.datasets.make_moons(n_samples=n, noise=noise) (HMC_classification.py:36, BBB_classification.py:39),
    restated on a NumPy generator (the scripts do not seed it)."""
    n_out = n // 2
    n_in = n - n_out
    outer = np.stack([np.cos(np.linspace(0, np.pi, n_out)), np.sin(np.linspace(0, np.pi, n_out))], 1)
    inner = np.stack([1 - np.cos(np.linspace(0, np.pi, n_in)), 1 - np.sin(np.linspace(0, np.pi, n_in)) - 0.5], 1)
    x = np.concatenate([outer, inner])
    y = np.concatenate([np.zeros(n_out, dtype=np.int32), np.ones(n_in, dtype=np.int32)])
    x = x + rng.normal(scale=noise, size=x.shape)
    return x.astype(np.float32), y


def split(x, y, rng):
    """Dataset.py:113-122: shuffle, then 80 / 10 / 10."""
    perm = rng.permutation(len(x))
    x, y = x[perm], y[perm]
    n_tr, n_te = int(0.8 * len(x)), int(0.1 * len(x))
    return (x[:n_tr], y[:n_tr]), (x[n_tr:n_tr + n_te], y[n_tr:n_tr + n_te])


# ------------------------------------------------------------------------------------------------ oracle backend
def oracle_hmc(eps, m, L, seed, prior_sigma=-1.0, n_sampling=100):
    from oracle import hmc as o_hmc, mlp as o_mlp
    rng = np.random.default_rng(seed)
    (xt, yt), (xe, ye) = split(*make_moons(2000, 0.2, rng), rng)
    spec = o_mlp.MLPSpec((2, 50, 2), ("relu", "softmax"), "scce")
    D = spec.n_params
    chain = o_hmc.HMCChain(np.zeros(D))                                   # HMC.py:69-72: q <- prior mean
    for _ in range(10):                                                   # HMC.py:111-116
        r = o_hmc.hmc_step(chain.q, rng.normal(size=D), xt, yt, spec, 0.0, prior_sigma, L, eps, m, u=rng.random(), burning=True)
        chain.record(r, sampling=False)
    accepted = 0
    # a NaN potential rejects every proposal whatever it is: a handful is run to show it, the rest is bookkeeping
    n_run = n_sampling if prior_sigma > 0 else 3
    for _ in range(n_run):
        r = o_hmc.hmc_step(chain.q, rng.normal(size=D), xt, yt, spec, 0.0, prior_sigma, L, eps, m, u=rng.random())
        chain.record(r, sampling=True)
        accepted += int(r["accepted"])
    if prior_sigma <= 0:
        assert accepted == 0 and len(chain.samples) == 1
        chain.frequency[-1] += n_sampling - n_run
    w = np.asarray(chain.frequency, dtype=np.float64)
    probs = sum(wk * o_mlp.predict(s, xe, spec) for wk, s in zip(w, chain.samples)) / w.sum()   # BayesianModel.py:119-128
    return {"accuracy": 100.0 * float((probs.argmax(1) == ye).mean()), "accepted": accepted, "distinct_samples": len(chain.samples)}


def oracle_svgd_regression(lr, batch, M, seed, steps=500):
    from oracle import mlp as o_mlp, svgd as o_svgd
    rng = np.random.default_rng(seed)
    x = rng.uniform(1, 20, size=(600, 1)).astype(np.float32)              # SVGD_regression.py:88-89
    y = (2 * x + 2).astype(np.float32)
    (xt, yt), (xe, ye) = split(x, y, rng)
    spec = o_mlp.MLPSpec((1, 1, 1), ("linear", "linear"), "mse")
    st = o_svgd.SVGDState(rng.normal(size=(M, spec.n_params)), wdtype=np.float32)    # SVGD.py:143-157, prior N(0, 1)
    s = 0
    while s < steps:
        perm = rng.permutation(len(xt))
        for o in range(0, len(xt), batch):
            if s == steps:
                break
            rows = perm[o:o + batch]
            o_svgd.svgd_step(st, xt[rows], yt[rows], spec, lr, 1.0)
            s += 1
    pred = np.mean([o_mlp.predict(p, xe, spec) for p in st.particles], axis=0)       # SVGD_regression.py:80-81
    return {"mse": float(((pred - ye) ** 2).mean())}


def oracle_bbb_classification(lr, alpha, batch, hidden, seed, steps=600):
    from oracle import bbb as o_bbb, mlp as o_mlp
    rng = np.random.default_rng(seed)
    (xt, yt), (xe, ye) = split(*make_moons(2000, 0.2, rng), rng)
    spec = o_mlp.MLPSpec((2, hidden, 2), ("relu", "softmax"), "scce")
    D = spec.n_params
    pm, pr = o_bbb.mix_prior(0.0, -1.0)                                   # BBB.py:258-270
    mu, rho = np.full(D, pm), np.full(D, pr)                              # BBB.py:277-296
    s = 0
    while s < steps:
        perm = rng.permutation(len(xt))
        for o in range(0, len(xt), batch):
            if s == steps:
                break
            rows = perm[o:o + batch]
            out = o_bbb.bbb_step(mu, rho, rng.normal(size=D), xt[rows], yt[rows], spec, lr, alpha, pm, pr)
            mu, rho = out["mu"], out["rho"]
            s += 1
    loc, scale = o_bbb.result_distribution(mu, rho)
    probs = np.mean([o_mlp.predict(loc + scale * rng.normal(size=D), xe, spec) for _ in range(100)], axis=0)
    probs = np.nan_to_num(probs)
    return {"accuracy": 100.0 * float((probs.argmax(1) == ye).mean())}


def oracle_svgd_classification(lr, batch, M, seed, steps=1000):
    from oracle import mlp as o_mlp, svgd as o_svgd
    rng = np.random.default_rng(seed)
    (xt, yt), (xe, ye) = split(*make_moons(2000, 0.2, rng), rng)          # SVGD_classification.py:130,145-149
    spec = o_mlp.MLPSpec((2, 64, 2), ("relu", "softmax"), "scce")         # :152-155
    st = o_svgd.SVGDState(rng.normal(size=(M, spec.n_params)), wdtype=np.float32)    # prior N(0, 1): :158
    s = 0
    while s < steps:
        perm = rng.permutation(len(xt))
        for o in range(0, len(xt), batch):
            if s == steps:
                break
            rows = perm[o:o + batch]
            o_svgd.svgd_step(st, xt[rows], yt[rows], spec, lr, 1.0)
            s += 1
    probs = np.mean([o_mlp.predict(p, xe, spec) for p in st.particles], axis=0)      # :86-93 ensemble mean of the softmax outputs
    return {"accuracy": float((probs.argmax(1) == ye).mean())}


def oracle_bbb_regression(lr, alpha, batch, hidden, seed, steps=2000):
    from oracle import bbb as o_bbb, mlp as o_mlp
    rng = np.random.default_rng(seed)
    x = rng.uniform(1, 20, size=(600, 1)).astype(np.float32)              # BBB_regression.py:37-38
    y = (2 * x + 2).astype(np.float32)
    (xt, yt), (xe, ye) = split(x, y, rng)
    spec = o_mlp.MLPSpec((1, hidden, 1), ("linear", "linear"), "mse")     # :48-51
    D = spec.n_params
    pm, pr = o_bbb.mix_prior(0.0, 1.0)                                    # :54; BBB.py:258-270
    mu, rho = np.full(D, pm), np.full(D, pr)
    s = 0
    while s < steps:
        perm = rng.permutation(len(xt))
        for o in range(0, len(xt), batch):
            if s == steps:
                break
            rows = perm[o:o + batch]
            out = o_bbb.bbb_step(mu, rho, rng.normal(size=D), xt[rows], yt[rows], spec, lr, alpha, pm, pr)
            mu, rho = out["mu"], out["rho"]
            s += 1
    loc, scale = o_bbb.result_distribution(mu, rho)
    pred = np.mean([o_mlp.predict(loc + scale * rng.normal(size=D), xe, spec) for _ in range(10)], axis=0)   # :73 n_boundaries=10
    pred = np.nan_to_num(pred)
    return {"mse": float(((pred - ye.reshape(pred.shape)) ** 2).mean())}


# ------------------------------------------------------------------------------------------------ gpu backend
def _surface():
    from bayesian_inference_for_nn_amd.datasets import Dataset
    from bayesian_inference_for_nn_amd.distributions import GaussianPrior
    from bayesian_inference_for_nn_amd.losses import MeanSquaredError, SparseCategoricalCrossentropy
    from bayesian_inference_for_nn_amd.nn import sequential_json
    from bayesian_inference_for_nn_amd.optimizers import BBB, HMC, SVGD
    from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters
    return locals()


def _test_split(ds):
    xe, ye = next(iter(ds.test_data.batch(ds.test_size)))
    return xe.numpy(), ye.numpy()


def gpu_hmc(eps, m, L, seed, prior_sigma=-1.0, n_sampling=100):
    import random
    s = _surface()
    random.seed(seed)                                                     # HMC.py:91 draws from Python's generator
    rng = np.random.default_rng(seed)
    ds = s["Dataset"](make_moons(2000, 0.2, rng), s["SparseCategoricalCrossentropy"], "Classification", seed=seed)
    opt = s["HMC"]()
    opt.compile(s["HyperParameters"](epsilon=eps, m=m, L=L), s["sequential_json"](2, [50, 2], ["relu", "softmax"]), ds,
                verbose=False, prior=s["GaussianPrior"](0.0, prior_sigma), seed=seed)
    opt.train(n_sampling)
    bm = opt.result()
    xe, ye = _test_split(ds)
    _, mean = bm.predict(xe, nb_samples=100)
    return {"accuracy": 100.0 * float((np.asarray(mean).argmax(1) == ye.reshape(-1)).mean()), "accepted": int(opt._accepted_runs),
            "distinct_samples": len(opt._samples)}


def gpu_svgd_regression(lr, batch, M, seed, steps=500):
    s = _surface()
    rng = np.random.default_rng(seed)
    x = rng.uniform(1, 20, size=(600, 1)).astype(np.float32)
    ds = s["Dataset"]((x, (2 * x + 2).astype(np.float32)), s["MeanSquaredError"], "Regression", seed=seed)
    opt = s["SVGD"]()
    opt.compile(s["HyperParameters"](lr=lr, batch_size=batch, M=M), s["sequential_json"](1, [1, 1], ["linear", "linear"]), ds,
                verbose=False, prior=s["GaussianPrior"](0, 1), seed=seed)
    opt.train(steps)
    models, _, _ = opt.result()
    xe, ye = _test_split(ds)
    pred = np.mean([np.asarray(mm.predict(xe)) for mm in models], axis=0)
    return {"mse": float(((pred - ye.reshape(pred.shape)) ** 2).mean())}


def gpu_bbb_classification(lr, alpha, batch, hidden, seed, steps=600):
    s = _surface()
    rng = np.random.default_rng(seed)
    ds = s["Dataset"](make_moons(2000, 0.2, rng), s["SparseCategoricalCrossentropy"], "Classification", seed=seed)
    opt = s["BBB"]()
    opt.compile(s["HyperParameters"](lr=lr, alpha=alpha, batch_size=batch), s["sequential_json"](2, [hidden, 2], ["relu", "softmax"]),
                ds, verbose=False, prior=s["GaussianPrior"](0.0, -1.0), seed=seed)
    opt.train(steps)
    bm, _, _ = opt.result()
    xe, ye = _test_split(ds)
    _, mean = bm.predict(xe, nb_samples=100)
    return {"accuracy": 100.0 * float((np.asarray(mean).argmax(1) == ye.reshape(-1)).mean())}


def gpu_svgd_classification(lr, batch, M, seed, steps=1000):
    s = _surface()
    rng = np.random.default_rng(seed)
    ds = s["Dataset"](make_moons(2000, 0.2, rng), s["SparseCategoricalCrossentropy"], "Classification", seed=seed)
    opt = s["SVGD"]()
    opt.compile(s["HyperParameters"](lr=lr, batch_size=batch, M=M), s["sequential_json"](2, [64, 2], ["relu", "softmax"]), ds,
                verbose=False, prior=s["GaussianPrior"](0, 1), seed=seed)
    opt.train(steps)
    models, _, _ = opt.result()
    xe, ye = _test_split(ds)
    probs = np.mean([np.asarray(mm.predict(xe)) for mm in models], axis=0)
    return {"accuracy": float((probs.argmax(1) == ye.reshape(-1)).mean())}


def gpu_bbb_regression(lr, alpha, batch, hidden, seed, steps=2000):
    s = _surface()
    rng = np.random.default_rng(seed)
    x = rng.uniform(1, 20, size=(600, 1)).astype(np.float32)
    ds = s["Dataset"]((x, (2 * x + 2).astype(np.float32)), s["MeanSquaredError"], "Regression", seed=seed)
    opt = s["BBB"]()
    opt.compile(s["HyperParameters"](lr=lr, alpha=alpha, batch_size=batch), s["sequential_json"](1, [hidden, 1], ["linear", "linear"]),
                ds, verbose=False, prior=s["GaussianPrior"](0.0, 1.0), seed=seed)
    opt.train(steps)
    bm, _, _ = opt.result()
    xe, ye = _test_split(ds)
    _, mean = bm.predict(xe, nb_samples=10)
    mean = np.asarray(mean)
    return {"mse": float(((mean - ye.reshape(mean.shape)) ** 2).mean())}


BACKENDS = {"oracle": (oracle_hmc, oracle_svgd_regression, oracle_bbb_classification),
            "gpu": (gpu_hmc, gpu_svgd_regression, gpu_bbb_classification)}
EXTRA = {"oracle": (oracle_svgd_classification, oracle_bbb_regression),
         "gpu": (gpu_svgd_classification, gpu_bbb_regression)}


def replay_extra(backend: str, what=("svgd_cls", "bbb_reg"), seeds=2, svgd_rows=None, bbb_rows=None):
    """The two logs added in round 3 (SVGD_classification_FULL.txt, BBB_regression_FULL.txt): one run per seed and row."""
    f_svgd, f_bbb = EXTRA[backend]
    g = golden()
    out = {}
    if "svgd_cls" in what:
        out["svgd_cls"] = []
        for row in (svgd_rows if svgd_rows is not None else g["svgd_classification"]["rows"]):
            runs = [f_svgd(row["lr"], row["batch_size"], row["M"], seed)["accuracy"] for seed in range(seeds)]
            out["svgd_cls"].append(dict(row, runs=runs, mean=float(np.mean(runs))))
    if "bbb_reg" in what:
        out["bbb_reg"] = []
        for row in (bbb_rows if bbb_rows is not None else g["bbb_regression"]["rows"]):
            runs = [f_bbb(row["lr"], row["alpha"], row["batch_size"], row["hidden_dims"], seed)["mse"] for seed in range(seeds)]
            out["bbb_reg"].append(dict(row, runs=runs, median=float(np.median(runs))))
    return out


def replay(backend: str, what=("hmc", "svgd", "bbb"), seeds=3, hmc_rows=None, svgd_rows=None, bbb_rows=None):
    """-> {"hmc": [{..row.., "reference": logged value, "runs": [..]}], ...}; one run per seed."""
    f_hmc, f_svgd, f_bbb = BACKENDS[backend]
    g = golden()
    out = {}
    if "hmc" in what:
        out["hmc"] = []
        for row in (hmc_rows if hmc_rows is not None else g["hmc_classification"]["rows"]):
            runs = [f_hmc(row["epsilon"], row["m"], row["L"], seed)["accuracy"] for seed in range(seeds)]
            out["hmc"].append(dict(row, runs=runs, mean=float(np.mean(runs))))
    if "svgd" in what:
        out["svgd"] = []
        for row in (svgd_rows if svgd_rows is not None else g["svgd_regression"]["rows"]):
            runs = [f_svgd(row["lr"], row["batch_size"], row["M"], seed)["mse"] for seed in range(seeds)]
            out["svgd"].append(dict(row, runs=runs, median=float(np.median(runs))))
    if "bbb" in what:
        out["bbb"] = []
        for row in (bbb_rows if bbb_rows is not None else g["bbb_classification"]["rows"]):
            runs = [f_bbb(row["lr"], row["alpha"], row["batch_size"], row["hidden_dims"], seed)["accuracy"] for seed in range(seeds)]
            out["bbb"].append(dict(row, runs=runs, mean=float(np.mean(runs))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=list(BACKENDS), default="oracle")
    ap.add_argument("--what", default="hmc,svgd,bbb")
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--contrast", action="store_true", help="also run the HMC rows with a VALID prior scale (+1.0)")
    args = ap.parse_args()
    res = replay(args.backend, tuple(args.what.split(",")), args.seeds)
    for kind, rows in res.items():
        for r in rows:
            print(json.dumps({"kind": kind, **r}), flush=True)
    if args.contrast:
        f_hmc = BACKENDS[args.backend][0]
        for row in golden()["hmc_classification"]["rows"]:
            runs = [f_hmc(row["epsilon"], row["m"], row["L"], seed, prior_sigma=1.0) for seed in range(args.seeds)]
            print(json.dumps({"kind": "hmc_valid_prior", **row, "runs": [r["accuracy"] for r in runs],
                              "accepted": [r["accepted"] for r in runs]}), flush=True)


if __name__ == "__main__":
    main()
