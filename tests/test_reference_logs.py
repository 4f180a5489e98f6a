"""Statistical pin against the ONLY outputs the reference holds for this path: the grid-search logs under
/root/reference/logs (rows copied as numbers into tests/golden/reference_logs.json, with line numbers).

The logs are single unseeded runs per row (test-set accuracy on 200 rows / MSE on 60 rows), so what can be
asserted is coarse: bands and orderings over the means of a few seeded replays.  What this pins:
  * BBB.py:107-124,152-201 -- the alpha-weighted KL term and its closed-form gradients: alpha = 0 learns the
    moons (80 - 98.5 %), alpha >= 0.1 ends at chance through a diverged posterior and predict's NaN -> 0
    (BBB_classification_FULL.txt);
  * the per-particle Keras legacy Adam of SVGD (Appendix A3): MSE decades by learning rate
    (SVGD_regression_FULL.txt);
  * HMC with GaussianPrior(0.0, -1.0) (HMC_classification_FULL.txt): accuracy ordered by epsilon * L / m, and the
    NaN-potential reading (Appendix A2: no sampling step is accepted) reproduces the logged accuracies.
What it does NOT pin: bit-level parity, the RNG streams, nor A2 against a validated-scale reading on the rows
where both give the same accuracy (tools/hmc_log_hypotheses.py quantifies both readings over all 27 rows).
Parity stays "partial" (DESIGN.md section 2).

CPU tests replay with the float64 oracle; `-m gpu` tests replay through the drop-in surface on the HIP kernels."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import replay_reference_logs as rr  # noqa: E402


def _rows(kind, **match):
    rows = rr.golden()[kind]["rows"]
    return [r for r in rows if all(r[k] == v for k, v in match.items())]


def check_bbb(res):
    for r in res["bbb"]:
        if r["alpha"] == 0.0:
            assert r["mean"] >= 80.0 and abs(r["mean"] - r["reference_accuracy"]) <= 9.0, r
        else:
            assert r["mean"] <= 65.0 and r["reference_accuracy"] <= 65.0, r
    learns = [r["mean"] for r in res["bbb"] if r["alpha"] == 0.0]
    chance = [r["mean"] for r in res["bbb"] if r["alpha"] > 0.0]
    assert min(learns) > max(chance) + 15.0


def check_svgd(res):
    by_lr = {}
    for r in res["svgd"]:
        by_lr.setdefault(r["lr"], []).append(r)
    for r in by_lr[0.1]:
        assert r["median"] <= 1e-3 and r["reference_mse"] <= 1e-3, r
    for r in by_lr[0.01]:
        assert 1e-5 <= r["median"] <= 1.0 and 1e-5 <= r["reference_mse"] <= 1.0, r
    for r in by_lr[0.001]:
        assert r["median"] >= 10.0 and r["reference_mse"] >= 10.0, r


def check_hmc(res, need_all=True):
    high = [r for r in res["hmc"] if r["reference_accuracy"] >= 95.0]
    mid = [r for r in res["hmc"] if 85.0 <= r["reference_accuracy"] < 95.0]
    low = [r for r in res["hmc"] if r["reference_accuracy"] < 85.0]
    for r in high:
        assert r["mean"] >= 92.0, r
    for r in mid:
        assert 78.0 <= r["mean"] <= 96.0 and abs(r["mean"] - r["reference_accuracy"]) <= 8.0, r
    for r in low:
        assert r["mean"] < 85.0, r
    if need_all:
        assert len(high) >= 3 and len(mid) >= 3 and len(low) >= 2
        assert np.mean([r["mean"] for r in low]) < np.mean([r["mean"] for r in mid]) < np.mean([r["mean"] for r in high])


# ------------------------------------------------------------------ CPU: the oracle against the logs
def test_oracle_reproduces_the_bbb_log_bands():
    check_bbb(rr.replay("oracle", ("bbb",), seeds=2))


def test_oracle_reproduces_the_svgd_log_decades():
    rows = _rows("svgd_regression", M=10) + _rows("svgd_regression", lr=0.1, M=5) + _rows("svgd_regression", lr=0.001)
    check_svgd(rr.replay("oracle", ("svgd",), seeds=3, svgd_rows=rows))


def test_oracle_reproduces_hmc_log_rows_under_the_nan_potential():
    rows = [r for r in rr.golden()["hmc_classification"]["rows"] if (r["epsilon"], r["m"], r["L"]) in
            {(0.005, 0.5, 10), (0.005, 0.5, 30), (0.005, 2.0, 10)}]
    res = rr.replay("oracle", ("hmc",), seeds=2, hmc_rows=rows)
    check_hmc(res, need_all=False)
    one = rr.oracle_hmc(0.005, 0.5, 10, seed=0)
    assert one["accepted"] == 0 and one["distinct_samples"] == 1      # Appendix A2: only the burn-in moved q


def test_committed_27_row_analysis_favours_the_nan_potential_reading():
    """tests/golden/hmc_log_hypotheses.json is the output of tools/hmc_log_hypotheses.py (all 27 rows x 3 seeds x the two
    readings of GaussianPrior(0.0, -1.0), ~20 CPU-minutes with the oracle; not recomputed here).  Appendix A2 (NaN
    potential: nothing accepted after the burn-in) is unbiased against the log (mean signed deviation within 2
    points), the validated-scale reading overshoots it by more than 3 points on average, and most rows sit
    closer to A2."""
    import json
    s = json.load(open(os.path.join(ROOT, "tests", "golden", "hmc_log_hypotheses.json")))["summary"]
    assert s["rows"] == 27 and s["seeds"] >= 3
    assert abs(s["mean_signed_log_minus_A2"]) < 2.0 and s["mean_signed_log_minus_valid"] < -3.0
    assert s["mean_abs_dev_log_vs_A2"] < s["mean_abs_dev_log_vs_valid"] and s["rows_where_log_is_closer_to_A2"] >= 15


def check_svgd_cls(res):
    """logs/SVGD_classification_FULL.txt: 48 rows, accuracies 0.83 - 0.99 with no trend in lr / batch size / M (moons is
    easy for 3 - 20 particles after 1 000 steps): what the log pins is the BAND -- a replay outside it would mean the
    loss-gradient drive, the legacy Adam or the ensemble-mean read-out (SVGD.py:104-120, SVGD_classification.py:86-93)
    is off."""
    band = rr.golden()["svgd_classification"]["band"]
    assert band["n_rows"] == 48 and band["min_accuracy"] >= 0.8
    for r in res["svgd_cls"]:
        assert band["min_accuracy"] - 0.03 <= r["mean"] <= 1.0, r


def check_bbb_reg(res):
    """logs/BBB_regression_FULL.txt (y = 2x + 2, 2 000 steps, posterior rho from the raw prior rho = 1): MSEs of
    0.05 - 11 on a target of variance ~120.  Rows with lr <= 5e-4 replay within a factor 30 of the log (one unseeded run
    per row there); lr = 1e-4 ends above lr = 5e-4 like in the log (alpha = 0, one hidden unit: 1.72 / 0.80 against
    0.059 / 0.053).  lr = 1e-3 is marginal for inputs up to 20 (the log itself holds 9.2 and 10.8 there; a replay may
    diverge) and is not asserted.  Pins the closed-form rho / mu updates of BBB.py:152-201 over thousands of steps."""
    rows = [r for r in res["bbb_reg"] if r["lr"] <= 5e-4]
    assert len(rows) >= 4
    for r in rows:
        assert r["reference_mse"] / 30.0 <= r["median"] <= r["reference_mse"] * 30.0 and r["median"] < 30.0, r
    slow = [r["median"] for r in rows if r["lr"] == 1e-4 and r["alpha"] == 0.0]
    fast = [r["median"] for r in rows if r["lr"] == 5e-4 and r["alpha"] == 0.0]
    assert slow and fast and np.mean(slow) > np.mean(fast)


def test_oracle_reproduces_the_svgd_classification_band():
    rows = [r for r in rr.golden()["svgd_classification"]["rows"] if r["M"] <= 10]
    check_svgd_cls(rr.replay_extra("oracle", ("svgd_cls",), seeds=1, svgd_rows=rows))


def test_oracle_reproduces_the_bbb_regression_log():
    check_bbb_reg(rr.replay_extra("oracle", ("bbb_reg",), seeds=3))


# ------------------------------------------------------------------ GPU: the drop-in surface against the logs
@pytest.mark.gpu
def test_gpu_surface_reproduces_the_svgd_classification_band(gpu_device):
    check_svgd_cls(rr.replay_extra("gpu", ("svgd_cls",), seeds=2))


@pytest.mark.gpu
def test_gpu_surface_reproduces_the_bbb_regression_log(gpu_device):
    check_bbb_reg(rr.replay_extra("gpu", ("bbb_reg",), seeds=3))


@pytest.mark.gpu
def test_gpu_surface_reproduces_the_bbb_log_bands(gpu_device):
    check_bbb(rr.replay("gpu", ("bbb",), seeds=3))


@pytest.mark.gpu
def test_gpu_surface_reproduces_the_svgd_log_decades(gpu_device):
    check_svgd(rr.replay("gpu", ("svgd",), seeds=3))


@pytest.mark.gpu
def test_gpu_surface_reproduces_the_hmc_log_grid(gpu_device):
    res = rr.replay("gpu", ("hmc",), seeds=3)
    check_hmc(res)
    one = rr.gpu_hmc(0.005, 0.5, 30, seed=1)
    assert one["accepted"] == 0 and one["distinct_samples"] == 1
