#!/usr/bin/env python3
"""All 27 rows of the reference's logs/HMC_classification_FULL.txt replayed with the CPU oracle under the two
readings of GaussianPrior(0.0, -1.0):
  A2     tfp Normal.log_prob does not validate its scale: log(-1) = NaN, every Metropolis test fails, the
         posterior is the state the 10 forced burn-in trajectories reach (the oracle's
         and the library's behaviour);
  valid  the prior scale taken as +1.0: normal acceptance over the 100 sampling steps.
Writes tests/golden/hmc_log_hypotheses.json: per row the logged accuracy and the accuracies of `--seeds` runs
under each reading, and the mean absolute deviation of the log from each reading's mean.  (About 20 minutes of
CPU for 3 seeds; test infrastructure, never imported by the product.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_reference_logs as rr  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rows = rr.golden()["hmc_classification"]["all_rows"]
    out = []
    for row in rows:
        a2 = [rr.oracle_hmc(row["epsilon"], row["m"], row["L"], s, prior_sigma=-1.0)["accuracy"] for s in range(seeds)]
        va = [rr.oracle_hmc(row["epsilon"], row["m"], row["L"], s, prior_sigma=1.0)["accuracy"] for s in range(seeds)]
        out.append(dict(row, A2=a2, valid=va))
        print(json.dumps(out[-1]), flush=True)
    ref = np.array([r["reference_accuracy"] for r in out])
    a2m = np.array([np.mean(r["A2"]) for r in out])
    vam = np.array([np.mean(r["valid"]) for r in out])
    summary = {"rows": len(out), "seeds": seeds,
               "mean_abs_dev_log_vs_A2": float(np.abs(ref - a2m).mean()), "mean_abs_dev_log_vs_valid": float(np.abs(ref - vam).mean()),
               "mean_signed_log_minus_A2": float((ref - a2m).mean()), "mean_signed_log_minus_valid": float((ref - vam).mean()),
               "rows_where_log_is_closer_to_A2": int((np.abs(ref - a2m) < np.abs(ref - vam)).sum())}
    json.dump({"summary": summary, "rows": out}, open(os.path.join(ROOT, "tests", "golden", "hmc_log_hypotheses.json"), "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
