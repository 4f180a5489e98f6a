#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of csrc/pyz_api.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950 without a GPU).

    python tools/kernel_resources.py [filter-substring] [-D...]

Prints one line per kernel; `spill` > 0 or `scratch` > 0 is what the HMC / SVGD kernels must not show."""

from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesian_inference_for_nn_amd", "csrc")


def collect(defines=()):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-function",
               "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", *defines, "pyz_api.hip", "-o",
               os.path.join(tmp, "lib.so")]
        res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if res.returncode != 0:
            raise SystemExit(res.stderr[-4000:])
    rows = []
    for block in res.stderr.split("Function Name: ")[1:]:
        mangled = block.split()[0]

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", block)
            return int(m.group(1)) if m else -1
        rows.append(dict(mangled=mangled, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("TotalSGPRs"), spill=g("VGPRs Spill"),
                         scratch=g("ScratchSize [bytes/lane]"), occ=g("Occupancy [waves/SIMD]"),
                         lds=g("LDS Size [bytes/block]")))
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n)
        r["name"] = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return rows


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("-D")]
    defines = [a for a in sys.argv[1:] if a.startswith("-D")]
    for r in collect(defines):
        if flt and not any(f in r["name"] for f in flt):
            continue
        print(f"{r['name'][:64]:64s} vgpr {r['vgpr']:4d} agpr {r['agpr']:4d} sgpr {r['sgpr']:4d} spill {r['spill']:5d} "
              f"scratch {r['scratch']:5d} occ {r['occ']} lds {r['lds']}")


if __name__ == "__main__":
    main()
