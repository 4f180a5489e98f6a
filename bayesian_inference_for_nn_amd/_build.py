"""Builds csrc/libpyz.so (hand-written HIP for gfx950) in-tree with hipcc."""

from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpyz.so")
SOURCES = ["pyz_api.hip"]
HEADERS = ["pyz_common.h", "pyz_gemm.h", "pyz_kernels.h", "pyz_rng.h", "pyz_hmc_fused.h",
           os.path.join("..", "..", "include", "pyz.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if it is missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 library")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-Wno-pass-failed"] + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
