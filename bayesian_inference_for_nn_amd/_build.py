"""Builds csrc/libpyz.so (hand-written HIP for gfx950) in-tree with hipcc."""

from __future__ import annotations

import fcntl
import glob
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpyz.so")
SOURCES = ["pyz_api.hip"]


class NoCompiler(RuntimeError):
    """hipcc is not installed (as opposed to: hipcc ran and failed)."""


def _inputs():
    """Everything the library is compiled from: the translation unit, every header beside it, the C-ABI header."""
    return ([os.path.join(CSRC, f) for f in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) +
            [os.path.join(CSRC, "..", "..", "include", "pyz.h")])


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in _inputs())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if it is missing or older than its sources; returns its path.
    Safe when several ranks of one node call it at once: one of them compiles (under a file lock, into a
    temporary file that is renamed into place), the others wait and find the fresh library."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise NoCompiler("hipcc not found: cannot build the gfx950 library")
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():        # another process built it while this one waited
                return LIB
            tmp = LIB + f".tmp{os.getpid()}"
            cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall",
                   "-Wno-unused-function", "-Wno-pass-failed"] + SOURCES + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
