#!/usr/bin/env python3
"""Instruction histogram of one kernel: python tools/asm_histogram.py <mangled-name-substring> [-D...]
(hipcc -S of csrc/pyz_api.hip for gfx950; prints the most frequent opcodes, SGPR-spill lane moves and the
register counts of every kernel whose symbol contains the substring)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesian_inference_for_nn_amd", "csrc")


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("-D")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-function",
                        "-Wno-pass-failed", *defs, "pyz_api.hip", "-o", out], cwd=CSRC, check=True, capture_output=True)
        s = open(out).read()
    for m in re.finditer(r"^(_Z\w+):\s*;[^\n]*\n(.*?)\.end_amdhsa_kernel", s, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        c = collections.Counter(re.findall(r"^\s+([a-z_0-9]+)", body, flags=re.M))
        tail = s[m.end():m.end() + 3000]
        regs = re.findall(r"; (?:TotalNumSgprs|NumVgprs|NumAgprs|ScratchSize|Occupancy): \d+", tail)[:5]
        print(name, regs)
        print("   lane spills:", c["v_writelane_b32"], "writes /", c["v_readlane_b32"], "reads;  top:", c.most_common(14))


if __name__ == "__main__":
    main()
