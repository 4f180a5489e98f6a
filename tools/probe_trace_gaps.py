#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory and lists the idle gaps of the device between
consecutive operations (kernels and copies), largest first, with what ran before and after."""
import csv, glob, os, sys
d = sys.argv[1]
ops = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ops.sort()
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ops = ops[lo:]
gaps = []
end = ops[0][1]
for i in range(1, len(ops)):
    g = ops[i][0] - end
    if g > 0:
        gaps.append((g, ops[i - 1][2], ops[i][2], i))
    end = max(end, ops[i][1])
tot = ops[-1][1] - ops[0][0]
print(f"{len(ops)} operations over {tot / 1e3:.1f} us; idle {sum(g for g, *_ in gaps) / 1e3:.1f} us in {len(gaps)} gaps")
for g, a, b, i in sorted(gaps, reverse=True)[:25]:
    print(f"  {g / 1e3:8.1f} us  after {a:42s} before {b} (op {i})")
