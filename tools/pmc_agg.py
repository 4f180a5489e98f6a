#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: python tools/pmc_agg.py <dir-with-*counter_collection.csv> ...
Prints one JSON line per kernel: {kernel, launches, COUNTER: mean per launch, ...} (all passes merged)."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        pats = [d] if d.endswith(".csv") else glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        for path in pats:
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].replace("void ", "").split("(")[0]
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in sorted(acc.items()):
        out = {"kernel": name, "launches": max(len(v) for v in cs.values())}
        out.update({c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())})
        print(json.dumps(out))


if __name__ == "__main__":
    main()
