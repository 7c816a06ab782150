"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (developer tool).

    python tools/pmc_summary.py <dir-with-*counter_collection.csv> [...]
Prints JSON: {kernel: {counter: {"sum":..., "dispatches":..., "avg":...}}}
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name") or row.get("Kernel Name")
                    c = row.get("Counter_Name") or row.get("Counter Name")
                    v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                    if not k or "gn::" not in k:
                        continue
                    k = k.replace("void ", "").replace("gn::(anonymous namespace)::", "").replace("gn::", "")
                    k = k.split("(")[0]
                    out[k][c][0] += v
                    out[k][c][1] += 1
    res = {k: {c: {"sum": s, "dispatches": n, "avg": s / max(n, 1)} for c, (s, n) in cs.items()} for k, cs in out.items()}
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
