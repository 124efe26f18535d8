#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, the mean counter value
per launch -- over all launches, and over the launches of the kernel's largest grid alone (a
bench run also holds single-batch launches of the same kernels: its pilot and latency passes).
Usage: python tools/pmc_summary.py <dir-or-csv> [...]"""
import collections
import csv
import glob
import json
import os
import sys


def summarise(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        per_dispatch = collections.defaultdict(float)
        names, grids = {}, {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
            grids[r["Dispatch_Id"]] = int(r.get("Grid_Size") or 0)
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append((grids[disp], v))
    out = {}
    for k, ctrs in acc.items():
        out[k] = {}
        for c, gv in ctrs.items():
            top = max(g for g, _ in gv)
            big = [v for g, v in gv if g == top]
            out[k][c] = {"launches": len(gv), "mean": sum(v for _, v in gv) / len(gv), "largest_grid": top,
                         "launches_largest_grid": len(big), "mean_largest_grid": sum(big) / len(big)}
    return out


if __name__ == "__main__":
    res = {}
    for p in sys.argv[1:]:
        for k, v in summarise(p).items():
            res.setdefault(k, {}).update(v)
    print(json.dumps(res, indent=1))
