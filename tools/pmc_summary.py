#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, the mean counter value
per launch.  Usage: python tools/pmc_summary.py <dir-or-csv> [...]"""
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
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
    out = {}
    for k, ctrs in acc.items():
        out[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in ctrs.items()}
    return out


if __name__ == "__main__":
    res = {}
    for p in sys.argv[1:]:
        for k, v in summarise(p).items():
            res.setdefault(k, {}).update(v)
    print(json.dumps(res, indent=1))
