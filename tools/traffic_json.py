#!/usr/bin/env python3
"""Turn tools/pmc_summary.py output (FETCH_SIZE / WRITE_SIZE means per launch, KiB) of the front
kernel into the per-launch HBM traffic record bench.py reports as roofline.traffic.
Usage: python tools/traffic_json.py <pmc_summary.json> <config> <sets_per_launch> <out.json>"""
import json
import re
import sys

src, cfg, sets, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
s = json.load(open(src))
name = [k for k in s if "k_front_heads_v3" in k][0]
# (the launches of the largest grid: the bench's launches of `sets` batches, not its single-batch passes)
fetch, write = s[name]["FETCH_SIZE"]["mean_largest_grid"], s[name]["WRITE_SIZE"]["mean_largest_grid"]
try:
    out = json.load(open(dst))
except (OSError, ValueError):
    out = {}
out[cfg] = {
    "kernel": re.search(r"k_front_heads_v3\w*(<[^>]*>)?", name).group(0), "sets_per_launch": sets,
    "launches_averaged": s[name]["FETCH_SIZE"]["launches_largest_grid"],
    "FETCH_SIZE_KiB_raw": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
    "fetch_bytes_corrected": int(2 * fetch * 1024), "write_bytes": int(write * 1024),
    "traffic_bytes": int(2 * fetch * 1024 + write * 1024),
    "note": "per launch; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of a coalesced stream "
            "at 64 B; calibrated on a 387 MB input (profiles/r01_front_kernel_traffic.json) where 2*FETCH matched the input bytes to "
            "0.03%); separate rocprofv3 --pmc passes of `python3 bench.py --no-cpu-baseline --steps 20 "
            "--warmup 2` (tools/profile_round.sh)",
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out[cfg]))
