#!/usr/bin/env python3
"""Register, scratch and LDS use of every kernel of libyolopost.so, from the compiler's own
remarks (-Rpass-analysis=kernel-resource-usage; cross-compiles for gfx950 without a GPU).

    python tools/kernel_resources.py [-DYP_...] > table.json          print the table
    python tools/kernel_resources.py --check tests/golden/kernel_resources.json [-DYP_...]

--check compares the build with a committed table and exits 1 when a kernel leaves its resource
class: more scratch per lane, an LDS block of which fewer fit a CU, fewer waves per SIMD, or a
kernel the table does not know.  tools/ab_variants.sh and tools/ab_image.sh refuse to build a variant for the GPU
box that fails this check (a build with other occupancy or spills is a different machine load,
to be looked at on paper first)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "2d_object_detection_amd", "csrc")
SOURCES = ("yolopost.hip", "gridcodec.hip", "decode_grad.hip", "metrics.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull]
LDS_PER_CU = 160 * 1024
FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    if out.returncode != 0:
        return names
    res = []
    for n in out.stdout.splitlines():
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        res.append(re.sub(r"\(.*$", "", n))      # name and template arguments, not the parameter list
    return res


def collect(defines, hipcc="/opt/rocm/bin/hipcc"):
    table = {}
    for src in SOURCES:
        out = subprocess.run([hipcc] + FLAGS + list(defines) + [os.path.join(CSRC, src)],
                             capture_output=True, text=True)
        if out.returncode != 0:
            sys.stderr.write(out.stderr[-4000:])
            raise SystemExit("compilation of %s failed" % src)
        cur, mangled = None, []
        for line in out.stderr.splitlines():
            m = re.search(r"remark: +Function Name: (\S+)", line)
            if m:
                cur = {}
                mangled.append((m.group(1), cur))
                continue
            m = re.search(r"remark: +([A-Za-z][A-Za-z \[\]/]*): (\d+)", line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
        for name, (_, vals) in zip(demangle([n for n, _ in mangled]), mangled):
            table[name] = vals
    return table


def check(table, want):
    bad = []
    for name, got in sorted(table.items()):
        ref = want.get(name)
        if ref is None:
            bad.append("%s: not in the table" % name)
            continue
        if got["scratch"] > ref["scratch"]:
            bad.append("%s: scratch %d B/lane (table %d)" % (name, got["scratch"], ref["scratch"]))
        # (LDS: what counts is how many workgroups fit the 160 KB of a CU)
        if got["lds"] > ref["lds"] and LDS_PER_CU // got["lds"] < LDS_PER_CU // max(ref["lds"], 1):
            bad.append("%s: LDS %d B (table %d)" % (name, got["lds"], ref["lds"]))
        if got["occupancy"] < ref["occupancy"]:
            bad.append("%s: %d waves/SIMD (table %d)" % (name, got["occupancy"], ref["occupancy"]))
    return bad


if __name__ == "__main__":
    args = sys.argv[1:]
    ref = None
    if args and args[0] == "--check":
        ref = json.load(open(args[1]))
        args = args[2:]
    table = collect(args)
    if ref is None:
        json.dump(table, sys.stdout, indent=1, sort_keys=True)
        print()
    else:
        bad = check(table, ref)
        for b in bad:
            print(b)
        sys.exit(1 if bad else 0)
