#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory (rocprofv3 csv files) into a
small text summary fit for profiles/: per-kernel stats and per-kernel PMC means."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    return name[-60:]


def main(d):
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        print("== kernel stats (%s)" % os.path.relpath(f, d))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:14]:
            print("%-62s calls %6s total_ns %14s avg_ns %12s pct %6s" % (
                short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        seen = {}
        for r in rows:
            k = short(r.get("Kernel_Name", ""))
            if k not in seen:
                seen[k] = r
        print("== kernel resources (first dispatch of each)")
        for k, r in seen.items():
            print("%-62s grid %10s wg %5s vgpr %4s accum %4s sgpr %4s lds %6s scratch %s" % (
                k, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
                r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
    for pd in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(pd):
            continue
        for f in sorted(glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[short(r.get("Kernel_Name", ""))][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
            print("== pmc %s (mean per dispatch)" % os.path.basename(pd))
            for k, cs in acc.items():
                if not any(x in k for x in ("k_search", "k_walk", "k_score", "k_post", "k_emit", "k_restore")):
                    continue
                print("%-62s %s" % (k, "  ".join("%s=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in cs.items())))


if __name__ == "__main__":
    main(sys.argv[1])
