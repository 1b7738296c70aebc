#!/usr/bin/env python3
"""pmc_calib.py <dir> — FETCH_SIZE / WRITE_SIZE per request of tools/microbench/fetch_calib.hip's kernels (known request and byte
counts) from the two rocprofv3 --pmc passes under <dir>/calib_FETCH_SIZE, <dir>/calib_WRITE_SIZE (tools/gpu/run.sh ... calib)."""
import csv, glob, os, re, sys
d = sys.argv[1]
known = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        for ln in open(os.path.join(d, "calib_%s.txt" % c)):
            m = re.match(r"(\S+) requests (\d+) bytes_requested (\d+)", ln)
            if m:
                known[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    except OSError:
        pass
print("kernel                 counter      KB/dispatch      requests   bytes asked for   counter bytes per request   counter / bytes asked for")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(os.path.join(d, "calib_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(per.items()):
        name = next((n for n in known if k.startswith("void " + n) or k.startswith(n)), None)
        if name is None:
            continue
        kb = sum(v) / len(v)
        req, asked = known[name]
        print("%-22s %-10s %14.0f %14d %16d %18.1f %24.3f" % (name, c, kb, req, asked, kb * 1000.0 / req, kb * 1000.0 / asked))
