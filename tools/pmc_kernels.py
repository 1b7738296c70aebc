#!/usr/bin/env python3
"""pmc_kernels.py <dir> <preset> — FETCH_SIZE / WRITE_SIZE per dispatch of EVERY kernel of a preset's batch (search, post, score,
count ...) from the rocprofv3 --pmc passes under <dir> (tools/gpu/run.sh ... pmc:<preset>), averaged over the dispatches."""
import csv, glob, os, sys
d, preset = sys.argv[1], sys.argv[2]
tab = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, "pmc_%s_%s" % (c, preset), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                e = tab.setdefault(r["Kernel_Name"], {"FETCH_SIZE": [], "WRITE_SIZE": []})
                e[c].append(float(r["Counter_Value"]))
print("preset %s: KB per dispatch (FETCH_SIZE = 64-byte fetches as counted; WRITE_SIZE as counted), dispatches" % preset)
rows = []
for k, e in tab.items():
    f = sum(e["FETCH_SIZE"]) / max(1, len(e["FETCH_SIZE"]))
    w = sum(e["WRITE_SIZE"]) / max(1, len(e["WRITE_SIZE"]))
    rows.append((f + w, f, w, max(len(e["FETCH_SIZE"]), len(e["WRITE_SIZE"])), k))
for t, f, w, n, k in sorted(rows, reverse=True)[:24]:
    print("%14.0f %14.0f %6d  %s" % (f, w, n, k[:150]))
