#!/usr/bin/env python3
"""make_pmc_json.py <dir> <preset> [<kernel substring>] — one preset's entry of profiles/pmc_traffic.json (what bench.py reports
as roofline.traffic) from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --config <preset>` under <dir>
(pmc_FETCH_SIZE_<preset>/, pmc_WRITE_SIZE_<preset>/: tools/gpu/run.sh pmc:<preset>).  The file is keyed by preset and stamped with the
sha256 of the kernel sources the passes ran on: bench.py gives the figure only while the sources are still those, and entries
collected on other sources are dropped when a new one comes in."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
d, preset = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "k_search2"
vals, names = {}, set()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(os.path.join(d, "pmc_%s_%s" % (c, preset), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    if not per:
        sys.exit("no %s rows for %s under %s" % (c, kernel, d))
    # the search kernel of the preset = the matching kernel with the largest total (the instrumented COUNT build runs once)
    name = max(per, key=lambda k: sum(per[k]))
    names.add(name)
    vals[c] = sum(per[name]) / len(per[name])
P = bench.PRESETS[preset]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
sha = bench.kernel_source_sha()
try:
    out = json.load(open(path))
    if out.get("kernel_source_sha256") != sha or "presets" not in out:
        out = {}
except Exception:
    out = {}
out.setdefault("_comment", "HBM traffic of each preset's search kernel from rocprofv3 --pmc passes of `bench.py --config <preset>` (tools/gpu/run.sh ... pmc:<preset>). "
               "The kernel's loads are 8- and 16-byte requests to random lines.  FETCH_SIZE = TCC_EA_RDREQ x 64 B; calibrated on this access pattern with "
               "tools/microbench/random_granule (profiles/r05_fetch_size_calibration.txt): one 64-byte fetch per random 16-byte request, the counter is taken "
               "as is (the x2 correction of MI355X_MICROARCH.md applies to 128-byte requests tallied at 64, which this kernel does not make).  Counter values are "
               "KB per dispatch, averaged over the dispatches of the run.")
out["formula"] = "FETCH_SIZE + WRITE_SIZE (64-byte fetches of 8/16-byte requests: no x2)"
out["kernel_source_sha256"] = sha
out.setdefault("presets", {})[preset] = {
    "kernel": sorted(names)[0], "genomes": P["genomes"], "genome_len": P["genome_len"], "reads": P["reads"], "read_len": P["read_len"],
    "fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
    "traffic_bytes_per_launch": int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1000),
}
json.dump(out, open(path, "w"), indent=2)
print(json.dumps(out["presets"][preset], indent=2))
