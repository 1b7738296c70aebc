#!/usr/bin/env python3
"""make_pmc_json.py <gpu_profile dir> — profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from the rocprofv3
--pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --config 2` in that directory, stamped with the sha256 of the kernel sources
the passes ran on (bench.py gives the figure only while the sources are still those)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
d = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_search2_l1<4, false, 0, false>"
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
                v.append(float(r["Counter_Value"]))
    if not v:
        sys.exit("no %s rows for %s under %s" % (c, kernel, d))
    vals[c] = sum(v) / len(v)
P = bench.PRESETS["2"]
out = {
    "_comment": "HBM traffic of the dominant kernel from rocprofv3 --pmc passes of `bench.py --config 2` (tools/gpu/run.sh ... pmc). The kernel's loads are 8- and "
                "16-byte requests to random lines; FETCH_SIZE = TCC_EA_RDREQ x 64 B then counts one 64-byte fetch per request and is taken as is (the x2 "
                "correction of MI355X_MICROARCH.md applies to 128-byte requests tallied at 64, which this kernel does not make). WRITE_SIZE is uncalibrated "
                "and small. Counter values are KB per dispatch, averaged over the dispatches of the run.",
    "kernel": kernel, "preset": "2", "genomes": P["genomes"], "genome_len": P["genome_len"], "reads": P["reads"], "read_len": P["read_len"],
    "fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
    "formula": "FETCH_SIZE + WRITE_SIZE (64-byte fetches of 8/16-byte requests: no x2)",
    "traffic_bytes_per_launch": int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1000),
    "kernel_source_sha256": bench.kernel_source_sha(),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=2)
print(json.dumps(out, indent=2))
