#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC passes of one tools/gpu/run.sh session (pmc:<preset> ..., mix:2): the per-preset traffic of
the search kernel and, for preset 2, its SQ counters per launch — stamped with the sha256 of the kernel sources they were collected on
(bench.py quotes them only while the sources are still those).
usage: tools/assemble_pmc_json.py gpurun_out/<tag> [gpurun_out/pmc_<tag>_mix_2]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(run_dir, mix_dir=None):
    import bench
    old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    out = {"_comment": old["_comment"], "formula": old["formula"], "kernel_source_sha256": bench.kernel_source_sha(), "presets": {},
           "collected_by": "tools/gpu/run.sh %s" % os.path.basename(run_dir.rstrip("/"))}
    for f in sorted(os.listdir(run_dir)):
        m = re.match(r"pmc_traffic_(\w+-?)\.json$", f)
        if not m:
            continue
        try:
            out["presets"][m.group(1)] = json.load(open(os.path.join(run_dir, f)))
        except Exception as e:        # noqa: BLE001
            print("skipped", f, e, file=sys.stderr)
    if mix_dir and "2" in out["presets"]:
        sq = {}
        kernel = re.search(r"(k_\w+(<[^>]*>)?)\(", out["presets"]["2"]["kernel"]).group(1)      # k_search2_l1<4, false, 0, false>
        for ln in open(os.path.join(mix_dir, "summary.txt")):
            if ln.startswith(kernel + " "):
                for name, val in re.findall(r"(SQ_\w+)=([0-9.e+]+)", ln):
                    sq[name] = float(val)
        if sq:
            out["presets"]["2"]["sq_counters_per_launch"] = sq
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v.get("traffic_bytes_per_launch") for k, v in out["presets"].items()}), out["kernel_source_sha256"][:12], sorted(out["presets"].get("2", {}).get("sq_counters_per_launch", {})))


if __name__ == "__main__":
    main(*sys.argv[1:3])
