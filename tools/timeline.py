#!/usr/bin/env python3
"""timeline.py <rocprofv3 dir> [search-kernel substring] [first launch] — how the kernels of the pipelined steps sit on the device's timeline.
Reads <dir>/*_kernel_trace.csv (rocprofv3 --kernel-trace --output-format csv), takes the window between the 4th and the 9th
launch of the search kernel (or five steps from launch `first launch`, counted from 0: bench.py's resident-input region begins at
launch max(warmup, slots) + steps + min(warmup, slots)) and prints, per kernel name: launches, mean duration, and how much of its time another kernel of
the listed set was running beside it; then the window's wall time, the sum of kernel durations in it and the time no kernel ran."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_search2"
f = glob.glob(d + "/*kernel_trace.csv")[0]
K = [r for r in csv.DictReader(open(f))]
name = lambda r: r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("cfamd::", "").split("(")[0]
S = sorted((r for r in K if pat in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
if len(S) < 10:
    sys.exit("fewer than 10 launches of %s" % pat)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if len(S) < first + 6:
    sys.exit("fewer than %d launches of %s" % (first + 6, pat))
t0, t1 = int(S[first]["Start_Timestamp"]), int(S[first + 5]["Start_Timestamp"])
W = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r), r.get("Queue_Id", "?")) for r in K
            if t0 <= int(r["Start_Timestamp"]) < t1), key=lambda x: x[0])
ev = sorted([(s, 1) for s, e, n, q in W] + [(e, -1) for s, e, n, q in W])
busy = over = 0; depth = 0; last = t0
for t, dlt in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += dlt; last = t
tot = sum(e - s for s, e, n, q in W)
print("window: 5 steps, %.2f ms per step; kernels sum %.2f ms per step; some kernel running %.2f ms per step; two or more %.2f ms per step; idle %.2f ms per step" %
      ((t1 - t0) / 5e6, tot / 5e6, busy / 5e6, over / 5e6, (t1 - t0 - busy) / 5e6))
agg = collections.OrderedDict()
for s, e, n, q in W:
    a = agg.setdefault(n, [0, 0, 0, set()])
    a[0] += 1; a[1] += e - s; a[3].add(q)
for n, (c, dur, _, qs) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%-40s launches/step %5.1f  mean %8.1f us  per step %7.2f ms  queues %s" % (n[:40], c / 5, dur / c / 1e3, dur / 5e6, sorted(qs)))
# one step in order
s0 = int(S[first + 2]["Start_Timestamp"]); s1 = int(S[first + 3]["Start_Timestamp"])
print("\none step, launch order (start offset us, duration us, queue):")
for s, e, n, q in W:
    if s0 <= s < s1: print("  %9.1f %9.1f  q%-3s %s" % ((s - s0) / 1e3, (e - s) / 1e3, q, n[:50]))
