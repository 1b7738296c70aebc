#!/bin/bash
# VGPRs / scratch / LDS of every kernel in a HIP source (cross-compiles for gfx950; no GPU needed).
# usage: tools/kernel_resources.sh centrifuge_amd/csrc/cf_device.hip [grep-pattern]
set -e
src=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -c "$src" -o $tmp/dev.bundle 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/dev.bundle --output=$tmp/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c '
import sys, re
rows, cur, inK = [], None, False
for ln in sys.stdin:
    if "amdhsa.kernels:" in ln: inK = True; continue
    if not inK: continue
    if re.match(r"\s{2}- \.", ln): cur = {}; rows.append(cur); ln = ln.replace("- ", "  ", 1)     # a new kernel entry
    if re.match(r"\S", ln) and not ln.startswith(" "): inK = False; continue
    m = re.match(r"\s{4}\.(\w+):\s+(.*)", ln)
    if m and cur is not None and m.group(1) in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"): cur[m.group(1)] = m.group(2).strip()
for r in rows:
    n = r.get("name", "")
    if "hipcub" in n or "rocprim" in n or not n: continue
    print("%s vgpr %3s sgpr %3s scratch %5s lds %6s" % (n, r.get("vgpr_count"), r.get("sgpr_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
' | grep -E "$pat"
rm -rf $tmp
