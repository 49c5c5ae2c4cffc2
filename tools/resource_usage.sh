#!/bin/bash
# -Rpass-analysis=kernel-resource-usage for every kernel of the library, as a table (CPU container: hipcc cross-compiles).
# usage: bash tools/resource_usage.sh > profiles/rNN_kernel_resource_usage.txt
cd "$(dirname "$0")/../3dgs.cpp_amd/csrc"
printf "%-64s %6s %6s %6s %8s %6s %9s\n" kernel SGPRs VGPRs AGPRs scratch occ "LDS bytes"
for f in gs_scene.hip gs_preprocess.hip gs_radix.hip gs_bin_l1.hip gs_bin_l2.hip gs_blend.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c -x hip $f -o /dev/null 2>&1 |
  python3 -c '
import re, subprocess, sys
cur = {}
def flush():
    if cur.get("name"):
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "").replace("gs::", "")
        print("%-64s %6s %6s %6s %8s %6s %9s" % (name[:64], cur.get("TotalSGPRs", "?"), cur.get("VGPRs", "?"), cur.get("AGPRs", "?"), cur.get("ScratchSize [bytes/lane]", "?"), cur.get("Occupancy [waves/SIMD]", "?"), cur.get("LDS Size [bytes/block]", "?")))
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        flush(); cur = {"name": v}
    else:
        cur[k] = v
flush()
'
done
