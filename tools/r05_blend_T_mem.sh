#!/bin/bash
R=$PWD; O=$R/gpurun_out/blendTmem; mkdir -p $O; exec < /dev/null
cd /tmp && export TMPDIR=/tmp
for v in l0 cur; do
  L=$R/3dgs.cpp_amd/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$R/3dgs.cpp_amd/libgs3d_hip.so
  i=0
  for set in "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    GS3D_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${v}_$i -o p -- python $R/tools/tune_sweep.py --no-prime --batches 1 --fif 1 --frames 3 --warm 2 --gaussians 6000000 --scene T > /dev/null 2>&1
  done
done
cd $R; find $O -name '*_kernel_trace.csv' -delete
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/blendTmem/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_blend" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[2], {c: f"{x:.4g}" for c, x in acc.items()})
PY
