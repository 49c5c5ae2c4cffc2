#!/bin/bash
# why is the hand-written pair loop slower at T(6e6)?  counters of k_blend for the compiler's loop (l0) and the hand-written one (cur)
R=$PWD; O=$R/gpurun_out/blendT; mkdir -p $O; exec < /dev/null
cd /tmp && export TMPDIR=/tmp
for v in l0 cur; do
  L=$R/3dgs.cpp_amd/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$R/3dgs.cpp_amd/libgs3d_hip.so
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    GS3D_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${v}_$tag -o p -- python $R/tools/tune_sweep.py --no-prime --batches 1 --fif 1 --frames 3 --warm 2 --gaussians 6000000 --scene T > /dev/null 2>&1
  done
done
cd $R; find $O -name '*_kernel_trace.csv' -delete
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/blendT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if "k_blend" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[2], {c: f"{x:.4g}" for c, x in acc.items()})
PY
