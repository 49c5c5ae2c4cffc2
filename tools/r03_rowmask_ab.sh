#!/bin/bash
# round-2 verdict item 9: the row-interval lane mask in the blend -- A/B against the shipped library (serial k_blend, frames/s,
# SQ_INSTS_VALU per launch), every frame compared bit for bit with the first one
set -u
exec < /dev/null
R=$(pwd); P=$R/3dgs.cpp_amd; O=$R/gpurun_out/rowmask_ab; mkdir -p $O
python -m pytest tests/test_gpu_blend_modes.py -x -q 2>&1 | tail -2
for rep in 1 2; do for v in cur rowmask; do
  L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
  echo "== sweep B $v"; GS3D_HIP_LIB=$L timeout 120 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 3 --ref-image /tmp/ref_B_rm.npy
done; done
cd /tmp && export TMPDIR=/tmp
for v in cur rowmask; do
  L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
  GS3D_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $O/$v -o p -- python $R/tools/tune_sweep.py --no-prime --batches 1 --fif 1 --frames 3 --warm 2 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/$v/*/p_counter_collection.csv k_blend 2>/dev/null || python $R/tools/pmc_summary.py $(find $O/$v -name '*counter_collection.csv' | head -1) k_blend
done
