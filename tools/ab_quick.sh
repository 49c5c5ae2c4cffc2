#!/bin/bash
# A/B of library builds with tools/tune_sweep.py --quick in one gpurun call: tools/ab_quick.sh tagA tagB ...
# (3dgs.cpp_amd/libgs3d_hip_<tag>.so; "cur" = the default library).  The first build's frame is the reference image
# the others must reproduce bit for bit.
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; mkdir -p $O; exec < /dev/null
rm -f /tmp/ab_ref.npy
for v in "$@"; do
  L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
  echo "== $v"
  GS3D_HIP_LIB=$L timeout 60 python tools/tune_sweep.py --quick --frames 300 --batches 3 --ref-image /tmp/ab_ref.npy $AB_ARGS 2>&1 | tail -5
done | tee $O/ab_quick.txt
