#!/bin/bash
# round-3 GPU trip 3: LDS-DMA SH fetch A/B, scalarised k_bin_fast, cooperative level-1 emission
set -u
exec < /dev/null
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests_3.log 2>&1
tail -3 gpurun_out/r03_gputests_3.log | head -1
P=3dgs.cpp_amd
{
  echo "== build timing B"; N=1000000 W=1920 H=1080 timeout 120 python tools/build_timing.py
  echo "== build timing E"; N=6000000 W=3840 H=2160 timeout 200 python tools/build_timing.py
  for rep in 1 2; do for v in cur presh0 presh24; do
    L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
    echo "== sweep B $v"; GS3D_HIP_LIB=$L timeout 120 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 3 --ref-image /tmp/ref_B.npy
  done; done
  for v in cur presh0 presh24; do
    L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
    echo "== sweep E $v"; GS3D_HIP_LIB=$L timeout 200 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 2 --frames 60 --gaussians 6000000 --width 3840 --height 2160 --ref-image /tmp/ref_E.npy
  done
  echo "== sweep C"; timeout 200 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 2 --frames 100 --gaussians 6000000
  echo "== sweep T"; timeout 200 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 2 --frames 100 --gaussians 6000000 --scene T
} > gpurun_out/r03_trip3.log 2>&1
grep -v "^#" gpurun_out/r03_trip3.log | tail -60
