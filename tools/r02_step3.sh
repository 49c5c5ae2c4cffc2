#!/bin/bash
# level-1 / level-2 kernel variants: per-kernel durations (one frame in flight), then a quick parity check of each library
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
for v in "" _wl; do for sh in 2 3; do
  echo "== lib$v shift $sh"; timeout 150 bash tools/prof_quick.sh v${v}_s$sh GS3D_HIP_LIB=$P/libgs3d_hip$v.so GS_BIN_SHIFT=$sh 2>&1 | grep -v amdgpu.ids | grep "gs::"
done; done
for v in "" _wl; do
  GS3D_HIP_LIB=$P/libgs3d_hip$v.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "config_a or every_bin_size or golden or ragged or one_dense or sort_paths or capacity" 2>&1 | tail -3
done
