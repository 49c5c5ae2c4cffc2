#!/bin/bash
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
for sh in 2 3; do GS_BIN_SHIFT=$sh timeout 120 python tools/build_timing.py 2>&1 | grep -v amdgpu.ids; done
for v in "" _nodpp; do for sh in 2 3; do
  echo "== lib$v shift $sh"; timeout 150 bash tools/prof_quick.sh w${v}_s$sh GS3D_HIP_LIB=$P/libgs3d_hip$v.so GS_BIN_SHIFT=$sh 2>&1 | grep -v amdgpu.ids | grep "gs::"
done; done
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "config_a or every_bin_size or golden or ragged or one_dense or sort_paths or capacity or degenerate or small_counts" 2>&1 | tail -3
