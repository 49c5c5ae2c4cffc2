#!/bin/bash
# round-3 GPU trip 2: the record-streaming k_bin_fast -- correctness, phase timing, mode-independent sweeps, work counters
set -u
exec < /dev/null
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests_2.log 2>&1
tail -3 gpurun_out/r03_gputests_2.log
{
  echo "== valu_rate"; timeout 120 tools/ubench/valu_rate
  echo "== build timing B"; N=1000000 W=1920 H=1080 timeout 120 python tools/build_timing.py
  echo "== build timing E"; N=6000000 W=3840 H=2160 timeout 200 python tools/build_timing.py
  echo "== sweep B"; timeout 120 python tools/tune_sweep.py --fif 1,3 --batches 3
  echo "== sweep B fast blend"; timeout 120 python tools/tune_sweep.py --fif 1,3 --batches 3 --exp-mode 0 --contract 1
  echo "== sweep C"; timeout 200 python tools/tune_sweep.py --fif 1,3 --batches 2 --frames 100 --gaussians 6000000
  echo "== sweep T"; timeout 200 python tools/tune_sweep.py --fif 1,3 --batches 2 --frames 100 --gaussians 6000000 --scene T
  echo "== sweep E"; timeout 200 python tools/tune_sweep.py --fif 1,3 --batches 2 --frames 60 --gaussians 6000000 --width 3840 --height 2160
  echo "== blend work"; GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_stats.so timeout 300 python tools/blend_stats.py --out gpurun_out/r03_blend_work.json B C T E
} > gpurun_out/r03_trip2.log 2>&1
tail -40 gpurun_out/r03_trip2.log
