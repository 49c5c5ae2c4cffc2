#!/bin/bash
set -u
exec < /dev/null
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -x -q > gpurun_out/r03_gputests_4.log 2>&1
tail -3 gpurun_out/r03_gputests_4.log | head -1
{
  echo "== build timing B"; N=1000000 W=1920 H=1080 timeout 120 python tools/build_timing.py
  echo "== build timing E"; N=6000000 W=3840 H=2160 timeout 200 python tools/build_timing.py
  echo "== sweep B"; timeout 120 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 3
  echo "== sweep E"; timeout 200 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 2 --frames 60 --gaussians 6000000 --width 3840 --height 2160
  echo "== sweep C"; timeout 200 python tools/tune_sweep.py --no-prime --fif 1,3 --batches 2 --frames 100 --gaussians 6000000
} > gpurun_out/r03_trip4.log 2>&1
grep -v "^#" gpurun_out/r03_trip4.log | tail -40
