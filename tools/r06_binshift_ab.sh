#!/bin/bash
# round 6: config B with 8x8-tile bins (k_bin_fast<8>, 135 workgroups) against 4x4-tile bins (k_bin_fast<4>, 510): frames/s with 1 and 3
# frames in flight (tools/tune_sweep.py --quick, frames bit-identical) and the serial per-kernel durations (rocprofv3 --kernel-trace --stats)
R=$(pwd); O=$R/gpurun_out/r06_binshift; mkdir -p $O; exec < /dev/null
rm -f /tmp/ab_ref.npy
for s in 3 2 3 2; do
  echo "== GS_BIN_SHIFT=$s"
  GS_BIN_SHIFT=$s timeout 120 python tools/tune_sweep.py --quick --frames 300 --batches 3 --ref-image /tmp/ab_ref.npy 2>&1 | tail -5
done | tee $O/ab.txt
for s in 3 2; do
  echo "== serial kernels, GS_BIN_SHIFT=$s"
  bash tools/prof_quick.sh binshift$s GS_BIN_SHIFT=$s
done | tee $O/kernels.txt
