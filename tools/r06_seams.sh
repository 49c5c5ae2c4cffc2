#!/bin/bash
# round 6 (verdict r5 item 5): the seams of a frame rendered alone -- begin/end timestamps of consecutive kernels on one stream
# (rocprofv3 --kernel-trace), graph replay off and on.  Output: gpurun_out/r06_seams/seams_{stream,graph}.txt
R=$(pwd); O=$R/gpurun_out/r06_seams; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in stream graph; do
  G=0; [ $mode = graph ] && G=1
  GS_GRAPH=$G timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -o t -- \
      python $R/tools/tune_sweep.py --fif 1 --frames 300 --batches 2 --no-prime --warm 220 $SEAM_ARGS > $O/run_$mode.txt 2>&1
  python $R/tools/seams_summary.py $O/$mode > $O/seams_$mode.txt
  rm -rf $O/$mode
  cat $O/seams_$mode.txt
done
