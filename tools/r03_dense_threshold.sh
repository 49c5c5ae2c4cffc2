#!/bin/bash
# Where the dense lists of visible Gaussians start to pay: GS_L1_DENSE_MIN=0 (lists) against a huge value (planes), same library,
# scenes S(n) at 1920x1080 for the given n.   gpurun --timeout 150 -- 'bash tools/r03_dense_threshold.sh 2000000 3500000'
R=$(pwd); O=$R/gpurun_out; exec < /dev/null
mkdir -p "$O"; LOG=$O/r03_dense_threshold.txt; : > "$LOG"
for n in "$@"; do
  rm -f /tmp/ab_ref.npy
  for dm in 0 1000000000; do
    echo "== S($n) GS_L1_DENSE_MIN=$dm" >> "$LOG"
    GS_L1_DENSE_MIN=$dm timeout 60 python tools/tune_sweep.py --quick --ref-image /tmp/ab_ref.npy --gaussians $n --width 1920 --height 1080 --frames 150 2>&1 | grep -v "^$" | tail -4 >> "$LOG"
  done
done
sed -E "s/\(min.*bit-equal/bit-equal/; s/ V [0-9]+ E1.*spans us/ spans/" "$LOG"
