#!/bin/bash
# the blend's lockstep, measured by the renderer itself: what it picks per workload and what the frame rates are (cur = automatic) against pinned off / on
exec < /dev/null
for W in "B:--frames 300" "C:--gaussians 6000000 --frames 100" "E:--gaussians 6000000 --width 3840 --height 2160 --frames 60" "T:--gaussians 6000000 --scene T --frames 100" "T1:--gaussians 1000000 --scene T --frames 200"; do
  echo "#### ${W%%:*}"
  for m in auto 0 1; do
    if [ $m = auto ]; then unset GS_BLEND_LOCKSTEP; else export GS_BLEND_LOCKSTEP=$m; fi
    echo "== lockstep $m"
    timeout 120 python tools/tune_sweep.py --quick --batches 3 ${W#*:} 2>&1 | grep fif | awk '{ print $1,$2,$3,$4, $9,$10,$11,$12, "spans", $(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF }'
  done
done
