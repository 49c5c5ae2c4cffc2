#!/bin/bash
# level-2 kernels after the per-phase refresh of the thread coordinates (spills gone) against the library of before (libgs3d_hip_oldloop.so:
# round 4's level 2 and the compiler's blend loop), frames must stay bit-identical.  B, C, E, T.
exec < /dev/null
for W in "B:--frames 300" "C:--gaussians 6000000 --frames 100" "E:--gaussians 6000000 --width 3840 --height 2160 --frames 60" "T:--gaussians 6000000 --scene T --frames 100"; do
  echo "#### ${W%%:*}"
  AB_ARGS="${W#*:}" bash tools/ab_quick.sh oldloop cur 2>&1 | grep "==\|fif" | awk '{ if ($1=="fif") print $1,$2,$3,$4, $9,$10, "lvl", $18, "spans", $(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF; else print }'
done
