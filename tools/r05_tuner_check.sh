#!/bin/bash
# what the blend tuner decides, three frames in flight from the first frame on (the bench's situation), four fresh processes per workload
exec < /dev/null
for W in "B:--frames 200" "E:--gaussians 6000000 --width 3840 --height 2160 --frames 60" "T:--gaussians 6000000 --scene T --frames 100" "C:--gaussians 6000000 --frames 100"; do
  for rep in 1 2 3 4; do
    timeout 120 python tools/tune_sweep.py --batches 2 --fif 3 ${W#*:} 2>&1 | grep fif | tail -1 | awk -v w=${W%%:*} '{ print w, $1,$2,$3,$4, $11,$12 }'
  done
done
