#!/bin/bash
# level 4 as one launch over a queue (k_bin_queue) against k_bin_slabs + k_slab_work: GS_L2_QUEUE=0/1, same library, T(6e6)
R=$PWD; O=$R/gpurun_out; exec < /dev/null
rm -f /tmp/ab_ref_T.npy
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dense_bin or depth_slabs or level_changes or regrows or overflow" 2>&1 | tail -3
for rep in 1 2; do for q in 0 1; do
  echo "== GS_L2_QUEUE=$q"
  GS_L2_QUEUE=$q timeout 120 python tools/tune_sweep.py --quick --frames 100 --batches 3 --gaussians 6000000 --scene T --ref-image /tmp/ab_ref_T.npy 2>&1 | grep fif
done; done | tee $O/queue_ab_T.txt
