#!/bin/bash
# The exact mode's pair loop, hand-written (blend_pair_loop_exact) against the compiler's form: frames bit-identical (the first process
# saves its frame, the others must equal it), serial and three-in-flight rates, serial k_blend time.  One gpurun call:
#   xc = -DGS_BLEND_ASM_LOOP_EXACT=0 (the compiler's loop), cur = the default library, xnb = -DGS_BLEND_EXECZ_BRANCH_EXACT=0
R=$PWD; O=$R/gpurun_out/exact5; mkdir -p $O; exec < /dev/null
P=$R/3dgs.cpp_amd
run() {  # tag lib scene-args frames
  for v in xc cur xnb; do
    L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
    echo "-- $1 $v"
    GS3D_HIP_LIB=$L timeout 300 python tools/tune_sweep.py --quick --exp-mode 2 --batches 3 --frames $3 $2 --ref-image /tmp/ref_$1.npy 2>&1 | grep -E "fif|differs|identical|Error|error" 
  done
}
run B "--gaussians 1000000 --scene S" 300
run T "--gaussians 6000000 --scene T" 80
run E "--gaussians 6000000 --scene S --width 3840 --height 2160" 60
# the default (guarded) mode must not have moved
echo "-- B default mode"; timeout 200 python tools/tune_sweep.py --quick --batches 3 --frames 300 2>&1 | grep fif
echo "== tests"
GS_EXP_MODE=2 timeout 500 python -m pytest tests/test_gpu_blend_modes.py tests/test_gpu_parity.py -m gpu -q -x -k "blend or config_a or ties or ragged or non_finite or needles or overflow" 2>&1 | tail -4
