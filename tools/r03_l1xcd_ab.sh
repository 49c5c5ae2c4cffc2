#!/bin/bash
# A/B of the XCD-aware level-1 block order (GS_L1_XCD_RUN=32, the default) against launch order (libgs3d_hip_l1flat.so,
# `make variant TAG=l1flat DEFS=-DGS_L1_XCD_RUN=0`), interleaved on one box; every frame must equal the first bit for bit.
#   gpurun --timeout 420 -- 'bash tools/r03_l1xcd_ab.sh'
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
mkdir -p "$O"
LOG=$O/r03_l1xcd_ab.txt
: > "$LOG"
run() {  # name, args
  local name=$1; shift
  rm -f /tmp/ab_ref.npy
  for rep in 1 2; do for v in cur l1flat; do
    L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
    echo "== $name $v (run $rep)" >> "$LOG"
    GS3D_HIP_LIB=$L timeout 100 python tools/tune_sweep.py --quick --ref-image /tmp/ab_ref.npy "$@" 2>&1 | grep -v "^$" | tail -5 >> "$LOG"
  done; done
}
run B --gaussians 1000000 --width 1920 --height 1080 --scene S --frames 300
run C --gaussians 6000000 --width 1920 --height 1080 --scene S --frames 100
run T --gaussians 6000000 --width 1920 --height 1080 --scene T --frames 100
run E --gaussians 6000000 --width 3840 --height 2160 --scene S --frames 60
cat "$LOG"
timeout 240 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
