#!/bin/bash
# A/B of a level-1 variant build against the default library, interleaved on one box; every frame must equal the first bit for bit.
#   l1flat   launch-order blocks instead of runs of 32 per XCD   (make variant TAG=l1flat DEFS=-DGS_L1_XCD_RUN=0)
#   nodense  level 1 walks the N-wide planes instead of the dense list of visible Gaussians   (make variant TAG=nodense DEFS=-DGS_L1_DENSE=0)
#   x64, x128  runs of 64 / 128 blocks per XCD   (make variant TAG=x64 DEFS=-DGS_L1_XCD_RUN=64)
#   gpurun --timeout 420 -- 'bash tools/r03_l1_ab.sh nodense "1 2"'        (variants, repetitions, workloads; NO_TESTS=1 skips the parity file)
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
mkdir -p "$O"
OTHER=${1:-l1flat}; REPS=${2:-"1 2"}; WL=${3:-"B C T E"}
LOG=$O/r03_l1_ab_${OTHER// /_}.txt
: > "$LOG"
run() {  # name, args
  local name=$1; shift
  rm -f /tmp/ab_ref.npy
  for rep in $REPS; do for v in cur $OTHER; do
    L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
    echo "== $name $v (run $rep)" >> "$LOG"
    GS3D_HIP_LIB=$L timeout 100 python tools/tune_sweep.py --quick --ref-image /tmp/ab_ref.npy "$@" 2>&1 | grep -v "^$" | tail -5 >> "$LOG"
  done; done
}
for W in $WL; do case $W in
  B) run B --gaussians 1000000 --width 1920 --height 1080 --scene S --frames 300;;
  C) run C --gaussians 6000000 --width 1920 --height 1080 --scene S --frames 100;;
  T) run T --gaussians 6000000 --width 1920 --height 1080 --scene T --frames 100;;
  E) run E --gaussians 6000000 --width 3840 --height 2160 --scene S --frames 60;;
esac; done
sed -E "s/\(min.*bit-equal/bit-equal/; s/ V [0-9]+ E1.*spans us/ spans/" "$LOG"
[ -n "${NO_TESTS:-}" ] || timeout 240 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
