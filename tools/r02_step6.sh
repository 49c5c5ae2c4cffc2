#!/bin/bash
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_golden.py -m gpu -x -q -k "graph_replay or fp16 or 4_gib or dist or broadcast or pose or config_a or every_bin_size or golden or degenerate or sort_paths" 2>&1 | tail -12
GS_BIN_SHIFT=2 timeout 120 python tools/build_timing.py 2>&1 | grep -v amdgpu.ids
for sh in 2 3; do
  echo "== shift $sh"; timeout 150 bash tools/prof_quick.sh y_s$sh GS_BIN_SHIFT=$sh 2>&1 | grep -v amdgpu.ids | grep "gs::"
done
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], b["value"], "fps; one-in-flight", b["frames_per_s_one_in_flight"], "serial", b["passes_serial_ms"], "spread", b["timed"]["spread"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; env "$@" timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/ab3_$tag.json 2>/dev/null; line $O/ab3_$tag.json $tag; }
run shift2 GS_BIN_SHIFT=2
run shift3 GS_BIN_SHIFT=3
run shift3_hwexp GS_BIN_SHIFT=3 GS_EXP_MODE=1
run shift3_graph GS_BIN_SHIFT=3 GS_GRAPH=1
Z56=00000000000000000000000000000000000000000000000000000000
F56=ffffffffffffffffffffffffffffffffffffffffffffffffffffffff
M01=0101010101010101010101010101010101010101010101010101010101010101
MFE=fefefefefefefefefefefefefefefefefefefefefefefefefefefefefefefefe
ALL=${F56}ffffffff
run cu_all_all GS_BIN_SHIFT=3 GS_CU_MASK_PREP=$ALL GS_CU_MASK_BLEND=$ALL
run cu_interleave_32 GS_BIN_SHIFT=3 GS_CU_MASK_PREP=$M01 GS_CU_MASK_BLEND=$MFE
run cu_low_32 GS_BIN_SHIFT=3 GS_CU_MASK_PREP=${Z56}ffffffff GS_CU_MASK_BLEND=${F56}00000000
run cu_all_fe GS_BIN_SHIFT=3 GS_CU_MASK_PREP=$ALL GS_CU_MASK_BLEND=$MFE
