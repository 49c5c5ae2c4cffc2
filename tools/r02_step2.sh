#!/bin/bash
# one gpurun call: parity of the new binning, per-kernel profile, bench A/B (bin size, CU masks).  Everything under timeout.
R=$(pwd); O=$R/gpurun_out; exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -15 > $O/r02_tests4.log; tail -4 $O/r02_tests4.log
timeout 200 bash tools/prof_quick.sh s2c 2>&1 | grep -v amdgpu.ids
timeout 200 bash tools/prof_quick.sh s3c GS_BIN_SHIFT=3 2>&1 | grep -v amdgpu.ids
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], b["value"], "fps; one-in-flight", b["frames_per_s_one_in_flight"], "serial", b["passes_serial_ms"], "spread", b["timed"]["spread"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; env "$@" timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/ab2_$tag.json 2>/dev/null; line $O/ab2_$tag.json $tag; }
run default
run shift3 GS_BIN_SHIFT=3
run hwexp GS_EXP_MODE=1
Z56=00000000000000000000000000000000000000000000000000000000
F56=ffffffffffffffffffffffffffffffffffffffffffffffffffffffff
M01=0101010101010101010101010101010101010101010101010101010101010101
MFE=fefefefefefefefefefefefefefefefefefefefefefefefefefefefefefefefe
M03=0303030303030303030303030303030303030303030303030303030303030303
MFC=fcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfcfc
ALL=${F56}ffffffff
run cu_interleave_32 GS_CU_MASK_PREP=$M01 GS_CU_MASK_BLEND=$MFE
run cu_low_32 GS_CU_MASK_PREP=${Z56}ffffffff GS_CU_MASK_BLEND=${F56}00000000
run cu_interleave_64 GS_CU_MASK_PREP=$M03 GS_CU_MASK_BLEND=$MFC
run cu_all_all GS_CU_MASK_PREP=$ALL GS_CU_MASK_BLEND=$ALL
run cu_all_fe GS_CU_MASK_PREP=$ALL GS_CU_MASK_BLEND=$MFE
