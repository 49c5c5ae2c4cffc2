#!/bin/bash
R=$(pwd); O=$R/gpurun_out; exec < /dev/null
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "config_a or every_bin_size or golden or ragged or degenerate or refined or sort_paths" 2>&1 | tail -3
GS_BIN_SHIFT=2 timeout 100 python tools/build_timing.py 2>&1 | grep -v amdgpu.ids | tail -4
for sh in 2 3; do
  echo "== shift $sh"; timeout 150 bash tools/prof_quick.sh z_s$sh GS_BIN_SHIFT=$sh 2>&1 | grep -v amdgpu.ids | grep "gs::" | head -6
done
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], b["value"], "fps; one-in-flight", b["frames_per_s_one_in_flight"], "hwexp", b.get("frames_per_s_hw_exp"), "serial", b["passes_serial_ms"], "lvl", b["config"].get("sort_level"), "bin", b["config"].get("bin_tiles"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 40 > $O/ab5_$tag.json 2>/dev/null; line $O/ab5_$tag.json $tag; }
run s2 GS_BIN_SHIFT=2
run s3 GS_BIN_SHIFT=3
run s2b GS_BIN_SHIFT=2
run s3b GS_BIN_SHIFT=3
