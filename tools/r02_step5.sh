#!/bin/bash
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "config_a or every_bin_size or golden or ragged or one_dense or sort_paths or capacity or degenerate or small_counts or needle or candidate_overflow" 2>&1 | tail -12
for sh in 2 3; do GS_BIN_SHIFT=$sh timeout 120 python tools/build_timing.py 2>&1 | grep -v amdgpu.ids; done
for sh in 2 3; do
  echo "== shift $sh"; timeout 150 bash tools/prof_quick.sh x_s$sh GS_BIN_SHIFT=$sh 2>&1 | grep -v amdgpu.ids | grep "gs::"
done
for sh in 2 3; do GS_BIN_SHIFT=$sh timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r02_bench3_s$sh.json 2>/dev/null; python - $O/r02_bench3_s$sh.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], b["value"], "fps; one-in-flight", b["frames_per_s_one_in_flight"], b["passes_serial_ms"])
PY
done
