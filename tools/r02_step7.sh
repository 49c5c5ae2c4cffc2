#!/bin/bash
R=$(pwd); O=$R/gpurun_out; exec < /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_viewer.py -m gpu -x -q -k "refined or viewer or embedded or scripted or metrics or many_path" 2>&1 | tail -6
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], b["value"], "fps; one-in-flight", b["frames_per_s_one_in_flight"], "hwexp", b.get("frames_per_s_hw_exp"), "serial", b["passes_serial_ms"], "lvl", b["config"].get("sort_level"), "bin", b["config"].get("bin_tiles"), "path", b["config"]["depth_order_path"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 200 python bench.py --no-cpu-baseline "$@" > $O/ab4_$tag.json 2>/dev/null; line $O/ab4_$tag.json $tag; }
run fif2 --steps 200 --warmup 20 --frames-in-flight 2
run fif3 --steps 200 --warmup 20 --frames-in-flight 3
run fif4 --steps 200 --warmup 20 --frames-in-flight 4
run fif6 --steps 200 --warmup 20 --frames-in-flight 6
run configC --steps 100 --warmup 40 --gaussians 6000000
run configE --steps 100 --warmup 40 --gaussians 6000000 --width 3840 --height 2160
run configE_sh16 --steps 100 --warmup 40 --gaussians 6000000 --width 3840 --height 2160 --sh16
