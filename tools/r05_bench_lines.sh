#!/bin/bash
# the bench lines of the library as shipped (the counter files they quote are the committed profiles/r05_pmc_hbm_traffic.json: same kernel sources)
set -u; exec < /dev/null
R=$(pwd); O=$R/gpurun_out/r05_final; mkdir -p "$O"; export TMPDIR=/tmp
B="timeout 240 python bench.py"
$B --steps 20 --warmup 5 > "$O/r05_bench_driver_command.json" 2> /dev/null
$B --steps 200 --warmup 20 > "$O/r05_bench_default.json" 2> "$O/bench_default.err"
$B --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --exact > "$O/r05_bench_default_exact.json" 2> /dev/null
$B --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --bgra8-only > "$O/r05_bench_default_bgra8.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 > "$O/r05_bench_configC_standin.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 --scene T > "$O/r05_bench_configC_T.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 --width 3840 --height 2160 > "$O/r05_bench_configE.json" 2> /dev/null
python tools/bench_line.py "$O"/r05_bench_*.json
