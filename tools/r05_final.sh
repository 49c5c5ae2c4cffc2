#!/bin/bash
# Round 5's committed evidence in one gpurun call: kernel stats + PMC counters of the shipped library (tools/profile_lite.sh), their
# summaries written into this copy's profiles/ (so the bench lines below quote counters whose source hash matches the library), then the
# bench lines.  Everything to keep lands under gpurun_out/r05_final/ (profiles/ itself does not travel back):
#   gpurun --timeout 900 -- 'bash tools/r05_final.sh'   then here:   cp gpurun_out/r05_final/r05_* profiles/
set -u
exec < /dev/null
R=$(pwd); O=$R/gpurun_out/r05_final; mkdir -p "$O"
bash tools/profile_lite.sh r05 B C T E > "$O/profile_lite.log" 2>&1
python tools/profile_summary.py r05 > "$O/profile_summary.log" 2>&1
export TMPDIR=/tmp
B="timeout 240 python bench.py"
$B --steps 20 --warmup 5 > "$O/r05_bench_driver_command.json" 2> /dev/null
$B --steps 200 --warmup 20 > "$O/r05_bench_default.json" 2> "$O/bench_default.err"
$B --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --exact > "$O/r05_bench_default_exact.json" 2> /dev/null
$B --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --bgra8-only > "$O/r05_bench_default_bgra8.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 > "$O/r05_bench_configC_standin.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 --scene T > "$O/r05_bench_configC_T.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 --width 3840 --height 2160 > "$O/r05_bench_configE.json" 2> /dev/null
cp profiles/r05_kernel_stats_* profiles/r05_pmc_* "$O/" 2> /dev/null
python tools/bench_line.py "$O"/r05_bench_*.json
