#!/bin/bash
# Round 6's committed evidence in one gpurun call: kernel stats + PMC counters of the shipped library (tools/profile_lite.sh), their
# summaries written into this copy's profiles/ (so the bench lines below quote counters whose source hash matches the library), then the
# bench lines.  Everything to keep lands under gpurun_out/r06_final/ (profiles/ itself does not travel back):
#   gpurun --timeout 900 -- 'bash tools/r06_final.sh'   then here:   cp gpurun_out/r06_final/r06_* profiles/
set -u
exec < /dev/null
R=$(pwd); O=$R/gpurun_out/r06_final; mkdir -p "$O"
bash tools/profile_lite.sh r06 B C T E > "$O/profile_lite.log" 2>&1
python tools/profile_summary.py r06 > "$O/profile_summary.log" 2>&1
export TMPDIR=/tmp
B="timeout 240 python bench.py"
$B --steps 20 --warmup 5 > "$O/r06_bench_driver_command.json" 2> /dev/null
$B --steps 200 --warmup 20 > "$O/r06_bench_default.json" 2> "$O/bench_default.err"
$B --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --exact > "$O/r06_bench_default_exact.json" 2> /dev/null
$B --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --bgra8-only > "$O/r06_bench_default_bgra8.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 > "$O/r06_bench_configC_standin.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 --scene T > "$O/r06_bench_configC_T.json" 2> /dev/null
$B --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --gaussians 6000000 --width 3840 --height 2160 > "$O/r06_bench_configE.json" 2> /dev/null
cp profiles/r06_kernel_stats_* profiles/r06_pmc_* "$O/" 2> /dev/null
python tools/bench_line.py "$O"/r06_bench_*.json
# the blend's work counters (instrumented build) and the shader clock k_blend runs at (clock build), per workload
GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_stats.so timeout 600 python tools/blend_stats.py --out "$O/r06_blend_work.json" B C T E > "$O/blend_stats.log" 2>&1
GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_clk.so timeout 300 python tools/blend_clock.py B C T E > "$O/r06_blend_clock.txt" 2>&1
