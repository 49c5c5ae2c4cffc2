#!/bin/bash
# rocprofv3 --kernel-trace --stats around the DRIVER'S OWN command (python bench.py --gpus 1 --steps 20 --warmup 5), so that the kernel
# durations behind the bench line can be read next to it (the per-workload profiles of tools/profile_lite.sh use tools/tune_sweep.py,
# the same C-ABI calls without the torch import).  One gpurun call:  gpurun --timeout 600 -- 'bash tools/r06_rocprof_bench.sh'
set -u; exec < /dev/null
R=$(pwd); O=$R/gpurun_out/r06_rocprof_bench; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/raw" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench.err"
cd "$R"
S=$(find "$O/raw" -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5   (one process: the timed region with three frames in flight,"
  echo "# the one-frame-in-flight diagnostic, the parity frames, the other blend modes and other_configs all launch the same kernel names)"
  echo "# columns: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev"
  head -40 "$S"; } > "$O/r06_rocprof_bench_driver_command.txt"
find "$O/raw" -name '*_kernel_trace.csv' -delete
tail -1 "$O/bench_line.json" | cut -c1-400
head -12 "$O/r06_rocprof_bench_driver_command.txt" | cut -c1-200
