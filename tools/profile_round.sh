#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh rNN'
# then, back in the container:  python tools/profile_summary.py rNN   (writes profiles/rNN_*).
# Kernel-trace/stats runs and the PMC runs are separate commands (counters never share a run with a trace domain
# other than the kernel trace; FETCH_SIZE and WRITE_SIZE each get their own pass).
set -u
exec < /dev/null
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"/{default,serial,configE,pmc,fetch,write}
python -c "import __graft_entry__ as e; print(e.load_package().binding.library_source_hash())" > "$OUT/source_hash.txt"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"

timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/default" -o d -- $B --steps 200 --warmup 20 \
    > "$OUT/default/bench.json" 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/serial" -o s -- $B --steps 200 --warmup 20 \
    --frames-in-flight 1 > "$OUT/serial/bench.json" 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/configE" -o e -- $B --steps 50 --warmup 5 \
    --frames-in-flight 1 --gaussians 6000000 --width 3840 --height 2160 > "$OUT/configE/bench.json" 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc" -o p -- $B --steps 3 --warmup 1 \
    --frames-in-flight 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- $B --steps 3 --warmup 1 \
    --frames-in-flight 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- $B --steps 3 --warmup 1 \
    --frames-in-flight 1 > /dev/null 2>&1
rm -f "$OUT"/*/*_kernel_trace.csv  # per-dispatch traces are large; the stats and counter CSVs are what is summarised

cd "$R"
# the un-profiled lines the summaries are quoted beside
python bench.py --steps 200 --warmup 20 > "$OUT/bench_default.json" 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gaussians 6000000 --width 3840 --height 2160 \
    > "$OUT/bench_configE.json" 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gaussians 6000000 > "$OUT/bench_configC_standin.json" 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --hw-exp > "$OUT/bench_default_hwexp.json" 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gaussians 6000000 --width 3840 --height 2160 --sh16 \
    > "$OUT/bench_configE_sh16.json" 2>/dev/null
ls -R "$OUT" | head -40
tail -c 400 "$OUT/bench_default.json"
