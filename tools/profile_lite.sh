#!/bin/bash
# The rocprofv3 evidence of a round in about a minute of GPU time (tools/profile_round.sh is the long form: it also runs
# bench.py for configs C / E).  The profiled process is tools/tune_sweep.py -- the same C-ABI calls as bench.py's timed
# region, without the torch import.  Kernel-trace/stats runs and the PMC runs are separate commands (counters never share a
# run with a trace domain other than the kernel trace; FETCH_SIZE and WRITE_SIZE each get their own pass).
#   gpurun --timeout 200 -- 'bash tools/profile_lite.sh r02b'   then here:   python tools/profile_summary.py r02b
set -u
exec < /dev/null
TAG=${1:-r02b}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"/{default,serial,pmc,fetch,write}
python -c "import __graft_entry__ as e; print(e.load_package().binding.library_source_hash())" > "$OUT/source_hash.txt"
cd /tmp && export TMPDIR=/tmp
D="python $R/tools/tune_sweep.py --no-prime --batches 1"
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/serial" -o s -- $D --fif 1 --frames 300 --json-out "$OUT/serial/bench.json" > /dev/null 2>&1
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/default" -o d -- $D --fif 3 --frames 300 --json-out "$OUT/default/bench.json" > /dev/null 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc" -o p -- $D --fif 1 --frames 3 --warm 1 > /dev/null 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- $D --fif 1 --frames 3 --warm 1 > /dev/null 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- $D --fif 1 --frames 3 --warm 1 > /dev/null 2>&1
rm -f "$OUT"/*/*_kernel_trace.csv
cd "$R"
timeout 60 python tools/tune_sweep.py --fif 1,3 --json-out "$OUT/sweep_default.json" | tail -3
ls -R "$OUT" | head -30
