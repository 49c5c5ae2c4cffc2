#!/bin/bash
# the once-per-run ~40 ms stall at config B.  Run on the GPU box: gpurun -- bash tools/r05_stall.sh
set -u
R=$PWD
O=$R/gpurun_out/stall
mkdir -p $O
export GS_DEBUG_STALLS=2
probe() {  # name, command...
    name=$1; shift
    ( "$@" ) > $O/$name.json 2> $O/$name.err
    echo "== $name: $(cut -c1-1500 $O/$name.json)"; grep -c stall $O/$name.err | sed 's/^/   stall lines: /'; grep stall $O/$name.err | head -5
}
P="python tools/stall_probe.py"
probe baseline       $P --frames 8000 --label baseline
probe timing0        $P --frames 8000 --timing 0 --label timing0
probe fif1           $P --frames 6000 --fif 1 --label fif1
probe graph          $P --frames 8000 --graph --label graph
probe torch          $P --frames 8000 --torch --label torch
probe sync20         $P --frames 8000 --sync-every 20 --label sync20
probe batch100       env DEBUG_CLR_MAX_BATCH_SIZE=100 $P --frames 8000 --label batch100
probe cpusync        env DEBUG_CLR_BATCH_CPU_SYNC_SIZE=64 $P --frames 8000 --label cpusync
probe devkernarg     env HIP_FORCE_DEV_KERNARG=1 $P --frames 8000 --label devkernarg
probe small10k       $P --frames 8000 --n 10000 --width 256 --height 256 --label small10k
# what the runtime logs around the stall
AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/amdlog $P --frames 4000 --label amdlog > $O/amdlog.json 2> $O/amdlog.err
ls -la /tmp/amdlog* | head
for f in /tmp/amdlog*; do python tools/stall_trace_summary.py amdlog $f > $O/amdlog_summary.txt 2>&1; done
echo "== amdlog: $(cut -c1-700 $O/amdlog.json)"; head -c 8000 $O/amdlog_summary.txt
# API timeline
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --hsa-trace --kernel-trace --output-format csv -d /tmp/stalltrace -o t -- python $R/tools/stall_probe.py --frames 4000 --label rocprof > $O/rocprof.json 2> $O/rocprof.err
cd $R
python tools/stall_trace_summary.py rocprof /tmp/stalltrace > $O/rocprof_summary.txt 2>&1
echo "== rocprof: $(cut -c1-700 $O/rocprof.json)"; head -c 8000 $O/rocprof_summary.txt
