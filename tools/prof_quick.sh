#!/bin/bash
# quick per-kernel durations (one frame in flight): gpurun_out/pq_<tag>/ ; usage: bash tools/prof_quick.sh tag [env...]
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/pq_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --frames-in-flight 1 > $O/bench.json 2>/dev/null
rm -f $O/*_kernel_trace.csv
cd $R
python - $O <<'PY'
import csv,glob,sys,re
f=glob.glob(sys.argv[1]+'/*_kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
for r in rows:
    n=re.sub(r'^void ','',r['Name']).split('(')[0]
    print('%-40s calls %6d avg %9.2f us  pct %5.1f' % (n[:40], int(r['Calls']), float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
