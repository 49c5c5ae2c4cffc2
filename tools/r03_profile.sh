#!/bin/bash
# round-3 evidence: rocprofv3 stats + counters for B, C, T, E; bench lines; blend work counts
set -u
exec < /dev/null
TAG=${1:-r03}
bash tools/profile_lite.sh $TAG B C T E > gpurun_out/profile_lite_$TAG.log 2>&1
O=gpurun_out/prof_$TAG
GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_stats.so timeout 300 python tools/blend_stats.py --out $O/blend_work.json B C T E > $O/blend_work.log 2>&1
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --fast-blend > $O/bench_default_fast_blend.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --bgra8-only > $O/bench_default_bgra8.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gaussians 6000000 > $O/bench_configC_standin.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gaussians 6000000 --scene T > $O/bench_configC_T.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gaussians 6000000 --width 3840 --height 2160 > $O/bench_configE.json 2>/dev/null
tail -c 600 $O/bench_default.json; echo; tail -3 $O/bench_default.err
for f in $O/bench_*.json; do python tools/bench_line.py $f 2>/dev/null | head -2; done
