cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -3
( time python bench.py ) > gpurun_out/r04_bench_a.log 2>&1
tail -c 600 gpurun_out/r04_bench_a.log
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5
