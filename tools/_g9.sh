cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04_gpu_suite.log 2>&1
tail -15 gpurun_out/r04_gpu_suite.log
