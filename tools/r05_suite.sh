#!/bin/bash
# the whole GPU suite + the viewer's rate, one gpurun call
O=$PWD/gpurun_out; mkdir -p $O; exec < /dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "pytest rc $?"; tail -6 $O/gpu_suite.txt
python tools/viewer_rate.py > $O/viewer_rate.txt 2>&1; cat $O/viewer_rate.txt | tail -6
