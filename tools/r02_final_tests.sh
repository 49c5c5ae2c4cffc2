#!/bin/bash
R=$(pwd); O=$R/gpurun_out; exec < /dev/null
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/r02_tests_final.log
timeout 120 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
GS_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys,json
b=json.loads(sys.stdin.read()); print('2-rank rehearsal (gloo, one GPU):', b['value'], b['unit'], 'n_gpus', b['n_gpus'], b['config']['workload'])"
