#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke(), a 2-rank rehearsal of bench.py on one GPU (gloo)
R=$(pwd); O=$R/gpurun_out; exec < /dev/null
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/r02_tests_final.log
timeout 120 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
GS_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/r02_bench_2rank_rehearsal.json
python tools/bench_line.py $O/r02_bench_2rank_rehearsal.json
