#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/gpu_tests.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29700 bench.py --gpus 1 --force-partitioned --steps 5 --warmup 1 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist bench rc=$?"; cat $O/bench_dist1.json; tail -5 $O/bench_dist1.err
