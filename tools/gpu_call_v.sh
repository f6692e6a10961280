#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 64 32 16 8; do timeout 600 python tools/probe_batch.py --scale 23 --batch $b 2>&1 | grep "^batched" | cut -c1-300; done
PGH_PB=1 PGH_PB_FORCE=1 timeout 700 python tests/stress_filters.py --seconds 300 --seed 22 > $O/v_filters_pb.log 2>&1; echo "filters(pb) rc=$?"; tail -2 $O/v_filters_pb.log | cut -c1-300
env PGH_TEST_ENGINE=hip PGH_DIST_BACKEND=gloo PGH_DIST_NATIVE=external timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29802 tests/stress_partitioned.py --seconds 400 --seed 24 --max-scale 17 > $O/v_part2.log 2>&1; echo "part x2 rc=$?"; grep -E "stress ok|FAILED" $O/v_part2.log | head -3 | cut -c1-600
