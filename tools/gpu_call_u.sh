#!/bin/bash
# long randomised runs of the three harnesses on the final sources
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tests/stress_filters.py --seconds 500 --seed 21 > $O/u_filters.log 2>&1; echo "filters rc=$?"; tail -2 $O/u_filters.log
PGH_PB=1 PGH_PB_FORCE=1 timeout 700 python tests/stress_filters.py --seconds 300 --seed 22 > $O/u_filters_pb.log 2>&1; echo "filters(pb) rc=$?"; tail -2 $O/u_filters_pb.log
timeout 600 python tools/stress_gpu.py --seconds 300 --seed 23 > $O/u_gpu.log 2>&1; echo "stress_gpu rc=$?"; tail -2 $O/u_gpu.log
env PGH_TEST_ENGINE=hip PGH_DIST_BACKEND=gloo PGH_DIST_NATIVE=external timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29802 tests/stress_partitioned.py --seconds 400 --seed 24 --max-scale 17 > $O/u_part2.log 2>&1; echo "part x2 rc=$?"; grep -E "stress ok|FAILED" $O/u_part2.log | head -3 | cut -c1-600
env PGH_TEST_ENGINE=hip PGH_DIST_BACKEND=gloo PGH_DIST_NATIVE=external timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29804 tests/stress_partitioned.py --seconds 300 --seed 25 --max-scale 15 > $O/u_part4.log 2>&1; echo "part x4 rc=$?"; grep -E "stress ok|FAILED" $O/u_part4.log | head -3 | cut -c1-600
