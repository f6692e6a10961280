#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # name world backend native seconds seed extra-env...
  name=$1; world=$2; backend=$3; native=$4; secs=$5; seed=$6; shift 6
  env PGH_TEST_ENGINE=hip PGH_DIST_BACKEND=$backend PGH_DIST_NATIVE=$native "$@" timeout $((secs + 400)) python -m torch.distributed.run --nnodes=1 --nproc-per-node=$world --master-addr 127.0.0.1 --master-port $((29800 + world)) tests/stress_partitioned.py --seconds $secs --seed $seed --max-scale 16 > $O/q_$name.log 2>&1
  echo "$name rc=$?"; grep -E "stress ok|FAILED|AssertionError" $O/q_$name.log | cut -c1-900 | head -3
}
run rccl_x1 1 nccl auto 120 14 PGH_DIST_GATHER_ALONE=1 PGH_DIST_REDUCE_ALONE=1
run rccl_x1_three_queues 1 nccl auto 120 15 PGH_DIST_GATHER_ALONE=1 PGH_DIST_REDUCE_ALONE=1 PGH_DIST_SINGLE_STREAM=0
run rccl_x1_plain 1 nccl auto 60 17
