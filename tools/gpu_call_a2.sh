#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for i in 1 2 3; do timeout 300 python tools/probe_default_rule.py 2>&1 | tail -1; done
timeout 400 python tests/stress_filters.py --seconds 150 --seed 51 > $O/a2_stress.log 2>&1; echo "stress rc=$?"; tail -1 $O/a2_stress.log
PGH_PB=1 PGH_PB_FORCE=1 timeout 400 python tests/stress_filters.py --seconds 150 --seed 52 > $O/a2_stress_pb.log 2>&1; echo "stress(pb) rc=$?"; tail -1 $O/a2_stress_pb.log
