#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 400 python tests/stress_filters.py --seconds 240 --seed 3 > $O/n_stress.log 2>&1; echo "stress rc=$?"; tail -5 $O/n_stress.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/n_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/n_tests.log
