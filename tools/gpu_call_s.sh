#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg2" > $O/s_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/s_tests.log
PGH_PB=1 PGH_PB_FORCE=1 timeout 400 python tests/stress_filters.py --seconds 150 --seed 7 > $O/s_stress.log 2>&1; echo "stress(pb forced) rc=$?"; tail -3 $O/s_stress.log
timeout 300 python tools/stress_gpu.py --seconds 100 > $O/s_stress2.log 2>&1; echo "stress_gpu rc=$?"; tail -3 $O/s_stress2.log
