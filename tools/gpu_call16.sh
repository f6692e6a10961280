#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/gpu_tests.log
timeout 600 python tools/probe_batch.py --scale 23 --batch 64 > $O/batch23.log 2>&1; cat $O/batch23.log | tail -12
