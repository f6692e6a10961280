#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
timeout 400 python tests/stress_filters.py --seconds 200 --seed 51 2>&1 | tail -1 | cut -c1-400
bash tools/gpu_collect_r04.sh
