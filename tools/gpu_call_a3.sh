#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
PGH_STRESS_DUMP=$GRAFT_REPO_ROOT/gpurun_out timeout 400 python tests/stress_filters.py --seconds 150 --seed 51 2>&1 | tail -1 | cut -c1-400
