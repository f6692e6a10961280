#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/y_bench.json 2> gpurun_out/y_bench.err; echo "bench rc=$?"
