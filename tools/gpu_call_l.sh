#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python tools/probe_partition.py --seeds --worlds 2 4 8 > $O/l_slices.log 2>&1; cat $O/l_slices.log | cut -c1-400
