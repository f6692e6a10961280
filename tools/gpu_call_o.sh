#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for round in 1 2; do
timeout 900 python tools/probe_variants.py --scale 23 --seeds 100 --iters 20 tools/variants/libpgh_base.so tools/variants/libpgh_g9.so tools/variants/libpgh_pb128.so tools/variants/libpgh_pb256.so > $O/o_variants_$round.log 2>&1
cat $O/o_variants_$round.log | cut -c1-300
done
