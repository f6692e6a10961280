#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
V=tools/variants
export PGH_DEBUG=1
for cfg in "PGH_BLOCKS=4" "PGH_BLOCKS=8" "PGH_BLOCKS=2" "PGH_BLOCKS=1"; do
  for lib in base g7 g8 g9; do
    echo -n "$cfg "; env $cfg timeout 300 python tools/probe_variants.py --scale 23 $V/libpgh_$lib.so 2>&1 | tail -2
  done
done > $O/bsf7_probe23.log 2>&1
cat $O/bsf7_probe23.log
