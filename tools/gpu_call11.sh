#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/gpu_tests.log
V=tools/variants
for cfg in "PGH_BLOCKS=4" "PGH_BLOCKS=8"; do
  for lib in base g3 g7 hot31k ipt12 ipt16; do
    echo -n "$cfg "; env $cfg timeout 300 python tools/probe_variants.py --scale 23 $V/libpgh_$lib.so 2>&1 | tail -1
  done
done > $O/bsf8_probe23.log 2>&1
cat $O/bsf8_probe23.log
