#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/gpu_tests.log
L=pygrank_amd/csrc/libpgh_hip.so
for cfg in "PGH_FORMAT=csr" "PGH_BLOCKS=1 PGH_RELABEL=0 PGH_VALUES=1" "PGH_BLOCKS=1 PGH_RELABEL=1 PGH_VALUES=1" "PGH_BLOCKS=4 PGH_RELABEL=1 PGH_VALUES=1" "PGH_BLOCKS=4 PGH_RELABEL=1" "PGH_BLOCKS=8 PGH_RELABEL=1" "PGH_BLOCKS=2 PGH_RELABEL=1" "PGH_BLOCKS=4 PGH_RELABEL=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/probe_variants.py --scale 23 $L 2>&1 | tail -2
done > $O/bsf_variants23.log 2>&1
cat $O/bsf_variants23.log
