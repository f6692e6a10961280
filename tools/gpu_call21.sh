#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for B in 4 8; do
PGH_BLOCKS=$B timeout 300 python tools/probe_variants.py --scale 23 tools/variants/libpgh_base.so tools/variants/libpgh_g6.so 2>&1 | sed "s/^/B=$B /"
done | tee gpurun_out/bsf_v7_probe_scale23.log
