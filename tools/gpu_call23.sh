#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
V=tools/variants
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
PGH_BLOCKS=4 timeout 600 python tools/probe_variants.py --scale 23 $V/libpgh_base.so $V/libpgh_base.so $V/libpgh_notr.so $V/libpgh_nont.so $V/libpgh_g8s7.so $V/libpgh_g8s7nont.so $V/libpgh_g6.so 2>&1 | cut -c1-40,70-200 | tee gpurun_out/bsf_v7_transpose_scale23.log
