#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
V=tools/variants
PGH_BLOCKS=4 timeout 600 python tools/probe_variants.py --scale 23 $V/libpgh_g6.so $V/libpgh_g6.so $V/libpgh_g6s7.so $V/libpgh_g8.so $V/libpgh_g8s1.so $V/libpgh_g8s3.so $V/libpgh_g8s7.so $V/libpgh_s1.so $V/libpgh_s7.so 2>&1 | cut -c1-40,70-200 | tee gpurun_out/bsf_v7_decompose_scale23.log
