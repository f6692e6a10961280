#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
V=tools/variants
for lib in base aux1 aux2 aux16 aux17 aux3; do
  timeout 300 python tools/probe_variants.py --scale 23 $V/libpgh_$lib.so 2>&1 | tail -1
done > gpurun_out/aux_probe23.log 2>&1
cat gpurun_out/aux_probe23.log
