#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for v in 0; do
  timeout 600 python tools/probe_upload.py --scale 21 2>&1 | tail -2
done > gpurun_out/upload_probe.log 2>&1
cat gpurun_out/upload_probe.log
