#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
for poll in 0 1; do
  PGH_POLL=$poll timeout 300 python tools/probe_latency.py 14 2>&1 | grep latency
  PGH_POLL=$poll timeout 300 python tools/probe_latency.py 20 2>&1 | grep latency
  PGH_POLL=$poll timeout 300 python bench.py --no-cpu --steps 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['kernels_avg_us'])"
done > gpurun_out/poll_ab.log 2>&1
PROFILE=1 timeout 300 python tools/probe_latency.py 14 > gpurun_out/latency_profile.log 2>&1
cat gpurun_out/poll_ab.log; head -60 gpurun_out/latency_profile.log
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
