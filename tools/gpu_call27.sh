#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for tr in 0 1 1; do
PGH_TRIM=$tr timeout 300 python bench.py --no-cpu --steps 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trim=$tr bench', d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'])"
done
