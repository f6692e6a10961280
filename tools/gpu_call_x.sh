#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for ps in 1 0 1 0; do
echo "== PGH_PUBLISHED_STATE=$ps"
PGH_PUBLISHED_STATE=$ps timeout 300 python tools/probe_default_rule.py 2>&1 | tail -1
PGH_PUBLISHED_STATE=$ps timeout 600 python bench.py --no-cpu --no-secondary --steps 20 --warmup 3 > $O/x_bench_$ps.json 2> $O/x_bench_$ps.err
python - <<PY
import json
d=json.loads(open("$O/x_bench_$ps.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["device_loop_ms_per_step"], d["config"]["iterations_per_step"][:6])
PY
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 1800 python -m pytest tests -x -q -m gpu > $O/x_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/x_tests.log
timeout 400 python tests/stress_filters.py --seconds 200 --seed 31 > $O/x_stress.log 2>&1; echo "stress rc=$?"; tail -1 $O/x_stress.log
