#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for i in 1 2; do
timeout 300 python tools/probe_default_rule.py 2>&1 | tail -1
timeout 600 python bench.py --no-cpu --steps 20 --warmup 3 > $O/w_bench_$i.json 2> $O/w_bench_$i.err
python - <<PY
import json
d=json.loads(open("$O/w_bench_$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["device_loop_ms_per_step"])
for k,v in (d.get("secondary") or {}).items(): print("   ", k, v.get("gteps", v.get("edge_vector_products_per_s_G")), v.get("device_step_us"))
PY
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 1800 python -m pytest tests -x -q -m gpu > $O/w_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/w_tests.log
