#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "not row_partitioned and not cfg5" > $O/r_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r_tests.log
for fp in 1 0 1 0; do
  echo "== PGH_FIRST_PRED=$fp"
  PGH_FIRST_PRED=$fp timeout 300 python tools/probe_default_rule.py 2>&1 | tail -1
  PGH_FIRST_PRED=$fp timeout 600 python bench.py --no-cpu --no-secondary --steps 20 --warmup 3 > $O/r_bench_$fp.json 2> $O/r_bench_$fp.err
  python - <<PY
import json
d=json.loads(open("$O/r_bench_$fp.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("paused_runs"), d["config"].get("parity"))
PY
done
