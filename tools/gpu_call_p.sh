#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "row_partitioned or cfg5 or two_gpus or partition" > $O/p_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/p_tests.log
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu --steps 20 --warmup 3 > $O/p_part$i.json 2> $O/p_part$i.err; echo "part rc=$?"
python - <<PY
import json
d=json.loads(open("$O/p_part$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("us_per_iteration"), d["config"].get("kernel_us"), d["config"].get("exchange"))
PY
done
PGH_DIST_GATHER_ALONE=1 timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu --steps 20 --warmup 3 > $O/p_part_ga.json 2> $O/p_part_ga.err; echo "part(gather alone) rc=$?"
python - <<PY
import json
d=json.loads(open("$O/p_part_ga.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("us_per_iteration"), d["config"].get("kernel_us"))
PY
timeout 600 python bench.py --no-cpu --no-secondary --steps 20 --warmup 3 > $O/p_bench.json 2> $O/p_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/p_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
