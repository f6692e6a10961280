#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for head in 4096 29696 131072 0; do
  export PGH_DEAL_HEAD=$head
  if [ $head = 0 ]; then export PGH_DEAL_RUNS=0; fi
  timeout 300 python tools/probe_default_rule.py > $O/k_default_$head.log 2>&1; echo "head $head: $(tail -1 $O/k_default_$head.log)"
  timeout 600 python bench.py --gpus 1 --no-cpu --no-secondary > $O/k_bench_$head.json 2> $O/k_bench_$head.err; python -c "
import json; d=json.load(open('$O/k_bench_$head.json')); print('   bench', d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'])"
  timeout 600 python bench.py --gpus 1 --no-cpu --no-secondary > $O/k_bench2_$head.json 2> $O/k_bench2_$head.err; python -c "
import json; d=json.load(open('$O/k_bench2_$head.json')); print('   bench', d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'])"
done
