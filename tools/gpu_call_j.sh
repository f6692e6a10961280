#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 300 python tools/probe_default_rule.py > $O/j_default.log 2>&1; tail -1 $O/j_default.log
PGH_DEAL_RUNS=0 timeout 300 python tools/probe_default_rule.py > $O/j_default_rr.log 2>&1; tail -1 $O/j_default_rr.log
timeout 600 python bench.py --gpus 1 --no-cpu --no-secondary > $O/j_bench.json 2> $O/j_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/j_bench.json')); print(d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'])"
PGH_DEAL_RUNS=0 timeout 600 python bench.py --gpus 1 --no-cpu --no-secondary > $O/j_bench_rr.json 2> $O/j_bench_rr.err; python -c "
import json; d=json.load(open('$O/j_bench_rr.json')); print('one by one:', d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'])"
timeout 1800 python -m pytest tests -x -q -m gpu > $O/j_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/j_tests.log
timeout 600 python bench.py --gpus 1 > $O/j_bench_full.json 2> $O/j_bench_full.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/j_bench_full.json')); print(d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'], d['parity']); [print('  ',k,v.get('gteps',v.get('edge_vector_products_per_s_G')),v.get('device_step_us')) for k,v in d['secondary'].items()]"
