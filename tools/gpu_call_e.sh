#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 300 python tools/probe_default_rule.py > $O/e_default.log 2>&1; tail -2 $O/e_default.log
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/e_part1.json 2> $O/e_part1.err; echo "part1 rc=$?"; python -c "
import json; d=json.load(open('$O/e_part1.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernels_avg_us'])"
timeout 1800 python -m pytest tests -x -q -m gpu > $O/e_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/e_tests.log
timeout 600 python bench.py --gpus 1 > $O/e_bench.json 2> $O/e_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/e_bench.json')); print(d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us'], d['parity']); [print('  ',k,v.get('gteps',v.get('edge_vector_products_per_s_G')),v.get('device_step_us')) for k,v in d['secondary'].items()]"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
