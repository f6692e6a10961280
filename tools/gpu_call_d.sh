#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/d_part1.json 2> $O/d_part1.err; echo "part1 rc=$?"; python -c "
import json; d=json.load(open('$O/d_part1.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernels_avg_us'])"
PGH_DIST_SINGLE_STREAM=0 timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/d_part1_3q.json 2> $O/d_part1_3q.err; echo "part1 three queues rc=$?"; python -c "
import json; d=json.load(open('$O/d_part1_3q.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernels_avg_us'])"
timeout 300 python tools/probe_default_rule.py > $O/d_default.log 2>&1; tail -2 $O/d_default.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cheb or heat or core or kernels or cfg2 or filters_match" > $O/d_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/d_tests.log
cd /tmp && export TMPDIR=/tmp
for pass in "A:FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "B:WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/d_pmc_$name -- python3 $R/tools/probe_default_rule.py --runs 4 > $O/d_pmc_$name.log 2>&1; echo "pmc $name rc=$?"
done
cd $R; python tools/summarize_pmc.py $O/d_pmc_default.json $O/d_pmc_A $O/d_pmc_B | grep -E "permute_out|pair_scan|permute_in_pair"
rm -rf $O/d_pmc_A $O/d_pmc_B
timeout 600 python bench.py --gpus 1 --no-cpu > $O/d_bench.json 2> $O/d_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/d_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us']); [print('  ',k,v.get('gteps',v.get('edge_vector_products_per_s_G')),v.get('device_step_us')) for k,v in d['secondary'].items()]"
