#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_partitioned or two_gpus or scipy_graph_on_gpu or bench_two_ranks" > $O/h_dist_tests.log 2>&1; echo "dist tests rc=$?"; tail -8 $O/h_dist_tests.log
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/h_part1.json 2> $O/h_part1.err; echo "part1 rc=$?"; python -c "
import json; d=json.load(open('$O/h_part1.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernels_avg_us'], d['config'].get('finish_in_two_launches'))"
PGH_DIST_SINGLE_STREAM=0 timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/h_part1_3q.json 2> $O/h_part1_3q.err; python -c "
import json; d=json.load(open('$O/h_part1_3q.json')); print('3q', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernels_avg_us'], d['config'].get('finish_in_two_launches'))"
PGH_DIST_SINGLE_STREAM=0 PGH_DIST_FINISH_SPLIT=2 timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/h_part1_3q_split.json 2> $O/h_part1_3q_split.err; python -c "
import json; d=json.load(open('$O/h_part1_3q_split.json')); print('3q split', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernels_avg_us'], d['config'].get('finish_in_two_launches'))"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg5" > $O/h_full_tests.log 2>&1; echo "fullsize tests rc=$?"; tail -4 $O/h_full_tests.log
