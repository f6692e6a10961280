#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_partitioned or scipy_graph_on_gpu or bench_two_ranks" > $O/b_dist_tests.log 2>&1; echo "dist tests rc=$?"; tail -15 $O/b_dist_tests.log
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/b_part1.json 2> $O/b_part1.err; echo "part1 rc=$?"; cat $O/b_part1.json; tail -3 $O/b_part1.err
PGH_DIST_SINGLE_STREAM=1 timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/b_part1_ss.json 2> $O/b_part1_ss.err; echo "part1 single stream rc=$?"; cat $O/b_part1_ss.json
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg5" > $O/b_full_tests.log 2>&1; echo "fullsize tests rc=$?"; tail -5 $O/b_full_tests.log
