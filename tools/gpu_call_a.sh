#!/bin/bash
# round 4, first GPU call: the partitioned loop's tests, the new full-size tests, 1-rank partitioned bench, headline bench
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_partitioned or two_gpus or scipy_graph_on_gpu or bench_two_ranks" > $O/a_dist_tests.log 2>&1; echo "dist tests rc=$?"; tail -15 $O/a_dist_tests.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "signed or independent or cfg5" > $O/a_full_tests.log 2>&1; echo "fullsize tests rc=$?"; tail -15 $O/a_full_tests.log
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/a_part1.json 2> $O/a_part1.err; echo "part1 rc=$?"; cat $O/a_part1.json; tail -3 $O/a_part1.err
PGH_FUSED_RES=0 timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/a_part1_unfused.json 2> $O/a_part1_unfused.err; echo "part1 unfused rc=$?"; cat $O/a_part1_unfused.json
timeout 600 python bench.py --gpus 1 > $O/a_bench.json 2> $O/a_bench.err; echo "bench rc=$?"; cat $O/a_bench.json; tail -3 $O/a_bench.err
