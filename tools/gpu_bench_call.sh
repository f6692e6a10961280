#!/bin/bash
# bench.py (N=1) + rocprofv3 kernel-trace/stats + PMC passes of the same command
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r01}
python bench.py --gpus 1 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cat $O/bench_$TAG.json; tail -3 $O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$TAG -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu > $O/trace_$TAG.log 2>&1; echo "trace rc=$?"
for pass in "A:FETCH_SIZE" "B:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_${TAG}_$name -- python3 $R/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu > $O/pmc_${TAG}_$name.log 2>&1
  echo "pmc $name rc=$?"
done
