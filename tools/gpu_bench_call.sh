#!/bin/bash
# bench.py (N=1) + rocprofv3 kernel-trace/stats + PMC passes of the same command (every profiler run under its own timeout)
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r05}
timeout 600 python bench.py --gpus 1 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cat $O/bench_$TAG.json; tail -3 $O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$TAG -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu --no-secondary > $O/trace_$TAG.log 2>&1; echo "trace rc=$?"
for pass in "A:FETCH_SIZE" "B:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_${TAG}_$name -- python3 $R/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu --no-secondary > $O/pmc_${TAG}_$name.log 2>&1
  echo "pmc $name rc=$?"
done
cd $R; python tools/summarize_pmc.py $O/pmc_${TAG}_summary.json $O/pmc_${TAG}_A $O/pmc_${TAG}_B | grep -E "bsf_partial|bsf_combine|bsf_fixup|k_pb_|residual|step_close"
python tools/summarize_trace.py $(find $O/trace_$TAG -name "*kernel_trace.csv" | head -1) > $O/trace_${TAG}_step_kernels.csv; cp $(find $O/trace_$TAG -name "*kernel_stats.csv" | head -1) $O/trace_${TAG}_kernel_stats.csv; cat $O/trace_${TAG}_step_kernels.csv | head -20
