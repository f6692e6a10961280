#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python bench.py > $O/t_bench.json 2> $O/t_bench.err; echo "bench rc=$?"
for fp in 1 0; do
PGH_FIRST_PRED=$fp PGH_DEBUG_RES=1 timeout 300 python bench.py --no-cpu --no-secondary --steps 12 --warmup 2 2>&1 >/dev/null | grep "recursive run" | sort | uniq -c | sort -rn | head -8
done
