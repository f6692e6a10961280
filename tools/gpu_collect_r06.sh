#!/bin/bash
# round 6, final collection on the final sources: the whole GPU suite, then bench.py with its rocprofv3 trace / PMC / SQ passes, the PMC
# passes of the multi-seed and the f64 routes, the one-rank partitioned bench.  Summaries are installed by tools/install_profiles.sh r06.
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/r06_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r06_tests.log
bash tools/gpu_bench_call.sh r06 2>&1 | tail -30
bash tools/gpu_sq_call.sh sq_r06 2>&1 | tail -8
bash tools/gpu_pmc_cmd.sh spmm_r06 "k_mm_" tools/probe_batch_kernels.py --scale 23 --batch 64 2>&1 | tail -12
bash tools/gpu_pmc_cmd.sh cheb_r06 "k_bsf64|k_pb64" tools/probe_cheb.py --scale 23 2>&1 | tail -12
cd $R
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/r06_part1.json 2> $O/r06_part1.err; echo "part1 rc=$?"; cat $O/r06_part1.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
