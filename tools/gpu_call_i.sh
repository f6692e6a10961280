#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 300 python tools/probe_permute_out.py /tmp/pp > $O/i_permute_gen.log 2>&1; tail -3 $O/i_permute_gen.log
timeout 120 tools/scratch/permute_probe 8388608 /tmp/pp/idx_runs_of_1.bin /tmp/pp/idx_runs_of_8.bin /tmp/pp/idx_runs_of_32.bin /tmp/pp/idx_runs_of_64.bin /tmp/pp/idx_random.bin 2>&1 | tee $O/i_permute.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_partitioned" > $O/i_dist_tests.log 2>&1; echo "dist tests rc=$?"; tail -4 $O/i_dist_tests.log
