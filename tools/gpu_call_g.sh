#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 300 python tools/probe_default_rule.py > $O/g_default.log 2>&1; tail -1 $O/g_default.log
timeout 300 python tools/probe_default_rule.py > $O/g_default2.log 2>&1; tail -1 $O/g_default2.log
timeout 300 python tools/probe_dropout.py > $O/g_drop_blocked.log 2>&1; tail -1 $O/g_drop_blocked.log
PGH_FORMAT=csr timeout 300 python tools/probe_dropout.py --iters 6 > $O/g_drop_csrfmt.log 2>&1; tail -1 $O/g_drop_csrfmt.log
PGH_DROPOUT_CSR=1 timeout 300 python tools/probe_dropout.py --hooks --iters 6 > $O/g_drop_r3.log 2>&1; tail -1 $O/g_drop_r3.log
timeout 300 python tools/probe_dropout.py --hooks --iters 6 > $O/g_drop_hooks.log 2>&1; tail -1 $O/g_drop_hooks.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernels or core or filters_match or generic_route" > $O/g_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/g_tests.log
timeout 600 python bench.py --gpus 1 --no-cpu > $O/g_bench.json 2> $O/g_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/g_bench.json')); print(d['value'], d['ms_per_step'], d['config']['device_loop_ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_avg_us']); [print('  ',k,v.get('gteps',v.get('edge_vector_products_per_s_G')),v.get('device_step_us')) for k,v in d['secondary'].items()]"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
