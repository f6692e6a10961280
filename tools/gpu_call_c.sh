#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_partitioned or scipy_graph_on_gpu or bench_two_ranks" > $O/c_dist_tests.log 2>&1; echo "dist tests rc=$?"; tail -15 $O/c_dist_tests.log
timeout 600 python bench.py --gpus 1 --force-partitioned --no-cpu > $O/c_part1.json 2> $O/c_part1.err; echo "part1 rc=$?"; cat $O/c_part1.json; tail -3 $O/c_part1.err
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "not row_partitioned and not scipy_graph_on_gpu and not bench_two_ranks and not cfg5 and not cfg3" > $O/c_tests.log 2>&1; echo "other tests rc=$?"; tail -15 $O/c_tests.log
timeout 300 python tools/probe_default_rule.py > $O/c_default.log 2>&1; cat $O/c_default.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_trace_default -- python3 $R/tools/probe_default_rule.py --runs 10 > $O/c_trace_default.log 2>&1; echo "trace rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/c_trace_default/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run: kernels after the last k_pair_scan
idx = max(i for i, r in enumerate(rows) if "k_pair_scan" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx - 2: idx + 14]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} us  {r["Kernel_Name"][:90]}')
PY
timeout 600 python bench.py --gpus 1 > $O/c_bench.json 2> $O/c_bench.err; echo "bench rc=$?"; cat $O/c_bench.json; tail -3 $O/c_bench.err
timeout 300 python tools/probe_cheb.py > $O/c_cheb.log 2>&1; tail -8 $O/c_cheb.log
timeout 300 python tools/probe_cheb.py --lib tools/variants/libpgh_b64d3.so > $O/c_cheb_d3.log 2>&1; tail -8 $O/c_cheb_d3.log
