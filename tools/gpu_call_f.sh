#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "dropout or kernels or core" > $O/f_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/f_tests.log
timeout 300 python tools/probe_dropout.py > $O/f_drop_blocked.log 2>&1; tail -2 $O/f_drop_blocked.log
PGH_DROPOUT_CSR=1 timeout 300 python tools/probe_dropout.py > $O/f_drop_csr.log 2>&1; tail -2 $O/f_drop_csr.log
timeout 300 python tools/probe_dropout.py --hooks --iters 6 > $O/f_drop_hooks.log 2>&1; tail -2 $O/f_drop_hooks.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f_trace_default -- python3 $R/tools/probe_default_rule.py --runs 10 > $O/f_trace_default.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f_trace_drop -- python3 $R/tools/probe_dropout.py --iters 6 > $O/f_trace_drop.log 2>&1; echo "trace rc=$?"
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/f_trace_default/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ids = [i for i, r in enumerate(rows) if "k_pair_scan" in r["Kernel_Name"]]
t0 = int(rows[ids[-2]]["Start_Timestamp"])
for r in rows[ids[-2] - 2: ids[-1] + 1]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} us  {r["Kernel_Name"][:70]}')
f = glob.glob("gpurun_out/f_trace_drop/**/*kernel_stats.csv", recursive=True)
print(open(f[0]).read()[:1500])
PY
