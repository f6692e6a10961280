#!/bin/bash
# kernel durations of a small-graph PageRank loop (tools/probe_latency.py SCALE) under rocprofv3 --kernel-trace --stats
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
S=${1:-10}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small_$S -- python3 $R/tools/probe_latency.py $S > $R/gpurun_out/small_$S.log 2>&1; echo "rc=$?"
f=$(find $R/gpurun_out/small_$S -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
g=$(find $R/gpurun_out/small_$S -name "*kernel_trace.csv" | head -1)
python3 - "$g" <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-50s dur %6.2f gap %6.2f" % (r["Kernel_Name"][:50], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
EOF
rm -rf $R/gpurun_out/small_$S
