#!/bin/bash
# kernel timeline of the last launches of a short bench run: name, start (us), duration (us), gap to the previous kernel's end (us)
# usage (through gpurun): bash tools/gpu_timeline.sh TAG [extra bench.py args]
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
TAG=${1:-tl}; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -- python3 $R/bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu --no-secondary "$@" > $R/gpurun_out/tl_$TAG.log 2>&1; echo "trace rc=$?"
f=$(find $R/gpurun_out/tl_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$R/gpurun_out/timeline_$TAG.csv" <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = prev = None
with open(sys.argv[2], "w") as out:
    for r in rows[-160:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None:
            t0 = prev = s
        out.write("%s,%.2f,%.2f,%.2f\n" % (r["Kernel_Name"][:60].replace(",", ";"), (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
        prev = e
EOF
rm -rf $R/gpurun_out/tl_$TAG
tail -2 $R/gpurun_out/tl_$TAG.log
