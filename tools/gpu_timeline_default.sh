#!/bin/bash
# kernel timeline of the last runs of tools/probe_default_rule.py: name, start (us), duration (us), gap to the previous kernel's end (us)
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_default -- python3 $R/tools/probe_default_rule.py --runs 6 > $R/gpurun_out/tl_default.log 2>&1; echo "trace rc=$?"
f=$(find $R/gpurun_out/tl_default -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$R/gpurun_out/timeline_default.csv" <<'PYEOF'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = prev = None
with open(sys.argv[2], "w") as out:
    for r in rows[-36:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None:
            t0 = prev = s
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        out.write("%s,%.1f,%.1f,%.1f\n" % (m.group(1) if m else r["Kernel_Name"][:40].replace(",", ";"), (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
        prev = e
PYEOF
rm -rf $R/gpurun_out/tl_default
tail -1 $R/gpurun_out/tl_default.log
cat $R/gpurun_out/timeline_default.csv
