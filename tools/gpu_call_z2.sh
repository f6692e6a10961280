#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
run() { name=$1; shift; env "$@" timeout 900 python bench.py --no-cpu --no-secondary --steps 6 --warmup 2 --scale $SC --ef $EF > $O/z_$name.json 2> $O/z_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/z_$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]; k=r["kernels_avg_us"]
    print("$name", "scale $SC ef $EF", d["value"], "GTEPS frac", r["frac"], "spmv", k["spmv"], "pbA", k["pb_gather"], "pbB", k["pb_finish"], r["format"].split("first:")[-1])
except Exception as e:
    print("$name failed", e, open("$O/z_$name.err").read()[-300:])
PY
}
for bf in 6 9 12 16 24; do SC=25 EF=16; run s25_bf$bf PGH_PB_BINFILL=$bf; done
for bf in 6 9 12 16; do SC=24 EF=16; run s24_bf$bf PGH_PB_BINFILL=$bf; done
for bf in 6 9 12; do SC=23 EF=16; run s23_bf$bf PGH_PB_BINFILL=$bf; done
for bf in 9 16; do SC=26 EF=16; run s26_bf$bf PGH_PB_BINFILL=$bf; done
for w in 8 4 2; do for bf in 6 12; do echo "== slices world $w binfill $bf"; PGH_PB_BINFILL=$bf timeout 600 python tools/probe_partition.py --seeds --worlds $w 2>&1 | grep "step=" | cut -c1-170; done; done
