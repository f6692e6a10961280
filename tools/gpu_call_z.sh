#!/bin/bash
# large graphs on one GPU: the state of the final sources, and two knobs that exist
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
run() { name=$1; shift; env "$@" timeout 900 python bench.py --no-cpu --no-secondary --steps 6 --warmup 2 --scale $SC --ef $EF > $O/z_$name.json 2> $O/z_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/z_$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$name", "scale $SC ef $EF", d["value"], "GTEPS", d["ms_per_step"], "ms/run", "frac", r["frac"], r["kernels_avg_us"], r["format"][-110:])
except Exception as e:
    print("$name failed", e, open("$O/z_$name.err").read()[-400:])
PY
}
SC=25 EF=16; run s25 PGH_DEBUG=0
SC=25 EF=16; run s25_binfill12 PGH_PB_BINFILL=12
SC=25 EF=16; run s25_blocks4 PGH_BLOCKS=4
SC=27 EF=8;  run s27 PGH_DEBUG=0
SC=27 EF=8;  run s27_binfill12 PGH_PB_BINFILL=12
SC=26 EF=16; run s26 PGH_DEBUG=0
