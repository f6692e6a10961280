#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -x -q -m gpu > $O/z3_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/z3_tests.log
PGH_PB=1 PGH_PB_FORCE=1 timeout 400 python tests/stress_filters.py --seconds 200 --seed 41 > $O/z3_stress.log 2>&1; echo "stress(pb) rc=$?"; tail -1 $O/z3_stress.log
env PGH_TEST_ENGINE=hip PGH_DIST_BACKEND=gloo PGH_DIST_NATIVE=external timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29802 tests/stress_partitioned.py --seconds 200 --seed 42 --max-scale 17 > $O/z3_part2.log 2>&1; echo "part x2 rc=$?"; grep -E "stress ok|FAILED" $O/z3_part2.log | head -2 | cut -c1-400
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
SC=23 EF=16; run auto23 PGH_DEBUG=0
SC=24 EF=16; run auto24 PGH_DEBUG=0
SC=25 EF=16; run auto25 PGH_DEBUG=0
SC=25 EF=16; run fill6_25 PGH_PB_BINFILL=6
SC=26 EF=16; run auto26 PGH_DEBUG=0
SC=27 EF=8; run auto27 PGH_DEBUG=0
timeout 900 python tools/probe_partition.py --seeds --worlds 2 4 8 2>&1 | grep "step=" | cut -c1-170
