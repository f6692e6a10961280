#!/bin/bash
# same-box A/B of two builds of the library (ab/libpgh_hip_{old,new}.so): slice steps of the partitioned path, headline
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for round in 1 2; do
for v in old new; do
  cp ab/libpgh_hip_$v.so pygrank_amd/csrc/libpgh_hip.so
  timeout 600 python tools/probe_partition.py --seeds --worlds 2 8 > $O/m_slices_${v}_$round.log 2>&1
  echo "== $v $round"; grep -E "world=|step" $O/m_slices_${v}_$round.log | cut -c1-300 | tail -6
  timeout 300 python bench.py --steps 30 --warmup 5 > $O/m_bench_${v}_$round.json 2> $O/m_bench_${v}_$round.err
  python - <<PY
import json
d=json.loads(open("$O/m_bench_${v}_$round.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["config"].get("kernel_us"))
PY
done
done
cp ab/libpgh_hip_new.so pygrank_amd/csrc/libpgh_hip.so
