#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
for sc in 25 27; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout 900 python bench.py --force-partitioned --scale $sc --ef 8 --steps 3 --warmup 1 --no-cpu > gpurun_out/part_scale$sc.json 2> gpurun_out/part_scale$sc.err; echo "rc=$?"; tail -c 1500 gpurun_out/part_scale$sc.json; tail -3 gpurun_out/part_scale$sc.err
done
