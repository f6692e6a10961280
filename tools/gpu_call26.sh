#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
for cfg in "23 16" "27 8"; do set -- $cfg
timeout 900 python bench.py --force-partitioned --scale $1 --ef $2 --steps 5 --warmup 1 --no-cpu 2> gpurun_out/part_scale$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale $1', d['value'], d['ms_per_step'], d['config']['iterations_per_step'], d['config']['exchange_bytes_per_iteration_per_gpu'], d['config']['gather_vector_slots'], d['config']['n'], {k:(round(v,1) if v else v) for k,v in d['roofline']['kernels_avg_us'].items()})"
tail -2 gpurun_out/part_scale$1.err
done
