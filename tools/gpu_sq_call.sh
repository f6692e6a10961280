#!/bin/bash
# SQ / LDS counters of the step kernels of bench.py (three --pmc passes; diagnostic).  Usage: bash tools/gpu_sq_call.sh TAG
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/${TAG}_$i -- python3 $R/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu --no-secondary > $O/${TAG}_$i.log 2>&1
  echo "sq pass $i rc=$?"
done
cd $R; python tools/summarize_pmc.py $O/${TAG}_summary.json $O/${TAG}_1 $O/${TAG}_2 $O/${TAG}_3 | grep -E "bsf_partial|bsf_combine<1|k_pb_gather|k_pb_finish<1|step_residual"
