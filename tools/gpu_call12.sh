#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export PGH_BLOCKS=8
cd /tmp && export TMPDIR=/tmp
for lib in base g7; do
for pass in "S1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "S2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_LDS" "T1:TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUSY_avr" "T2:TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" "L1:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc12_${lib}_$name -- python3 $R/tools/probe_variants.py --scale 23 --iters 3 $R/tools/variants/libpgh_$lib.so > $O/pmc12_${lib}_$name.log 2>&1
  echo "pmc $lib $name rc=$?"
done
done
cd $R
for lib in base g7; do python tools/summarize_pmc.py $O/pmc12_$lib.json $O/pmc12_${lib}_S1 $O/pmc12_${lib}_S2 $O/pmc12_${lib}_T1 $O/pmc12_${lib}_T2 $O/pmc12_${lib}_L1 | grep -E "bsf_partial"; done
