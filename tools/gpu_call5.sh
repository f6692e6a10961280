#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
V=tools/variants
export PGH_BLOCKS=4
for lib in base g1 g3 g4 g5 ipt4 ipt11; do
  timeout 300 python tools/probe_variants.py --scale 23 $V/libpgh_$lib.so 2>&1 | tail -1
done > $O/bsf_probe23.log 2>&1
cat $O/bsf_probe23.log
cd /tmp && export TMPDIR=/tmp
for pass in "A:FETCH_SIZE" "B:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "C:TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "D:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "E:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" "F:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc5_$name -- python3 $R/tools/probe_variants.py --scale 23 --iters 4 $R/tools/variants/libpgh_base.so > $O/pmc5_$name.log 2>&1
  echo "pmc $name rc=$?"
done
cd $R; python tools/summarize_pmc.py $O/pmc5_summary.json $O/pmc5_A $O/pmc5_B $O/pmc5_C $O/pmc5_D $O/pmc5_E $O/pmc5_F | grep -E "bsf|resid"
