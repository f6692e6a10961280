#!/bin/bash
# GPU call 2: full gpu tests, variant probes, counter list, first PMC passes
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/gpu_tests.log
python tools/probe_variants.py --scale 22 --copy tools/variants/libpgh_base.so > $O/variants22.log 2>&1
python tools/probe_variants.py --scale 22 tools/variants/libpgh_g1.so tools/variants/libpgh_g2.so tools/variants/libpgh_g3.so tools/variants/libpgh_ipt4.so tools/variants/libpgh_ipt11.so tools/variants/libpgh_ipt15.so >> $O/variants22.log 2>&1
cat $O/variants22.log
python tools/probe_variants.py --scale 23 tools/variants/libpgh_base.so tools/variants/libpgh_g2.so tools/variants/libpgh_g3.so > $O/variants23.log 2>&1; cat $O/variants23.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
for pass in "A:FETCH_SIZE" "B:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "D:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_$name -- python3 $R/tools/probe_variants.py --scale 22 --iters 5 $R/tools/variants/libpgh_base.so > $O/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/probe_variants.py --scale 22 --iters 20 $R/tools/variants/libpgh_base.so > $O/trace.log 2>&1
echo "trace rc=$?"
find $O -name "*.csv" | head -30
