#!/bin/bash
# HBM / L2 counters of one python tool (two --pmc passes) + kernel trace.  Usage: bash tools/gpu_pmc_cmd.sh TAG KERNEL_REGEX tools/probe_x.py [args...]
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=$1; PAT=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/${TAG}_pmc_$i -- python3 $R/$1 "${@:2}" > $O/${TAG}_pmc_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/$1 "${@:2}" > $O/${TAG}_trace.log 2>&1; echo "trace rc=$?"
cd $R; python tools/summarize_pmc.py $O/${TAG}_pmc.json $O/${TAG}_pmc_1 $O/${TAG}_pmc_2 $O/${TAG}_pmc_3 | grep -E "$PAT"
cp $(find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv; head -12 $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_pmc_1 $O/${TAG}_pmc_2 $O/${TAG}_pmc_3 $O/${TAG}_trace
