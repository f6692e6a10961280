#!/bin/bash
# copies the summaries a tools/gpu_bench_call.sh TAG + gpu_sq_call.sh sq_TAG + gpu_pmc_cmd.sh spmm_TAG / cheb_TAG collection left in
# gpurun_out/ into profiles/<round>/ under the names bench.py and DESIGN.md cite.  usage: bash tools/install_profiles.sh TAG [round dir, default r06]
set -e
T=$1; O=gpurun_out; P=profiles/${2:-r06}; mkdir -p $P
cp $O/pmc_${T}_summary.json $P/bench_n1_pmc.json
cp $O/sq_${T}_summary.json $P/bench_n1_sq_counters.json
cp $O/trace_${T}_kernel_stats.csv $P/bench_n1_kernel_stats.csv
cp $O/trace_${T}_step_kernels.csv $P/bench_n1_step_kernels.csv
cp $O/spmm_${T}_pmc.json $P/spmm_final_pmc.json
cp $O/spmm_${T}_kernel_stats.csv $P/spmm_final_kernel_stats.csv
cp $O/cheb_${T}_pmc.json $P/cheb_f64_final_pmc.json
cp $O/cheb_${T}_kernel_stats.csv $P/cheb_f64_final_kernel_stats.csv
python - <<EOF
import json
for f in ("bench_n1_pmc", "spmm_final_pmc", "cheb_f64_final_pmc"):
    print(f, json.load(open("$P/" + f + ".json")).get("_meta"))
EOF
