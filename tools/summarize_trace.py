"""Per-kernel launch durations of the engine's kernels from a rocprofv3 --kernel-trace CSV.
Usage: python tools/summarize_trace.py <kernel_trace.csv> > profiles/rNN/<name>_step_kernels.csv

'working' launches: the run-ahead loop leaves <= 2 no-op iterations (2-5 us each) after convergence; a launch counts as
working when it lasts at least half as long as the kernel's longest launch."""
import collections
import csv
import re
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
    if not m:
        continue
    rows[m.group(1) + (m.group(2) or "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,launches,avg_all_us,working_launches,avg_working_us,min_us,max_us")
keep = ("k_bsf_partial", "k_pb_gather", "k_pb_finish", "k_bsf_fixup", "k_bsf_combine", "k_step_", "k_permute", "k_spmm", "k_mm_")
for name, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if not name.startswith(keep):
        continue
    work = [x for x in d if x >= 0.5 * max(d)]
    print(f'"{name}",{len(d)},{sum(d) / len(d):.2f},{len(work)},{sum(work) / len(work):.2f},{min(d):.2f},{max(d):.2f}')
