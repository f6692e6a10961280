"""Summarise rocprofv3 --pmc counter_collection.csv files into per-kernel averages (JSON).
Usage: python tools/summarize_pmc.py out.json dir_or_csv [dir_or_csv ...]

Derived fields (MI355X_MICROARCH.md, HBM/rocprofv3 section): on gfx950 FETCH_SIZE tallies 128-B read requests at
64 B, so hbm_read_bytes = 2 * FETCH_SIZE * 1024 for wide streaming reads (calibrated here on the residual kernel:
TCC_MISS * 128 B == its 8*n algorithmic bytes); WRITE_SIZE * 1024 is exact for streaming stores."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    """Hash of the kernel sources the counters were collected from (bench.py drops `traffic` when it does not match)."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "pygrank_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pygrank_amd", "csrc", "*.h"))):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:48]


def main():
    out_path, inputs = sys.argv[1], sys.argv[2:]
    files = []
    for p in inputs:
        files += [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, counters in sorted(agg.items()):
        if not k.startswith("k_"):
            continue
        row = {c: sum(v) / len(v) for c, v in counters.items()}
        row["dispatches"] = max(len(v) for v in counters.values())
        if "FETCH_SIZE" in row:
            row["hbm_read_bytes_corrected"] = 2 * row["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in row:
            row["hbm_write_bytes"] = row["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in row and "TCC_MISS_sum" in row:
            row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
            row["l2_miss_bytes_128B"] = row["TCC_MISS_sum"] * 128
        out[k] = row
    shown = dict(out)
    out["_meta"] = {"csrc_sha16": csrc_sha16()}
    json.dump(out, open(out_path, "w"), indent=1)
    for k, row in shown.items():
        print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in row.items()})


if __name__ == "__main__":
    main()
