"""How few bytes can the gather pass of the 64-seed SpMM (k_mm_partial, configs[2]) move on THIS graph in THIS stream order?  (VERDICT r5
item 8: "prove the bound or change the bytes".)

The multi-seed image streams the entries of M^T row by row (one column block, rows and sources relabelled by descending reference count);
every entry gathers one row of the slab (64 seeds x 4 B = 256 B).  For ANY cache that holds C slab rows -- whatever its replacement
policy, Belady's included -- a stretch of the stream that touches D distinct rows misses at least D - C of them (the cache holds at most C of
them when the stretch begins).  Summed over disjoint stretches that is a floor under the gather traffic of the stream order, for the L2 of
one XCD (4 MB = 16 384 rows), for all eight pooled (131 072 rows: no replication, perfect sharing) and for L2s + Infinity Cache pooled
(256 MB more = 1 179 648 rows).  The stretch length that maximises the floor is searched for.  The floor counts neither what the
partial-sum / epilogue passes move nor the concurrency of 256 CUs (which interleaves 256 positions of the stream: worse, never better).

    python tools/count_gather_rows.py [--scale 23] [--ef 16]       (CPU only; ~2 min and ~12 GB at scale 23)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rmat_np  # noqa: E402   (the numpy twin of the device generator: the bench graph, bit for bit)

ROW_BYTES = 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=23)
    ap.add_argument("--ef", type=int, default=16)
    args = ap.parse_args()
    t0 = time.time()
    A = rmat_np.rmat_csr(args.scale, args.ef, seed=0)               # rows = sources, cols = destinations, data = multiplicities
    n = A.shape[0]
    counts = np.asarray(A.sum(axis=1)).ravel()                      # references of every source (multiplicities included)
    has_row = np.asarray(A.sum(axis=0)).ravel() > 0
    key = (np.minimum(counts, 0x7fffffff).astype(np.int64) << 1) | has_row.astype(np.int64)       # k_relabel_keys
    order = np.argsort(-key, kind="stable")
    new_id = np.empty(n, dtype=np.int64)
    new_id[order] = np.arange(n)
    MT = A.T.tocsr()                                                # rows = destinations (output rows), cols = sources
    MT.sort_indices()
    rows_old = np.repeat(np.arange(n), np.diff(MT.indptr))
    stream_rows, stream_cols = new_id[rows_old], new_id[MT.indices]
    perm = np.lexsort((stream_cols, stream_rows))                   # the image's order: (row, source), both relabelled
    seq = stream_cols[perm].astype(np.int32)                        # the slab row every entry gathers, in stream order
    del stream_rows, stream_cols, rows_old, perm
    E = len(seq)
    live = int((counts > 0).sum())
    print(f"RMAT scale {args.scale} ef {args.ef}: {n} ids, {live} referenced sources, {E} distinct (row, source) entries "
          f"({int(counts.sum())} with multiplicities); built in {time.time() - t0:.0f} s")
    print(f"  every entry a miss:            {E * ROW_BYTES / 1e9:8.2f} GB")
    print(f"  every referenced row once:     {live * ROW_BYTES / 1e9:8.2f} GB   (what a source-major pass -- a scatter with atomics on the output slab -- would gather)")
    for label, cap in (("one XCD's L2 (4 MB)", 16384), ("eight L2s pooled (32 MB)", 131072), ("L2s + Infinity Cache pooled (288 MB)", 131072 + 1048576)):
        best = (0.0, 0)
        for stretch_rows in (2, 3, 4, 6, 8, 12, 16, 24, 32):        # stretches that touch ~ stretch_rows x C distinct rows
            # stretch boundaries by entry count: choose the length so that an average stretch holds ~ stretch_rows * cap DISTINCT rows; a
            # coarse search over lengths is enough (the floor is a maximum over any partition)
            length = int(cap * stretch_rows * 1.6)
            floor = 0
            for lo in range(0, E, length):
                d = len(np.unique(seq[lo:lo + length]))
                floor += max(0, d - cap)
            if floor * ROW_BYTES / 1e9 > best[0]:
                best = (floor * ROW_BYTES / 1e9, length)
        print(f"  floor, {label:38s} {best[0]:8.2f} GB   (stretches of {best[1]} entries)")
    # the hottest rows resident for ever (a pinned hot set of H rows) + everything else a miss unless repeated back to back
    for hot in (16384, 131072, 1179648):
        cold = seq[seq >= hot]
        repeats = int((cold[1:] == cold[:-1]).sum())
        print(f"  {hot:8d} hottest rows pinned, the rest fetched per use: {(len(cold) - repeats) * ROW_BYTES / 1e9:8.2f} GB   ({100.0 * len(cold) / E:.1f} % of the entries lie outside)")


if __name__ == "__main__":
    main()
