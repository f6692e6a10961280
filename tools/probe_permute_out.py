"""Diagnostic: index arrays of the way out of the id space for tools/permute_probe.hip -- the engine's relabelling of the bench graph
(new = (r % 8) * blk + r / 8) and the same ranks dealt to the blocks in runs of Q.  Usage: python tools/probe_permute_out.py outdir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RANK", "0")
from pygrank_amd import _lib as L  # noqa: E402
from pygrank_amd.distributed import rmat_partitioned  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
L.ensure_init()
pgr = rmat_partitioned(23, 16, 0, 1, a=0.57, b=0.19, c=0.19)
perm = np.asarray(pgr.perm, dtype=np.int64)                  # new -> old
n = len(perm)
B = 8
blk = n // B
new = np.arange(n)
rank = (new % blk) * B + new // blk                          # the rank of every new id
deg = np.asarray(pgr.graph.degrees())                        # row sums of M in new ids: > 0 <=> the id has out-edges
live = int(0.55 * n)                                         # ranks below: not isolated (45 % of the ids of this graph are)
old_of_rank = np.empty(n, dtype=np.int64)
old_of_rank[rank] = perm
for q in (1, 8, 32, 64):
    r = np.arange(n)
    new_q = ((r // q) % B) * blk + (r // (q * B)) * q + r % q
    idx = np.full(n, -1, dtype=np.int32)
    keep = r < live
    idx[old_of_rank[keep]] = new_q[keep]
    idx.tofile(os.path.join(out, f"idx_runs_of_{q}.bin"))
    print(q, "written", int((idx >= 0).sum()), "gathers")
rng = np.random.default_rng(0)
idx = np.full(n, -1, dtype=np.int32)
idx[rng.permutation(n)[:live]] = rng.permutation(n)[:live].astype(np.int32)
idx.tofile(os.path.join(out, "idx_random.bin"))
print(n)
