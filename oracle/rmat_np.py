"""TEST INFRASTRUCTURE -- numpy restatement of the device RMAT edge generator.

The reference (pygrank) has no graph generator; synthetic power-law inputs are a
build-side addition.  This file mirrors ``pgh_rmat_edges`` in ``pygrank_amd/csrc/pgh_graphgen.hip``
bit for bit (integer-only arithmetic: splitmix64 hash, 32-bit thresholds) so CPU tests and golden
fixtures see exactly the graph the GPU generates.

Spec: Graph500-style RMAT, per-bit quadrant sampling with (a, b, c, d), no vertex
permutation, self-loops kept, duplicate edges summed into weights (coo->csr semantics of
``pygrank/fastgraph/fastgraph.py:73-78``).
"""
import numpy as np
import scipy.sparse as sp

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_G = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_E = np.uint64(0xD6E8FEB86659FD93)


def splitmix64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _G
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def thresholds(a, b, c):
    """32-bit integer thresholds for the quadrant choice (identical on host and device)."""
    ta = int(np.floor(a * 4294967296.0))
    tb = int(np.floor((a + b) * 4294967296.0))
    tc = int(np.floor((a + b + c) * 4294967296.0))
    return ta, tb, tc


def rmat_edges(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=0, first_edge=0, num_edges=None):
    """Returns (src, dst) int64 arrays of the edges [first_edge, first_edge+num_edges)."""
    total = (1 << scale) * edge_factor
    if num_edges is None:
        num_edges = total - first_edge
    e = np.arange(first_edge, first_edge + num_edges, dtype=np.uint64)
    ta, tb, tc = thresholds(a, b, c)
    src = np.zeros(num_edges, dtype=np.int64)
    dst = np.zeros(num_edges, dtype=np.int64)
    with np.errstate(over="ignore"):
        emix = e * _E
        for pair in range((scale + 1) // 2):
            key = splitmix64(np.uint64(seed) * _G + np.uint64(pair + 1))
            h = splitmix64(key ^ emix)
            for half in range(2):
                level = 2 * pair + half
                if level >= scale:
                    break
                u = (h >> np.uint64(32)) if half == 0 else (h & np.uint64(0xFFFFFFFF))
                u = u.astype(np.int64)
                rbit = (u >= tb).astype(np.int64)                       # quadrants c, d -> src bit 1
                cbit = (((u >= ta) & (u < tb)) | (u >= tc)).astype(np.int64)  # quadrants b, d -> dst bit 1
                shift = scale - 1 - level
                src |= rbit << shift
                dst |= cbit << shift
    return src, dst


def rmat_csr(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=0):
    """scipy CSR adjacency (fp64 weights = edge multiplicities), rows = sources."""
    n = 1 << scale
    total = n * edge_factor
    chunk = 1 << 25                              # edges per pass: bounds the temporaries of the hash at full bench sizes
    A = None
    for first in range(0, total, chunk):
        src, dst = rmat_edges(scale, edge_factor, a, b, c, seed, first_edge=first, num_edges=min(chunk, total - first))
        part = sp.coo_array((np.ones(len(src)), (src, dst)), shape=(n, n)).tocsr()
        part.sum_duplicates()
        A = part if A is None else (A + part).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def seed_nodes(A, count=100, seed=1):
    """``count`` seed nodes drawn without replacement among nodes with out-degree > 0."""
    outdeg = np.diff(A.indptr)
    candidates = np.flatnonzero(outdeg > 0)
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(candidates, size=min(count, len(candidates)), replace=False))
