"""Synthetic power-law inputs built directly in HBM (BASELINE.json configs[1..4]).

``rmat_graph`` returns what ``preprocessor(normalization=...)(AdjacencyWrapper(A))`` would return for the
RMAT adjacency A -- an ``Adjacency`` whose ``.array`` is the engine's backend graph, with the
``_pygrank_node2id`` metadata of preprocessing.py:151 -- without materialising A on the host: edges are
generated, de-duplicated into weights, normalised and transposed on the GPU (pgh_graph_rmat).  The result can
be passed to any filter in place of a graph (preprocessing.py:88-90: preprocessed inputs are returned as-is).
"""
import ctypes as C

from pygrank_amd import _lib as L
from pygrank_amd import backend
from pygrank_amd.device import DeviceGraph
from pygrank_amd.preprocessing import Adjacency
from pygrank_amd.signals import _IdentityMap

_NORMALIZATIONS = {"col": 0, "symmetric": 1, "none": 2}


def rmat_device_graph(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=0, normalization="col", symmetrize=False,
                      row_begin=0, row_end=0):
    L.ensure_init()
    h = L.c_graph()
    L.check(L.lib().pgh_graph_rmat(int(scale), int(edge_factor), float(a), float(b), float(c), int(seed),
                                   _NORMALIZATIONS[normalization], 1 if symmetrize else 0, int(row_begin), int(row_end),
                                   C.byref(h)))
    vals = [C.c_int64() for _ in range(4)]
    L.check(L.lib().pgh_graph_info(h, *[C.byref(v) for v in vals]))
    g = DeviceGraph(h, (vals[0].value, vals[1].value), vals[2].value)
    return g


def rmat_graph(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=0, normalization="auto", symmetrize=False):
    """Preprocessed RMAT graph: directed -> "col", symmetrised -> "symmetric" under normalization="auto"
    (preprocessing.py:101-102)."""
    if normalization == "auto":
        normalization = "symmetric" if symmetrize else "col"
    g = rmat_device_graph(scale, edge_factor, a, b, c, seed, normalization, symmetrize)
    ret = Adjacency(g)
    ret._pygrank_preprocessed = {backend.backend_name(): ret}
    ret._pygrank_node2id = _IdentityMap(g.shape[0])
    ret.directed = not symmetrize

    def is_directed():
        return ret.directed
    ret.is_directed = is_directed
    return ret
