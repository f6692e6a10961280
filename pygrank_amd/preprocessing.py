"""Graph -> normalised CSR -> HBM-resident backend graph, with the reference's caching semantics.

Restates pygrank/core/utils/preprocessing.py (Adjacency :9-28, to_sparse_matrix :50-152, MethodHasher
:181-230, preprocessor :233-287) and the graph entry points of pygrank/fastgraph (Graph.to_scipy_sparse_array
fastgraph.py:73-78, AdjacencyWrapper wrapgraph.py:4-22).  As in the reference, normalisation runs on the host
with scipy for every engine (preprocessing.py:99: ``with backend.Backend("numpy")``); the upload
``backend.scipy_sparse_to_backend`` (preprocessing.py:144) is where the matrix moves to HBM and is transposed.
"""
import uuid

import numpy as np
import scipy.sparse as sp

from pygrank_amd import backend
from pygrank_amd.signals import _IdentityMap


class Adjacency:
    """preprocessing.py:9-28: a thin owner of a backend matrix -- backend graph types are immutable handles, so the
    metadata a preprocessor attaches (cache of outcomes, node order) hangs on this object instead."""

    def __init__(self, array):
        self.array = array
        shape = getattr(array, "shape", None)
        if shape is not None:
            self.shape = shape

    def __len__(self):
        return len(self.array)

    def __getattr__(self, name):
        # sum / tocoo of the wrapped matrix (what the reference forwards explicitly); anything else is not part of a graph
        if name in ("sum", "tocoo"):
            return getattr(self.array, name)
        raise AttributeError(name)

    def _np(self):
        return self.array


class AdjacencyWrapper:
    """fastgraph/wrapgraph.py:4-22: a scipy matrix presented as a graph over the nodes 0 .. n - 1, in O(1)."""
    _nodes_are_positions = True                             # signals._node_index: the node -> position map is the identity

    def __init__(self, adj, directed=True):
        self.adj = getattr(adj, "array", adj)               # an Adjacency hands over what it owns
        self.directed = directed
        self.num_nodes = self.adj.shape[0]

    def is_directed(self):
        return self.directed

    def to_scipy_sparse_array(self):
        return self.adj

    def __len__(self):
        return self.num_nodes

    def __iter__(self):
        yield from range(self.num_nodes)


def _row_sums(M):                                           # numpy.py:76-77
    return np.asarray(M.sum(axis=1)).ravel()


def _inv_nonzero(v, sqrt=False):
    v = np.array(v, dtype=np.float64).ravel()
    if sqrt:
        v = np.sqrt(v)
    nz = v != 0
    v[nz] = 1.0 / v[nz]                                     # zero-degree rows stay zero (preprocessing.py:111)
    return v


def _scaling(v, shape):
    return sp.spdiags(v, 0, *shape, format="csr").tocsr()


def _with_factors(N, W, left, right):
    """Remember N = diag(left) W diag(right): the upload (DeviceGraph.from_scipy) hands the factors to the engine, which
    stores the 4 B/edge value-free layout when W holds small integer weights (include/pgh.h pgh_graph_from_factored_csr)."""
    N = sp.csr_array(N)
    if N.nnz == W.nnz and N.shape == W.shape:               # no entry vanished: same structure
        N.sort_indices()                                    # (scipy's product emits each row in reverse order)
        if not W.has_sorted_indices:
            W = W.copy()
            W.sort_indices()
        N._pgh_factors = (W, left, right)
    return N


def normalize_adjacency(M, normalization, reduction=None):
    """The host normalisations of preprocessing.py:109-142 on a scipy CSR matrix."""
    left_reduction = _row_sums if reduction is None else reduction

    def right_reduction(x):
        return left_reduction(x.T)

    if normalization == "col":                              # preprocessing.py:109-113
        left = _inv_nonzero(left_reduction(M))
        return _with_factors(_scaling(left, M.shape) @ M, M, left, None)
    if normalization == "symmetric":                        # preprocessing.py:131-138
        left, right = _inv_nonzero(left_reduction(M), True), _inv_nonzero(right_reduction(M), True)
        return _with_factors(_scaling(left, M.shape) @ M @ _scaling(right, M.shape), M, left, right)
    if normalization == "both":                             # preprocessing.py:123-130
        left, right = _inv_nonzero(left_reduction(M)), _inv_nonzero(right_reduction(M))
        return _with_factors(_scaling(left, M.shape) @ M @ _scaling(right, M.shape), M, left, right)
    if normalization == "laplacian":                        # preprocessing.py:114-122
        M = _scaling(_inv_nonzero(left_reduction(M), True), M.shape) @ M @ \
            _scaling(_inv_nonzero(right_reduction(M), True), M.shape)
        return -M + sp.eye(M.shape[0]).tocsr()
    if callable(normalization):                             # preprocessing.py:139-140
        return normalization(M)
    if normalization == "none":
        return M
    raise Exception(f"normalization {normalization!r}: expected none, col, symmetric, both, laplacian, auto or a callable")


def graph_to_scipy(G, weight="weight"):
    """preprocessing.py:103: fastgraph-style graphs expose to_scipy_sparse_array, networkx graphs are converted."""
    if hasattr(G, "to_scipy_sparse_array"):
        return G.to_scipy_sparse_array()
    import networkx as nx
    return nx.to_scipy_sparse_array(G, weight=weight, dtype=float)


def _identity(x):
    return x


def to_sparse_matrix(G, normalization="auto", weight="weight", renormalize=False, reduction=None,
                     transform_adjacency=_identity, cors=False):
    """preprocessing.py:50-152.  Returns an ``Adjacency`` whose ``.array`` is the engine's backend graph."""
    name = backend.backend_name()
    if hasattr(G, "_pygrank_preprocessed"):                 # preprocessing.py:88-98 (already preprocessed input)
        cache = G._pygrank_preprocessed
        if name in cache:
            return cache[name]
        ret = Adjacency(backend.scipy_sparse_to_backend(cache["numpy"].array))
        ret._pygrank_preprocessed = cache if cors else {name: ret}
        ret._pygrank_preprocessed[name] = ret
        ret._pygrank_node2id = G._pygrank_node2id
        return ret
    if isinstance(normalization, str):
        normalization = normalization.lower()
        if normalization == "auto":                         # preprocessing.py:101-102: directed graphs get "col"
            normalization = ("symmetric", "col")[bool(G.is_directed())]
    M = sp.csr_array(graph_to_scipy(G, weight), dtype=np.float64)
    renormalize = float(renormalize)                        # False / True are 0 / 1 self-loops
    square = M.shape[0] == M.shape[1]
    on_device = (name == "hip" and normalization in ("col", "symmetric", "both", "none", "laplacian")
                 and reduction is None and transform_adjacency is _identity and not cors
                 and (square or (normalization in ("col", "none") and renormalize == 0)))
    if on_device:
        # degree reductions, scaling, the self-loops of the renormalisation trick (preprocessing.py:107-108), the
        # laplacian's identity (:114-122) and the transposition in HBM; the host only hands over the raw adjacency
        from pygrank_amd.device import DeviceGraph
        ret = Adjacency(DeviceGraph.from_adjacency(M, normalization, renormalize))
    else:
        if renormalize:                                     # preprocessing.py:107-108
            M = M + sp.eye(M.shape[0]).tocsr() * renormalize
        M = normalize_adjacency(M, normalization, reduction)
        M = M if isinstance(M, sp.csr_array) else sp.csr_array(M)
        # preprocessing.py:143-145: the caller's last word on the matrix, then the upload to HBM
        M = transform_adjacency(M)
        ret = Adjacency(backend.scipy_sparse_to_backend(M))
    if cors:                                                # preprocessing.py:146-148
        ret._pygrank_preprocessed = {name: ret, "numpy": Adjacency(M)}
    else:
        ret._pygrank_preprocessed = {name: ret}
    if isinstance(G, AdjacencyWrapper):
        ret._pygrank_node2id = _IdentityMap(len(G))         # {v: i} over range(n) without n dict entries
    else:
        ret._pygrank_node2id = {node: at for at, node in enumerate(G)}   # preprocessing.py:151: iteration order
    return ret


def obj2id(obj):
    """A stable key for an argument (preprocessing.py:165-170): strings by hash, every other object by a uuid that is
    attached to it on first sight -- so two equal-looking graphs stay distinct and a graph keeps its key while it lives."""
    if isinstance(obj, str):
        return str(hash(obj))
    tag = getattr(obj, "uuid", None)
    if tag is None:
        tag = obj.uuid = uuid.uuid1()
    text = getattr(obj, "_uuid_text", None)                 # (formatting a uuid costs more than the dictionary lookup it keys)
    if text is None or text[0] is not tag:
        text = (tag, str(tag))
        try:
            obj._uuid_text = text
        except AttributeError:
            pass
    return text[1]


def _call_key(args, kwargs):
    """Key of a call for MethodHasher: positional identities, named identities, and the active backend (a graph
    preprocessed under another backend is another object)."""
    positional = ",".join(map(obj2id, args))
    named = ",".join(name + ":" + obj2id(value) for name, value in kwargs.items())
    return f"[{positional}]{{{named}}}{backend.backend_name()}"


class MethodHasher:
    """preprocessing.py:181-230: remembers the outcome of a method per call key (_call_key) while the caller promises not
    to mutate the arguments; without that promise every call goes through."""

    def __init__(self, method, assume_immutability=True):
        self._method, self._stored = method, {}
        self.assume_immutability = assume_immutability

    def clear_hashed(self):
        self._stored.clear()

    def __call__(self, *args, **kwargs):
        if not self.assume_immutability:
            return self._method(*args, **kwargs)
        key = _call_key(args, kwargs)
        try:
            return self._stored[key]
        except KeyError:
            outcome = self._stored[key] = self._method(*args, **kwargs)
            return outcome


def preprocessor(normalization="auto", assume_immutability=False, weight="weight", renormalize=False,
                 reduction=None, transform_adjacency=_identity, cors=False):
    """preprocessing.py:233-287."""
    options = dict(normalization=normalization, weight=weight, renormalize=renormalize, reduction=reduction, cors=cors,
                   transform_adjacency=transform_adjacency)

    def preprocess(G):
        return to_sparse_matrix(G, **options)
    if not assume_immutability:
        return preprocess
    remembered = MethodHasher(preprocess)                   # one outcome per graph identity and backend
    remembered.__name__ = preprocess.__name__
    return remembered
