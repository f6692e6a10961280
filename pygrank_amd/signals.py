"""Graph signals: node-dict <-> dense HBM vector, operators, and the NodeRanking base.

Restates the behaviour of pygrank/core/signals.py (GraphSignal :10-193, NodeRanking :196-249, to_signal
:281-319) for the hip engine.  One deliberate difference in mechanism, not behaviour: dictionary-style reads
(``signal[node]``, ``items()``) go through a lazily downloaded host mirror instead of one device read per
node, because ``float(self._np[i])`` (signals.py:89-90) would cost one HIP sync per node on a device vector.
"""
from collections.abc import MutableMapping

import numpy as np

from pygrank_amd import backend
from pygrank_amd.device import DeviceVector


def _node_count(graph):
    return graph.shape[0] if hasattr(graph, "shape") else len(graph)


def _node_index(graph):
    """node -> position for a graph that brings no mapping of its own: what a preprocessor recorded on its outcome, the
    identity for matrix-like graphs, else the iteration order of the nodes (signals.py:41-50)."""
    recorded = getattr(graph, "_pygrank_node2id", None)
    if recorded is not None:
        return recorded
    if hasattr(graph, "shape"):
        return _IdentityMap(graph.shape[0])
    if getattr(graph, "_nodes_are_positions", False):    # AdjacencyWrapper: nodes 0 .. n - 1 (wrapgraph.py:4-22) -- no n-entry dictionary
        return _IdentityMap(len(graph))                  # per signal (0.6 s at 8.4 M nodes, in front of a 2.5-ms loop)
    return {node: position for position, node in enumerate(graph)}


def _dense_values(obj, count, node2id):
    """The backend primitive behind a signal (signals.py:54-66): an array is taken as it is (its length must be the node
    count), nothing means all ones, a node -> value mapping is staged on the host and uploaded once."""
    if obj is None:
        return backend.repeat(1.0, count)
    if isinstance(obj, DeviceVector) and len(obj) == count:  # the engine's own vector (or an expression over such): nothing to look at
        return obj
    if backend.is_array(obj):
        have = backend.length(obj)
        if have != count:
            raise Exception(f"a signal over {count} nodes cannot be built from {have} values")
        return backend.to_array(obj)
    staging = np.zeros(count, dtype=np.float64)
    for node, value in obj.items():
        staging[node2id[node]] = float(value)
    return backend.to_array(staging)


class GraphSignal(MutableMapping):
    """signals.py:10-193.  ``np`` is the backend primitive (a DeviceVector under the hip engine)."""

    def __init__(self, graph, obj, node2id=None):
        self.graph = graph
        self.node2id = _node_index(graph) if node2id is None else node2id
        self._host = None
        self._np = _dense_values(obj, _node_count(graph), self.node2id)

    # ---- backend primitive
    @property
    def np(self):                                         # signals.py:81-83
        value = self._np
        return value if isinstance(value, DeviceVector) else backend.to_array(value)

    @np.setter
    def np(self, value):                                  # signals.py:85-87
        value = self._compliant(value)
        self._np = value if isinstance(value, DeviceVector) else backend.to_array(value)
        self._host = None

    def filter(self, exclude=None):                       # signals.py:68-75
        if exclude is None:
            return self._np
        return backend.filter_out(self._np, to_signal(self, exclude)._np)

    def __rshift__(self, other):
        return other(self)

    # ---- mapping protocol
    def _mirror(self):
        if self._host is None:
            self._host = np.asarray(self._np, dtype=np.float64)
        return self._host

    def __getitem__(self, key):
        return float(self._mirror()[self.node2id[key]])

    def _store(self, key, value):
        self._np[self.node2id[key]] = value
        self._host = None                                 # the host mirror is stale now

    def __setitem__(self, key, value):
        self._store(key, float(value))

    def __delitem__(self, key):
        self._store(key, 0.0)                             # a signal has no holes: deleting a node zeroes it

    def __iter__(self):
        return iter(self.node2id)

    def __len__(self):
        return len(self.node2id)

    def __str__(self):
        return "{" + ", ".join(repr(k) + ": " + str(v) for k, v in self.items()) + "}"

    # ---- arithmetic (signals.py:114-178)
    def _compliant(self, other):
        if not isinstance(other, GraphSignal):
            return other
        if other.graph is not self.graph:
            raise Exception("the operands are graph signals of two different graphs")
        return other.np

    def _new(self, value):
        if isinstance(value, DeviceVector) and len(value) == len(self._np):       # the outcome of arithmetic on this signal's own vector
            out = GraphSignal.__new__(GraphSignal)
            out.graph, out.node2id, out._host, out._np = self.graph, self.node2id, None, value
            return out
        return GraphSignal(self.graph, value, self.node2id)

    def __add__(self, o): return self._new(self.np + self._compliant(o))
    def __radd__(self, o): return self._new(self._compliant(o) + self.np)
    def __sub__(self, o): return self._new(self.np - self._compliant(o))
    def __rsub__(self, o): return self._new(self._compliant(o) - self.np)
    def __mul__(self, o): return self._new(self.np * self._compliant(o))
    def __rmul__(self, o): return self._new(self._compliant(o) * self.np)
    def __pow__(self, o): return self._new(self.np ** self._compliant(o))
    def __rpow__(self, o): return self._new(self._compliant(o) ** self.np)
    def __truediv__(self, o): return self._new(self.np / self._compliant(o))
    def __rtruediv__(self, o): return self._new(self._compliant(o) / self.np)
    def __neg__(self): return self._new(-self.np)
    def __pos__(self): return self

    def __iadd__(self, o):
        self.np = self.np + self._compliant(o)
        return self

    def __isub__(self, o):
        self.np = self.np - self._compliant(o)
        return self

    def __imul__(self, o):
        self.np = self.np * self._compliant(o)
        return self

    def __ipow__(self, o):
        self.np = self.np ** self._compliant(o)
        return self

    def __itruediv__(self, o):
        self.np = self.np / self._compliant(o)
        return self

    def normalized(self, normalize=True, copy=True):      # signals.py:180-193
        target = self._new(backend.copy(self._np)) if copy else self
        if normalize:
            target._np = backend.self_normalize(target._np)
            target._host = None
        return target


class _IdentityMap:
    """node2id of an externally defined graph: ``{i: i for i in range(n)}`` (signals.py:46-47) without
    materialising n dictionary entries for 10^8-node graphs."""

    def __init__(self, n):
        self.n = int(n)

    def __getitem__(self, key):
        i = int(key)
        if i != key or not 0 <= i < self.n:
            raise KeyError(key)
        return i

    def __contains__(self, key):
        try:
            self[key]
            return True
        except (KeyError, TypeError, ValueError):
            return False

    def __iter__(self):
        return iter(range(self.n))

    def __len__(self):
        return self.n

    def keys(self):
        return range(self.n)

    def items(self):
        return ((i, i) for i in range(self.n))

    def get(self, key, default=None):
        return self[key] if key in self else default


class NodeRanking:
    """signals.py:196-249: anything that turns a graph signal into another one; ``rank`` is the one method a subclass
    provides, calling the object, piping a signal into it (``ranker | signal``) and chaining (``a >> b``) all end there."""

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        raise Exception(type(self).__name__ + " does not implement rank()")

    def __call__(self, *args, **kwargs):
        return self.rank(*args, **kwargs)

    def __or__(self, signal):
        if isinstance(signal, GraphSignal):
            return self.rank(signal)
        raise Exception("only graph signals can be piped into a ranker; build one with to_signal(graph, data)")

    def __rshift__(self, downstream):
        downstream << self                                  # the downstream object decides what receiving a ranker means
        return downstream

    def propagate(self, graph, features, *args, **kwargs):
        """One rank() per feature column, columns joined again (signals.py:225-226)."""
        ranked = [self.rank(graph, column, *args, **kwargs)._np for column in backend.separate_cols(features)]
        return backend.combine_cols(ranked)

    # ---- self-description (signals.py:228-249: references() lists what the algorithm consists of, cite() reads it out)
    def references(self):
        return [self._reference()]

    def _reference(self):
        return type(self).__name__

    def cite(self):
        parts = list(self.references())
        if len(parts) == 1:
            return parts[0]
        head, rest = parts[0], parts[1:]
        tail = rest[-1] if len(rest) == 1 else ", ".join(rest[:-1]) + " and " + rest[-1]
        return head + " with " + tail

    def __str__(self):
        return self.cite()

    def __and__(self, other):
        return _Sum(self, other)

    def __invert__(self):
        return _Negation(self)


class _Negation(NodeRanking):                              # signals.py:252-264
    def __init__(self, ranker):
        self.ranker = ranker

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        return -self.ranker.rank(graph, personalization, *args, **kwargs)


class _Sum(NodeRanking):                                   # signals.py:267-278
    def __init__(self, ranker1, ranker2):
        self.ranker1, self.ranker2 = ranker1, ranker2

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        return self.ranker1.rank(graph, personalization, *args, **kwargs) + \
            self.ranker2(graph, personalization, *args, **kwargs)


def to_signal(graph, obj):
    """signals.py:281-319: (graph, data) -> GraphSignal.  Either argument may be a signal: as `graph` it lends its graph
    and node order, as `obj` it is returned as it is (it must live on that graph).  A list shorter than the graph is a
    list of seed nodes."""
    if graph is None and obj is None:
        raise Exception("to_signal needs a graph, a signal, or both")
    if obj is None and isinstance(graph, GraphSignal):
        return graph                                        # a signal alone stands for itself
    node2id = None
    if isinstance(graph, GraphSignal):
        node2id, graph = graph.node2id, graph.graph
    elif graph is None:
        if not isinstance(obj, GraphSignal):
            raise Exception("without a graph the data must already be a graph signal")
        graph = obj.graph
    elif backend.is_array(graph):
        raise Exception("an array cannot serve as the graph of a signal")
    if isinstance(obj, GraphSignal):
        if obj.graph is not graph:
            raise Exception("the signal belongs to a different graph")
        return obj
    if isinstance(obj, list) and len(obj) != len(graph):
        obj = dict.fromkeys(obj, 1)
    return GraphSignal(graph, obj, node2id)
