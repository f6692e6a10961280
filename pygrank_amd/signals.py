"""Graph signals: node-dict <-> dense HBM vector, operators, and the NodeRanking base.

Restates the behaviour of pygrank/core/signals.py (GraphSignal :10-193, NodeRanking :196-249, to_signal
:281-319) for the hip engine.  One deliberate difference in mechanism, not behaviour: dictionary-style reads
(``signal[node]``, ``items()``) go through a lazily downloaded host mirror instead of one device read per
node, because ``float(self._np[i])`` (signals.py:89-90) would cost one HIP sync per node on a device vector.
"""
from collections.abc import MutableMapping

import numpy as np

from pygrank_amd import backend


class GraphSignal(MutableMapping):
    """signals.py:10-193.  ``np`` is the backend primitive (a DeviceVector under the hip engine)."""

    def __init__(self, graph, obj, node2id=None):
        if node2id is not None:
            self.node2id = node2id
        elif hasattr(graph, "_pygrank_node2id"):          # signals.py:44-45: outcome of a preprocessor
            self.node2id = graph._pygrank_node2id
        elif hasattr(graph, "shape"):                     # signals.py:46-47
            self.node2id = _IdentityMap(graph.shape[0])
        else:
            self.node2id = {v: i for i, v in enumerate(graph)}
        self.graph = graph
        graph_len = graph.shape[0] if hasattr(graph, "shape") else len(graph)
        self._host = None
        if backend.is_array(obj):                         # signals.py:54-58
            if graph_len != backend.length(obj):
                raise Exception("Graph signal array dimensions " + str(backend.length(obj)) +
                                " should be equal to graph nodes " + str(graph_len))
            self._np = backend.to_array(obj)
        elif obj is None:                                 # signals.py:59-60
            self._np = backend.repeat(1.0, graph_len)
        else:                                             # signals.py:61-66: stage on the host, upload once
            staging = np.zeros(graph_len, dtype=np.float64)
            for key, value in obj.items():
                staging[self.node2id[key]] = float(value)
            self._np = backend.to_array(staging)

    # ---- backend primitive
    @property
    def np(self):                                         # signals.py:81-83
        return backend.to_array(self._np)

    @np.setter
    def np(self, value):                                  # signals.py:85-87
        self._np = backend.to_array(self._compliant(value))
        self._host = None

    def filter(self, exclude=None):                       # signals.py:68-75
        if exclude is not None:
            exclude = to_signal(self, exclude)
            return backend.filter_out(self._np, exclude._np)
        return self._np

    def __rshift__(self, other):
        return other(self)

    # ---- mapping protocol
    def _mirror(self):
        if self._host is None:
            self._host = np.asarray(self._np, dtype=np.float64)
        return self._host

    def __getitem__(self, key):
        return float(self._mirror()[self.node2id[key]])

    def __setitem__(self, key, value):
        self._np[self.node2id[key]] = float(value)
        self._host = None

    def __delitem__(self, key):
        self._np[self.node2id[key]] = 0
        self._host = None

    def __iter__(self):
        return iter(self.node2id)

    def __len__(self):
        return len(self.node2id)

    def __str__(self):
        return "{" + ", ".join(repr(k) + ": " + str(v) for k, v in self.items()) + "}"

    # ---- arithmetic (signals.py:114-178)
    def _compliant(self, other):
        if isinstance(other, GraphSignal):
            if id(other.graph) != id(self.graph):
                raise Exception("Can not operate between graph signals of different graphs")
            return other.np
        return other

    def _new(self, value):
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
        if copy:
            return GraphSignal(self.graph, backend.copy(self._np), self.node2id).normalized(normalize, copy=False)
        if normalize:
            self._np = backend.self_normalize(self._np)
            self._host = None
        return self


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
    """signals.py:196-249: callable ranking algorithms that transform graph signals."""

    def __call__(self, graph=None, personalization=None, *args, **kwargs):
        return self.rank(graph, personalization, *args, **kwargs)

    def __or__(self, data):
        if not isinstance(data, GraphSignal):
            raise Exception("Can only apply signals into rankers (use pygrank.to_signal(graph, data)) to create those)")
        return self(data)

    def __rshift__(self, other):
        other.__lshift__(self)
        return other

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        raise Exception("NodeRanking subclasses should implement a rank method")

    def propagate(self, graph, features, *args, **kwargs):   # signals.py:225-226
        return backend.combine_cols([self.rank(graph, col, *args, **kwargs)._np for col in backend.separate_cols(features)])

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
    """signals.py:281-319."""
    if obj is None and graph is None:
        raise Exception("Cannot create signal from two None arguments")
    known_node2id = None
    if obj is None and isinstance(graph, GraphSignal):
        obj, graph = graph, obj
    if graph is None:
        if isinstance(obj, GraphSignal):
            graph = obj.graph
        else:
            raise Exception("None graph allowed only for explicit graph signal input")
    elif isinstance(graph, GraphSignal):
        known_node2id = graph.node2id
        graph = graph.graph
    elif backend.is_array(graph):
        raise Exception("Graph cannot be an array")
    if isinstance(obj, list) and len(obj) != len(graph):  # signals.py:313-314: a short list is a seed list
        obj = {v: 1 for v in obj}
    if isinstance(obj, GraphSignal):
        if id(graph) != id(obj.graph):
            raise Exception("Graph signal tied to a different graph")
        return obj
    return GraphSignal(graph, obj, known_node2id)
