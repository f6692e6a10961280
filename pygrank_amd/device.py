"""HBM-resident primitives of the engine: DeviceVector, DeviceMatrix, DeviceGraph.

These are the "BackendPrimitive" / "BackendGraph" types of the hip backend (reference typing:
pygrank/core/typing.py:6-7).  DeviceVector implements the operator protocol pygrank's filters and
GraphSignal rely on (pygrank/core/signals.py:89-96,114-178): ``+ - * / **`` with
scalars and vectors, unary ``-``, comparisons producing 0/1 masks, mask indexing, ``float(x[i])`` and item
assignment.  Every operation is one HIP kernel launched through the C-ABI (include/pgh.h); ownership is
RAII-style through ``__del__`` because the backend API has no explicit free (SURVEY.md 8b "Ownership").
"""
import ctypes as C
import numbers

import numpy as np

from pygrank_amd import _lib as L


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p)


class DeviceVector:
    """Dense f32 vector in HBM."""

    __slots__ = ("_h", "_n", "_keepalive", "uuid", "_deps", "__weakref__")
    __array_priority__ = 1000      # numpy scalars/arrays defer to our reflected operators

    def __init__(self, handle, n, keepalive=None):
        self._h = handle
        self._n = int(n)
        self._keepalive = keepalive
        self._deps = None

    def _before_write(self):
        """Called by everything that overwrites this vector's memory in place: lazy expressions that still read it are evaluated first
        (they stand for the values the vector held when they were formed)."""
        deps, self._deps = self._deps, None
        if deps:
            for ref in deps:
                dep = ref()
                if dep is not None:
                    dep._h                  # noqa: B018 -- evaluates the expression into memory of its own
        for cache in _RESIDENT_CACHES:
            cache.pop(id(self), None)

    def _depend(self, lazy):
        """`lazy` reads this vector's memory when it is evaluated (weak references in a list: vectors compare elementwise and do not hash)."""
        import weakref
        if self._deps is None:
            self._deps = []
        elif len(self._deps) >= 64:
            self._deps = [ref for ref in self._deps if ref() is not None]
        self._deps.append(weakref.ref(lazy))

    # ------------------------------------------------------------------ construction
    @staticmethod
    def empty(n):
        L.ensure_init()
        h = L.c_vec()
        L.check(L.lib().pgh_vec_alloc(int(n), C.byref(h)))
        return DeviceVector(h, n)

    @staticmethod
    def full(n, value):
        v = DeviceVector.empty(n)
        L.check(L.lib().pgh_vec_fill(v._h, float(value)))
        return v

    @staticmethod
    def from_host(obj):
        arr = np.asarray(obj)
        if arr.ndim > 1:
            arr = arr.squeeze()          # numpy.py:41-43 flattens (n, 1)
        if arr.ndim == 0:
            arr = arr.reshape(1)
        if arr.ndim != 1:
            raise L.EngineError("DeviceVector needs one-dimensional data")
        v = DeviceVector.empty(arr.shape[0])
        if arr.dtype == np.float32:
            arr = np.ascontiguousarray(arr)
            L.check(L.lib().pgh_vec_h2d_f32(v._h, _ptr(arr), arr.shape[0]))
        else:
            arr = np.ascontiguousarray(arr, dtype=np.float64)
            L.check(L.lib().pgh_vec_h2d_f64(v._h, _ptr(arr), arr.shape[0]))
        return v

    @staticmethod
    def wrap(device_ptr, n, keepalive=None):
        """Non-owning view of externally allocated device memory (e.g. a torch tensor used for RCCL)."""
        L.ensure_init()
        h = L.c_vec()
        L.check(L.lib().pgh_vec_wrap(C.c_void_p(device_ptr), int(n), C.byref(h)))
        return DeviceVector(h, n, keepalive)

    def __del__(self):
        try:
            if self._h is not None and L._lib is not None:
                L._lib.pgh_vec_free(self._h)
        except Exception:
            pass
        self._h = None

    # ------------------------------------------------------------------ host transfer
    def numpy(self, dtype=np.float64):
        out = np.empty(self._n, dtype=dtype)
        if self._n:
            if dtype == np.float32:
                L.check(L.lib().pgh_vec_d2h_f32(self._h, _ptr(out), self._n))
            else:
                L.check(L.lib().pgh_vec_d2h_f64(self._h, _ptr(out), self._n))
        return out

    def __array__(self, dtype=None, copy=None):
        out = self.numpy()
        return out if dtype is None else out.astype(dtype)

    def tolist(self):
        return self.numpy().tolist()

    def __iter__(self):
        return iter(self.numpy())

    def __len__(self):
        return self._n

    @property
    def shape(self):
        return (self._n,)

    @property
    def ptr(self):
        return L.lib().pgh_vec_ptr(self._h)

    def copy(self):
        out = DeviceVector.empty(self._n)
        L.check(L.lib().pgh_vec_copy(out._h, self._h))
        return out

    def __repr__(self):
        head = self.numpy()[:6] if self._n else []
        return f"DeviceVector(n={self._n}, head={np.array2string(np.asarray(head), precision=6)})"

    def __bool__(self):
        if self._n == 1:
            return bool(self[0])
        raise ValueError("The truth value of a DeviceVector with more than one element is ambiguous")

    # ------------------------------------------------------------------ elementwise plumbing
    def _binary(self, op, other, reflected=False):
        if isinstance(other, DeviceVector):
            if other._n != self._n:
                raise L.EngineError(f"operands have different lengths {self._n} vs {other._n}")
            if LAZY and op in (L.MUL, L.ADD) and self._n > 0 and self._keepalive is None and other._keepalive is None \
                    and _kind(self) in _VV_OPERANDS and _kind(other) in _VV_OPERANDS:
                # p * lam and lam + deg of AbsorbingWalks._formula (adhoc.py:166-169) are formed again in every step: unevaluated, the
                # walk's formula is recognised whole (LazyVector "walk") and neither of them is ever computed
                return LazyVector.elementwise(op, (other, self) if reflected else (self, other))
            out = DeviceVector.empty(self._n)
            a, b = (other, self) if reflected else (self, other)
            L.check(L.lib().pgh_ewise_vv(op, a._h, b._h, out._h))
            return out
        if _is_scalar(other):
            if LAZY and self._keepalive is None and self._n > 0:
                # x * s, s * x, x / s: an unevaluated product (LazyVector) -- a * conv(x, M) + b * p then reaches the engine as ONE step
                factor = _scalar_factor(op, float(other), reflected)
                if factor is not None:
                    return LazyVector.scaled(self, factor)
            out = DeviceVector.empty(self._n)
            L.check(L.lib().pgh_ewise_vs(op, self._h, float(other), 1 if reflected else 0, out._h))
            return out
        if isinstance(other, (list, tuple, np.ndarray)):
            return self._binary(op, DeviceVector.from_host(other), reflected)
        return NotImplemented

    def _unary(self, op):
        out = DeviceVector.empty(self._n)
        L.check(L.lib().pgh_ewise_unary(op, self._h, out._h))
        return out

    def __add__(self, o): return self._binary(L.ADD, o)
    def __radd__(self, o): return self._binary(L.ADD, o, True)
    def __sub__(self, o): return self._binary(L.SUB, o)
    def __rsub__(self, o): return self._binary(L.SUB, o, True)
    def __mul__(self, o): return self._binary(L.MUL, o)
    def __rmul__(self, o): return self._binary(L.MUL, o, True)
    def __truediv__(self, o): return self._binary(L.DIV, o)
    def __rtruediv__(self, o): return self._binary(L.DIV, o, True)
    def __pow__(self, o): return self._binary(L.POW, o)
    def __rpow__(self, o): return self._binary(L.POW, o, True)
    def __gt__(self, o): return self._binary(L.GT, o)
    def __ge__(self, o): return self._binary(L.GE, o)
    def __lt__(self, o): return self._binary(L.LT, o)
    def __le__(self, o): return self._binary(L.LE, o)
    def __eq__(self, o): return self._binary(L.EQ, o)
    def __ne__(self, o): return self._binary(L.NE, o)
    __hash__ = None
    def __neg__(self): return self._unary(L.NEG)
    def __pos__(self): return self
    def __abs__(self): return self._unary(L.ABS)

    # ------------------------------------------------------------------ reductions
    def _reduce(self, kind):
        out = C.c_double()
        L.check(L.lib().pgh_reduce(kind, self._h, C.byref(out)))
        return out.value

    def sum(self): return self._reduce(L.SUM)
    def abssum(self): return self._reduce(L.ABSSUM)
    def max(self): return self._reduce(L.MAX)
    def min(self): return self._reduce(L.MIN)

    def mean(self):
        return self.sum() / self._n

    def ordinals(self):
        """1 for the largest entry, 2 for the next ... (ties: lower index first); one device radix sort."""
        out = DeviceVector.empty(self._n)
        L.check(L.lib().pgh_vec_ordinals(self._h, out._h))
        return out

    def kth_largest(self, k):
        out = C.c_double()
        L.check(L.lib().pgh_vec_kth_largest(self._h, int(k), C.byref(out)))
        return out.value

    def gap_threshold(self):
        """Threshold("gap") (postprocess.py:328-343): the score after the first largest relative drop of the descending order."""
        out = C.c_double()
        L.check(L.lib().pgh_vec_gap_threshold(self._h, C.byref(out)))
        return out.value

    def dot(self, other):
        out = C.c_double()
        L.check(L.lib().pgh_dot(self._h, other._h, C.byref(out)))
        return out.value

    # ------------------------------------------------------------------ indexing
    def __getitem__(self, key):
        if isinstance(key, DeviceVector):                # x[mask]: keep where mask != 0 (numpy semantics)
            exclude = key._binary(L.EQ, 0.0)
            return self.filter_out(exclude)
        if isinstance(key, (int, np.integer)):
            i = int(key)
            if i < 0:
                i += self._n
            out = C.c_double()
            L.check(L.lib().pgh_vec_get(self._h, i, C.byref(out)))
            return out.value
        if isinstance(key, slice):
            start, stop, step = key.indices(self._n)
            if step != 1:
                return DeviceVector.from_host(self.numpy()[key])
            count = max(stop - start, 0)
            view = DeviceVector.wrap((self.ptr or 0) + 4 * start, count, keepalive=self)
            return view.copy()
        if isinstance(key, (list, np.ndarray)):
            key = np.asarray(key)
            if key.dtype == bool:
                return self[DeviceVector.from_host(key.astype(np.float64))]
            return DeviceVector.from_host(self.numpy()[key])
        raise TypeError("unsupported index " + repr(type(key)))

    def __setitem__(self, key, value):
        self._before_write()
        i = int(key)
        if i < 0:
            i += self._n
        L.check(L.lib().pgh_vec_set(self._h, i, float(value)))

    def filter_out(self, exclude):
        """x[exclude == 0] (specification.py:113)."""
        staging = DeviceVector.empty(self._n)
        count = C.c_int64()
        L.check(L.lib().pgh_filter_out(self._h, exclude._h, staging._h, C.byref(count)))
        return staging[0:count.value] if count.value != self._n else staging

    def axpby(self, a, other, b):
        """a * self + b * other in one pass."""
        out = DeviceVector.empty(self._n)
        L.check(L.lib().pgh_axpby(float(a), self._h, float(b), other._h, out._h))
        return out


# ======================================================================================================================
# Lazy vectors: the backend-primitive route without a round trip through the caller's id space per primitive
# ======================================================================================================================
LAZY = True                    # False: every primitive is evaluated where it stands (one engine call each; A/B measurements, tests)
_RESIDENT_CACHES = []          # the per-graph caches of resident copies of plain vectors (DeviceVector._before_write drops its entry)
_RESIDENT_KINDS = ("res", "conv", "scaled", "axpby", "lin", "walk", "cmul", "cmul_add")
_MAX_DEPTH = 6
_VV_OPERANDS = (None, "mat", "plain")  # operands of an unevaluated elementwise product / sum in the caller's ids
_MEMORY_KINDS = (None, "mat")     # values held in the caller's ids: a plain DeviceVector, an expression already evaluated there


def _is_scalar(other):
    kind = type(other)
    return kind is float or kind is int or isinstance(other, (numbers.Number, np.generic)) or (kind is np.ndarray and other.ndim == 0)


def _scalar_factor(op, value, reflected):
    """x (op) value as a multiplication, or None."""
    if op == L.MUL:
        return value
    if op == L.DIV and not reflected and value != 0.0 and np.isfinite(value):
        return 1.0 / value
    return None


def _kind(vec):
    return vec._kind if isinstance(vec, LazyVector) else None


class _Resident:
    """One vector in the id space of `graph`: y [n_int] and (made on demand) its gather form; sum(y) when a step produced it."""
    __slots__ = ("graph", "y", "xg", "sum")

    def __init__(self, graph, y, xg=None, total=None):
        self.graph, self.y, self.xg, self.sum = graph, y, xg, total

    def gather_form(self):
        if self.xg is None and self.graph._n_gather:
            self.xg = DeviceVector.empty(self.graph._n_gather)
            L.check(L.lib().pgh_resident_gather(self.graph._h, self.y._h, self.xg._h))
        return self.xg


class LazyVector(DeviceVector):
    """A DeviceVector whose value is an expression the engine has not evaluated yet (the reference's filters reach a
    backend ONE PRIMITIVE AT A TIME -- pygrank/core/backend/__init__.py:59-80; PageRank._formula adhoc.py:34-36 is
    ``conv(ranks, M) * alpha + personalization * (1 - alpha)``, RecursiveGraphFilter._step abstract_filters.py:126-136 follows it
    with ``ranks / sum(ranks)``, ConvergenceManager convergence.py:96-101 with ``sum(abs(ranks - previous)) / length``).

    Evaluated primitive by primitive each of those is a pass over an n-vector, and every conv is a round trip through the engine's
    relabelled id space (pgh_spmv: way in, three launches, way out, a synchronisation).  Here a primitive returns an expression:

      kind        value                                  becomes
      "res"       scale * R         (R resident)         --
      "conv"      a * M^T src                            "res" by pgh_resident_step(mode 0)
      "scaled"    scale * parent    (parent lazy)        "res" sharing the parent's memory
      "axpby"     a * (node) + b * p, node = M^T src     "res" by ONE pgh_resident_step(mode 1), sum(y) included
      "lin"       sa * u + sb * v, optionally |.|        "res" by pgh_axpby in the id space; sum / max of |u - v| is ONE
                                                         pgh_scaled_residual on the resident operands, nothing is written
      "walk"      the absorbing walk's formula            "res" by ONE pgh_resident_step(mode 2): (a M^T src * deg + p * lam) / (lam + deg),
                  (adhoc.py:166-169), recognised piece by piece: conv * deg ("cmul"), + p * lam ("cmul_add"), / (lam + deg)
      "vv"        u * v or u + v    (caller ids)         memory of its own (one pgh_ewise_vv) when looked at -> "mat"
      "plain"     b * p             (p in caller ids)    memory of its own (one pgh_ewise_vs) when looked at -> "mat"
      "mat"       evaluated in the caller's ids          --

    and the way out of the id space (pgh_resident_out) runs when somebody looks at the values in the caller's ids: ``_h`` of a lazy
    vector is a property that evaluates, so every engine call and every DeviceVector method works on it unchanged.  An expression reads
    its operands when it is evaluated; DeviceVector._before_write evaluates the pending readers of a vector before the vector is
    overwritten in place, so the value is always the one of the moment the expression was formed.  Resident memory is never written
    in place."""

    # (no __slots__ here: the fields of an expression live in the instance dictionary over these class-level defaults, so creating
    # one -- five per PageRank step -- stores three attributes, not seventeen)
    _kind = _graph = _res = _src = _p = _u = _v = _mat = _op = _d = _lam = None
    _scale = _a = _b = _sa = _sb = 1.0
    _abs = False
    _depth = 0                      # unevaluated expressions below this one; a chain is evaluated before it grows past _MAX_DEPTH

    def __init__(self, n):
        self._n = n
        self._keepalive = self._deps = None

    # ---- constructors -----------------------------------------------------------------------------------------------
    @staticmethod
    def scaled(vec, factor):
        out = LazyVector(vec._n)
        k = _kind(vec)
        if k == "res":
            out._kind, out._graph, out._res, out._scale = "res", vec._graph, vec._res, vec._scale * factor
        elif k == "scaled":
            out._kind, out._graph, out._src, out._scale, out._depth = "scaled", vec._graph, vec._src, vec._scale * factor, vec._depth
        elif k in ("conv", "axpby", "lin", "walk", "cmul", "cmul_add"):      # evaluated once, shared by every view of it
            out._kind, out._graph, out._src, out._scale, out._depth = "scaled", vec._graph, vec, factor, vec._depth
        elif k == "plain":
            out._kind, out._p, out._b = "plain", vec._p, vec._b * factor
            vec._p._depend(out)
        else:                                              # a plain vector, or an expression already evaluated in the caller's ids
            out._kind, out._p, out._b = "plain", vec, factor
            vec._depend(out)
        return out

    @staticmethod
    def conv(graph, x):
        out = LazyVector(graph.shape[1])
        out._kind, out._graph, out._src, out._a = "conv", graph, x, 1.0
        if _kind(x) not in _RESIDENT_KINDS:
            x._depend(out)
        return out._bounded(getattr(x, "_depth", 0) + 1)

    @staticmethod
    def elementwise(op, pair):
        out = LazyVector(pair[0]._n)
        out._kind, out._op, out._u, out._v = "vv", op, pair[0], pair[1]
        for vec in pair:
            (vec._p if _kind(vec) == "plain" else vec)._depend(out)
        return out

    def _walk_stage(self, op, other):
        """The next piece of AbsorbingWalks._formula, or None: conv * deg -> "cmul"; cmul + p * lam -> "cmul_add"; cmul_add / (lam + deg) ->
        "walk" when the two lam and the two deg are the same objects (the filter's attributes)."""
        k = self._kind
        if op == L.MUL and _kind(other) in _MEMORY_KINDS and (k == "conv" or (k == "scaled" and _kind(self._src) == "conv")):
            node, coeff = (self, 1.0) if k == "conv" else (self._src, self._scale)
            out = LazyVector(self._n)
            out._kind, out._graph, out._src, out._a, out._d = "cmul", self._graph, node, coeff, other
            other._depend(out)
            return out._bounded(node._depth + 1)
        if op == L.ADD and k == "cmul" and _kind(other) == "vv" and other._op == L.MUL:
            out = LazyVector(self._n)
            out._kind, out._graph, out._src, out._a, out._d, out._p = "cmul_add", self._graph, self._src, self._a, self._d, other
            return out._bounded(self._depth)
        if op == L.DIV and k == "cmul_add" and _kind(other) == "vv" and other._op == L.ADD:
            q = self._p
            for lam, deg in ((other._u, other._v), (other._v, other._u)):
                if deg is not self._d:
                    continue
                p = q._v if q._u is lam else (q._u if q._v is lam else None)
                if p is None or _kind(lam) not in _VV_OPERANDS or _kind(p) not in _VV_OPERANDS:
                    continue
                out = LazyVector(self._n)
                out._kind, out._graph, out._src, out._a, out._d, out._lam, out._p = "walk", self._graph, self._src, self._a, deg, lam, p
                for vec in (lam, p):
                    (vec._p if _kind(vec) == "plain" else vec)._depend(out)
                deg._depend(out)
                return out._bounded(self._depth)
        return None

    def _bounded(self, depth):
        """A filter that never looks at its iterate (error_type="iters", no quotient) would nest one expression per step: the chain is
        evaluated where it passes _MAX_DEPTH."""
        if depth > _MAX_DEPTH:
            self._flush()
            depth = 0
        self._depth = depth
        return self

    # ---- evaluation -------------------------------------------------------------------------------------------------
    @staticmethod
    def _resident_of(graph, vec):
        """(resident form, scale) of `vec` in the id space of `graph`: an expression of that graph is flushed, a vector in the caller's
        ids is brought in (and remembered while it lives unwritten: the personalization of a run comes back every step)."""
        k = _kind(vec)
        if k in _RESIDENT_KINDS and vec._graph is graph:
            vec._flush()
            return vec._res, vec._scale
        if k == "plain" and _kind(vec._p) in _MEMORY_KINDS:
            return graph._resident_copy(vec._p), vec._b
        return graph._resident_copy(vec), 1.0

    def _flush(self):
        """Turns this expression into a resident vector ("res")."""
        k = self._kind
        if k == "res":
            return
        self._depth = 0
        g = self._graph
        lib = L.lib()
        if k == "scaled":
            parent = self._src
            parent._flush()
            self._kind, self._res, self._scale, self._src = "res", parent._res, parent._scale * self._scale, None
            return
        if k in ("conv", "axpby"):
            node = self if k == "conv" else self._src
            if k == "axpby" and (node._kind != "conv" or _shared(node)):
                # the product M^T src is somebody else's too (ClosedFormGraphFilter keeps it as its next power): evaluated once, kept
                # there; this expression is then arithmetic in the id space
                conv_res, conv_scale = self._resident_of(g, node)
                p_res, p_scale = self._resident_of(g, self._p)
                y = DeviceVector.empty(g._n_int)
                L.check(lib.pgh_axpby(conv_scale * self._a, conv_res.y._h, p_scale * self._b, p_res.y._h, y._h))
                self._kind, self._res, self._scale, self._src, self._p = "res", _Resident(g, y), 1.0, None, None
                return
            src_res, src_scale = self._resident_of(g, node._src)
            y = DeviceVector.empty(g._n_int)
            yg = DeviceVector.empty(g._n_gather) if g._n_gather else None
            xg = src_res.gather_form()
            if k == "conv":
                L.check(lib.pgh_resident_step(g._h, 0, src_res.y._h, xg._h if xg is not None else None, node._a * src_scale, None, 0.0,
                                              None, None, y._h, yg._h if yg is not None else None, None))
                total = None
            else:
                p_res, p_scale = self._resident_of(g, self._p)
                got = C.c_double()
                L.check(lib.pgh_resident_step(g._h, 1, src_res.y._h, xg._h if xg is not None else None, self._a * node._a * src_scale,
                                              p_res.y._h, self._b * p_scale, None, None, y._h, yg._h if yg is not None else None, C.byref(got)))
                total = got.value
            self._kind, self._res, self._scale, self._src, self._p = "res", _Resident(g, y, yg, total), 1.0, None, None
            return
        if k in ("cmul", "cmul_add"):
            # a piece of the walk's formula that somebody looked at before it was whole: the product in the id space (padding: 0 * 0)
            node = self._src
            conv_res, conv_scale = self._resident_of(g, node)
            d_res = g._resident_copy(self._d)
            y = DeviceVector.empty(g._n_int)
            L.check(lib.pgh_ewise_vv(L.MUL, conv_res.y._h, d_res.y._h, y._h))
            scale = conv_scale * self._a
            if k == "cmul_add":
                q_res, q_scale = self._resident_of(g, self._p)
                z = DeviceVector.empty(g._n_int)
                L.check(lib.pgh_axpby(scale, y._h, q_scale, q_res.y._h, z._h))
                y, scale = z, 1.0
            self._kind, self._res, self._scale, self._src, self._d, self._p = "res", _Resident(g, y), scale, None, None, None
            return
        if k == "walk":
            node = self._src
            src_res, src_scale = self._resident_of(g, node._src)
            p_mem = self._p
            lam_mem = self._lam
            for vec in (p_mem, lam_mem):
                if _kind(vec) == "plain":
                    vec._h                                   # noqa: B018 -- lam = ones * lambda, p = signal / norm: evaluated once, then remembered
            p_res = g._resident_copy(p_mem)
            lam_res = g._resident_copy(lam_mem, hole=1.0)    # padding: (0 * 0 + 0 * 1) / (1 + 0)
            d_res = g._resident_copy(self._d)
            y = DeviceVector.empty(g._n_int)
            yg = DeviceVector.empty(g._n_gather) if g._n_gather else None
            xg = src_res.gather_form()
            got = C.c_double()
            L.check(lib.pgh_resident_step(g._h, 2, src_res.y._h, xg._h if xg is not None else None, self._a * node._a * src_scale, p_res.y._h, 1.0,
                                          d_res.y._h, lam_res.y._h, y._h, yg._h if yg is not None else None, C.byref(got)))
            self._kind, self._res, self._scale = "res", _Resident(g, y, yg, got.value), 1.0
            self._src = self._p = self._d = self._lam = None
            return
        if k == "lin":
            u_res, u_scale = self._resident_of(g, self._u)
            v_res, v_scale = self._resident_of(g, self._v)
            y = DeviceVector.empty(g._n_int)
            L.check(lib.pgh_axpby(self._sa * u_scale, u_res.y._h, self._sb * v_scale, v_res.y._h, y._h))
            if self._abs:
                z = DeviceVector.empty(g._n_int)
                L.check(lib.pgh_ewise_unary(L.ABS, y._h, z._h))
                y = z
            self._kind, self._res, self._scale, self._u, self._v, self._abs = "res", _Resident(g, y), 1.0, None, None, False
            return
        raise L.EngineError("a lazy vector of kind " + repr(k) + " has no resident form")

    @property
    def _h(self):
        """The handle of this vector's values in the CALLER's ids: evaluates the expression (once)."""
        if self._mat is None:
            k = self._kind
            if k == "plain":
                out = DeviceVector.empty(self._n)
                L.check(L.lib().pgh_ewise_vs(L.MUL, self._p._h, float(self._b), 0, out._h))
                self._kind, self._p = "mat", None
            elif k == "vv":
                out = DeviceVector.empty(self._n)
                L.check(L.lib().pgh_ewise_vv(self._op, self._u._h, self._v._h, out._h))
                self._kind, self._u, self._v = "mat", None, None
            elif k in _RESIDENT_KINDS:
                self._flush()
                out = DeviceVector.empty(self._n)
                L.check(L.lib().pgh_resident_out(self._graph._h, self._res.y._h, float(self._scale), out._h))
            else:
                return None
            self._mat = out
        return self._mat._h

    @_h.setter
    def _h(self, value):            # DeviceVector.__init__ assigns None
        if value is not None:
            raise L.EngineError("a lazy vector owns no handle")

    def __del__(self):
        pass                        # whatever this expression holds is owned by plain vectors

    def _writable(self):
        """The caller is about to write into this vector's memory: it becomes an ordinary vector in the caller's ids."""
        self._h                     # noqa: B018
        self._kind, self._graph, self._res = "mat", None, None
        return self

    def __setitem__(self, key, value):
        self._writable()
        DeviceVector.__setitem__(self, key, value)

    def copy(self):
        if self._kind == "res":
            return LazyVector.scaled(self, 1.0)         # resident memory is never written in place: a copy is another view of it
        return DeviceVector.copy(self)

    # ---- arithmetic that stays lazy ---------------------------------------------------------------------------------
    def _binary(self, op, other, reflected=False):
        if isinstance(other, DeviceVector) and other._n == self._n and op in (L.MUL, L.ADD, L.DIV):
            a, b = (other, self) if reflected else (self, other)            # a (op) b
            stage = a._walk_stage(op, b) if isinstance(a, LazyVector) else None
            if stage is None and op != L.DIV and isinstance(b, LazyVector):  # (products and sums commute)
                stage = b._walk_stage(op, a)
            if stage is not None:
                return stage
        if self._kind != "mat":
            if _is_scalar(other):
                factor = _scalar_factor(op, float(other), reflected)
                if factor is not None:
                    return LazyVector.scaled(self, factor)
            elif isinstance(other, DeviceVector) and other._n == self._n and op in (L.ADD, L.SUB):
                sign = -1.0 if op == L.SUB else 1.0
                a, b = (other, self) if reflected else (self, other)
                out = _linear(a, 1.0, b, sign)
                if out is not None:
                    return out
        return DeviceVector._binary(self, op, other, reflected)

    # (a subclass's reflected operator is tried first only when the subclass DEFINES it: plain + lazy comes here)
    def __add__(self, o): return self._binary(L.ADD, o)
    def __radd__(self, o): return self._binary(L.ADD, o, True)
    def __sub__(self, o): return self._binary(L.SUB, o)
    def __rsub__(self, o): return self._binary(L.SUB, o, True)
    def __mul__(self, o): return self._binary(L.MUL, o)
    def __rmul__(self, o): return self._binary(L.MUL, o, True)
    def __truediv__(self, o): return self._binary(L.DIV, o)
    def __neg__(self): return LazyVector.scaled(self, -1.0) if self._kind != "mat" else DeviceVector.__neg__(self)

    def __abs__(self):
        if self._kind == "lin" and not self._abs:
            out = LazyVector(self._n)
            out._kind, out._graph, out._u, out._v, out._sa, out._sb, out._abs = "lin", self._graph, self._u, self._v, self._sa, self._sb, True
            out._depth = self._depth
            return out
        return DeviceVector.__abs__(self)

    # ---- reductions in the id space (padding slots hold zeros) ------------------------------------------------------------
    def _reduce(self, kind):
        if self._kind in _RESIDENT_KINDS:
            if self._kind == "lin" and self._abs and kind in (L.SUM, L.ABSSUM, L.MAX):
                return _residual_of(self._graph, L.ERR_LINF if kind == L.MAX else L.ERR_L1, self._u, self._sa, self._v, self._sb)
            if kind in (L.SUM, L.ABSSUM):
                self._flush()
                res = self._res
                if kind == L.SUM and res.sum is not None:
                    return res.sum * self._scale
                out = C.c_double()
                L.check(L.lib().pgh_reduce(kind, res.y._h, C.byref(out)))
                if kind == L.SUM:
                    res.sum = out.value
                    return out.value * self._scale
                return out.value * abs(self._scale)
        return DeviceVector._reduce(self, kind)


def _references(node):
    import sys
    return sys.getrefcount(node)


def _unshared_references():
    """What _references reports for an object held by ONE attribute and the caller's local (the situation of LazyVector._flush when only
    the expression being evaluated holds its product): measured, not assumed -- the count depends on how the interpreter passes arguments."""
    class Holder:
        pass
    holder = Holder()
    holder.node = Holder()
    node = holder.node
    return _references(node)


_UNSHARED = _unshared_references()


def _shared(node):
    """Does anything but the expression being evaluated hold `node`?  Only the cost depends on the answer: a shared product is evaluated
    on its own and kept (ClosedFormGraphFilter keeps M^T term as its next power), an unshared one is folded into the step."""
    import sys
    return sys.getrefcount(node) > _UNSHARED


def _linear(a, sa, b, sb):
    """sa * a + sb * b as an expression in the id space of the graph one of them lives in, or None (evaluate in the caller's ids)."""
    graph = None
    for vec in (a, b):
        if _kind(vec) in _RESIDENT_KINDS:
            if graph is not None and vec._graph is not graph:
                return None
            graph = vec._graph
    if graph is None:
        return None
    # a * conv(x, M) + b * p: ONE engine step
    for conv, second, sc, weight in ((a, b, sa, sb), (b, a, sb, sa)):
        node, coeff = conv, sc
        if _kind(node) == "scaled" and _kind(node._src) == "conv":
            node, coeff = node._src, coeff * node._scale
        if _kind(node) != "conv":
            continue
        if _kind(second) == "plain" and _kind(second._p) in _MEMORY_KINDS:
            second, weight = second._p, weight * second._b
        if _kind(second) not in (None, "mat", "res"):
            continue
        out = LazyVector(a._n)
        out._kind, out._graph, out._src, out._a, out._p, out._b = "axpby", graph, node, coeff, second, weight
        if _kind(second) != "res":
            second._depend(out)
        return out._bounded(node._depth + 1)
    out = LazyVector(a._n)
    out._kind, out._graph, out._u, out._v, out._sa, out._sb = "lin", graph, a, b, sa, sb
    for vec in (a, b):
        if _kind(vec) not in _RESIDENT_KINDS:
            (vec._p if _kind(vec) == "plain" else vec)._depend(out)
    return out._bounded(max(getattr(a, "_depth", 0), getattr(b, "_depth", 0)) + 1)


def _residual_of(graph, kind, u, su, v, sv):
    """sum |su u + sv v| (L1) or max |.| (LINF) over the resident operands: one pgh_scaled_residual, nothing materialised."""
    u_res, u_scale = LazyVector._resident_of(graph, u)
    v_res, v_scale = LazyVector._resident_of(graph, v)
    out = C.c_double()
    L.check(L.lib().pgh_scaled_residual(kind, u_res.y._h, su * u_scale, v_res.y._h, -sv * v_scale, C.byref(out)))
    return out.value


def lazy_residual(kind, a, b):
    """residual(kind, a, b) of two vectors without leaving the id space one of them lives in (measures.Supervised.evaluate), or None."""
    graph = None
    for vec in (a, b):
        if _kind(vec) in _RESIDENT_KINDS:
            if graph is not None and vec._graph is not graph:
                return None
            graph = vec._graph
    if graph is None or len(a) != len(b) or len(a) != graph.shape[1]:
        return None
    value = _residual_of(graph, L.ERR_LINF if kind == L.ERR_LINF else L.ERR_L1, a, 1.0, b, -1.0)
    return value / len(a) if kind == L.ERR_MABS else value


def _all_ones(data):
    """Are all weights 1 (the graph then travels as structure only)?  Chunk by chunk with an early exit -- a weighted graph is found out
    in its first 4 M entries -- and as min / max: `np.all(data == 1.0)` materialises a boolean per entry (6x the time on 131 M weights)."""
    if len(data) == 0:
        return False
    step = 1 << 22
    for lo in range(0, len(data), step):
        part = data[lo:lo + step]
        if part.min() != 1.0 or part.max() != 1.0:
            return False
    return True


class DeviceMatrix:
    """Row-major f32 [n, b] slab in HBM: the multi-seed batch layout (a gathered row is b*4 contiguous bytes)."""

    def __init__(self, handle, n, b):
        self._h, self.n, self.b = handle, int(n), int(b)

    @staticmethod
    def empty(n, b):
        L.ensure_init()
        h = L.c_mat()
        L.check(L.lib().pgh_mat_alloc(int(n), int(b), C.byref(h)))
        return DeviceMatrix(h, n, b)

    @staticmethod
    def from_host(arr):
        arr = np.ascontiguousarray(np.asarray(arr, dtype=np.float64))
        if arr.ndim != 2:
            raise L.EngineError("DeviceMatrix needs two-dimensional data")
        m = DeviceMatrix.empty(arr.shape[0], arr.shape[1])
        L.check(L.lib().pgh_mat_h2d_f64(m._h, _ptr(arr)))
        return m

    @staticmethod
    def from_columns(cols):
        cols = list(cols)
        m = DeviceMatrix.empty(len(cols[0]), len(cols))
        for j, col in enumerate(cols):
            if not isinstance(col, DeviceVector):
                col = DeviceVector.from_host(col)
            L.check(L.lib().pgh_mat_set_col(m._h, j, col._h))
        return m

    def __del__(self):
        try:
            if self._h is not None and L._lib is not None:
                L._lib.pgh_mat_free(self._h)
        except Exception:
            pass
        self._h = None

    @property
    def shape(self):
        return (self.n, self.b)

    def __len__(self):
        return self.n

    def numpy(self):
        out = np.empty((self.n, self.b), dtype=np.float64)
        if self.n:
            L.check(L.lib().pgh_mat_d2h_f64(self._h, _ptr(out)))
        return out

    def __array__(self, dtype=None, copy=None):
        out = self.numpy()
        return out if dtype is None else out.astype(dtype)

    def column(self, j):
        v = DeviceVector.empty(self.n)
        L.check(L.lib().pgh_mat_get_col(self._h, int(j), v._h))
        return v

    def columns(self):
        return [self.column(j) for j in range(self.b)]

    # ---- whole-slab operations (one kernel each; the per-column forms above cost one strided pass per column)
    def col_abssum(self):
        out = np.zeros(self.b, dtype=np.float64)
        L.check(L.lib().pgh_mat_col_abssum(self._h, _ptr(out)))
        return out

    def div_cols(self, divisors):
        """out[:, j] = self[:, j] / divisors[j]; a zero divisor copies the column."""
        d = np.ascontiguousarray(divisors, dtype=np.float64)
        out = DeviceMatrix.empty(self.n, self.b)
        L.check(L.lib().pgh_mat_div_cols(self._h, _ptr(d), out._h))
        return out

    def gemv(self, coeffs):
        """sum_j self[:, j] * coeffs[j] over the leading len(coeffs) columns (pgh_mat_gemv, f64 accumulation)."""
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        out = DeviceVector.empty(self.n)
        L.check(L.lib().pgh_mat_gemv(self._h, _ptr(c), len(c), out._h))
        return out

    def gemm(self, coeffs, out=None, accumulate=False):
        """self[:, :K] @ coeffs for a [K, P] coefficient matrix (P <= 64, pgh_mat_gemm): [n, P]; with `out` and
        `accumulate` the product is added to an existing slab (slabs of more than 64 powers, chunk by chunk)."""
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        if c.ndim != 2:
            raise Exception("gemm expects a [terms, probes] coefficient matrix")
        if out is None:
            out = DeviceMatrix.empty(self.n, c.shape[1])
            accumulate = False
        L.check(L.lib().pgh_mat_gemm(self._h, _ptr(c), int(c.shape[0]), int(c.shape[1]), 1 if accumulate else 0, out._h))
        return out

    def set_column(self, j, vec):
        L.check(L.lib().pgh_mat_set_col(self._h, int(j), vec._h))

    def get_cols(self, first, count):
        out = DeviceMatrix.empty(self.n, count)
        L.check(L.lib().pgh_mat_get_cols(self._h, int(first), out._h))
        return out

    def set_cols(self, first, src):
        L.check(L.lib().pgh_mat_set_cols(self._h, int(first), src._h))

    def __sub__(self, other):
        return DeviceMatrix.from_columns([a - b for a, b in zip(self.columns(), other.columns())])

    def __abs__(self):
        return DeviceMatrix.from_columns([abs(a) for a in self.columns()])

    def sum(self):
        return float(sum(c.sum() for c in self.columns()))


class DeviceGraph:
    """CSR(M^T) in HBM (f32 values, int32 columns) + merge-path tile table.  Produced by
    scipy_sparse_to_backend (specification.py:70-71); ``shape`` is the shape of the un-transposed M."""

    def __init__(self, handle, shape, nnz):
        self._h = handle
        self.shape = tuple(int(s) for s in shape)
        self.nnz = int(nnz)
        self._n_int = self._n_gather = None         # lengths of a resident iterate and of its gather form (pgh_graph_resident_len)
        self._resident = {}                          # id(plain vector) -> (weak reference, its resident copy)

    def _resident_lengths(self):
        if self._n_int is None:
            a, b = C.c_int64(), C.c_int64()
            L.check(L.lib().pgh_graph_resident_len(self._h, C.byref(a), C.byref(b)))
            self._n_int, self._n_gather = a.value, b.value
            if a.value:
                _RESIDENT_CACHES.append(self._resident)
        return self._n_int

    def _resident_copy(self, vec, hole=0.0):
        """`vec` (caller's ids) in this graph's id space; remembered while the vector lives unwritten (a run's personalization is an
        operand of every step; so are the degrees and the absorption of a walk, whose padding slots take `hole`).  At most six copies are
        kept."""
        import weakref
        hit = self._resident.get(id(vec))
        if hit is not None and hit[0]() is vec and hit[2] == hole:
            return hit[1]
        y = DeviceVector.empty(self._n_int)
        xg = DeviceVector.empty(self._n_gather) if (self._n_gather and hole == 0.0) else None
        L.check(L.lib().pgh_resident_in(self._h, vec._h, float(hole), y._h, xg._h if xg is not None else None))
        res = _Resident(self, y, xg)
        if _kind(vec) in _MEMORY_KINDS and vec._keepalive is None:
            while len(self._resident) >= 6:
                self._resident.pop(next(iter(self._resident)))
            self._resident[id(vec)] = (weakref.ref(vec), res, hole)
        return res

    @staticmethod
    def from_scipy(M):
        import scipy.sparse as sp
        L.ensure_init()
        factors = getattr(M, "_pgh_factors", None)
        if factors is not None:
            W, left, right = factors
            if W.format == "csr" and W.nnz == M.nnz and np.array_equal(W.indptr, M.indptr) and np.array_equal(W.indices, M.indices):
                return DeviceGraph.from_factored(W, left, right)
        M = sp.csr_array(M) if not sp.issparse(M) or M.format != "csr" else M
        indptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(M.indices, dtype=np.int32)
        data = np.ascontiguousarray(M.data, dtype=np.float64)
        h = L.c_graph()
        L.check(L.lib().pgh_graph_from_csr(M.shape[0], M.shape[1], len(data), _ptr(indptr), _ptr(indices), _ptr(data),
                                           0, C.byref(h)))
        return DeviceGraph(h, M.shape, len(data))

    @staticmethod
    def from_adjacency(W, normalization, renormalize=0.0):
        """The preprocessor's "col" / "symmetric" / "both" / "none" / "laplacian" normalisation of the raw adjacency W -- after the
        renormalisation trick W + renormalize * I when asked for (preprocessing.py:107-108) -- evaluated on the device (include/pgh.h
        pgh_graph_from_adjacency / _ex); unit weights travel as structure only."""
        import scipy.sparse as sp
        L.ensure_init()
        kind = {"col": L.NORM_COL, "symmetric": L.NORM_SYMMETRIC, "none": L.NORM_NONE, "both": L.NORM_BOTH,
                "laplacian": L.NORM_LAPLACIAN}[normalization]
        W = sp.csr_array(W) if not sp.issparse(W) or W.format != "csr" else W
        indptr = np.ascontiguousarray(W.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(W.indices, dtype=np.int32)
        data = np.ascontiguousarray(W.data, dtype=np.float64)
        unit = _all_ones(data)
        h = L.c_graph()
        if renormalize == 0 and kind != L.NORM_LAPLACIAN:
            L.check(L.lib().pgh_graph_from_adjacency(W.shape[0], W.shape[1], len(data), _ptr(indptr), _ptr(indices),
                                                     None if unit else _ptr(data), kind, 0, C.byref(h)))
            return DeviceGraph(h, W.shape, len(data))
        L.check(L.lib().pgh_graph_from_adjacency_ex(W.shape[0], W.shape[1], len(data), _ptr(indptr), _ptr(indices),
                                                    None if unit else _ptr(data), kind, float(renormalize), 0, C.byref(h)))
        vals = [C.c_int64() for _ in range(4)]
        L.check(L.lib().pgh_graph_info(h, *[C.byref(v) for v in vals]))
        return DeviceGraph(h, W.shape, vals[2].value)           # every row gained its diagonal entries

    @staticmethod
    def from_factored(W, left=None, right=None):
        """M = diag(left) W diag(right) evaluated on the device (include/pgh.h pgh_graph_from_factored_csr)."""
        import scipy.sparse as sp
        L.ensure_init()
        W = sp.csr_array(W) if not sp.issparse(W) or W.format != "csr" else W
        indptr = np.ascontiguousarray(W.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(W.indices, dtype=np.int32)
        data = np.ascontiguousarray(W.data, dtype=np.float64)
        left = None if left is None else np.ascontiguousarray(left, dtype=np.float64)
        right = None if right is None else np.ascontiguousarray(right, dtype=np.float64)
        if (left is not None and len(left) != W.shape[0]) or (right is not None and len(right) != W.shape[1]):
            raise L.EngineError("from_factored: scale vector length does not match the matrix")
        h = L.c_graph()
        L.check(L.lib().pgh_graph_from_factored_csr(W.shape[0], W.shape[1], len(data), _ptr(indptr), _ptr(indices), _ptr(data),
                                                    None if left is None else _ptr(left), None if right is None else _ptr(right),
                                                    0, C.byref(h)))
        return DeviceGraph(h, W.shape, len(data))

    def destroy(self):
        """Releases the device images now (a partitioned bench frees its slice before it builds the whole graph)."""
        self.__del__()

    def __del__(self):
        try:
            if self._h is not None and L._lib is not None:
                L._lib.pgh_graph_destroy(self._h)
        except Exception:
            pass
        self._h = None
        try:
            self._resident.clear()
            _RESIDENT_CACHES[:] = [c for c in _RESIDENT_CACHES if c is not self._resident]
        except Exception:
            pass

    def __len__(self):
        return self.shape[0]

    def info(self):
        vals = [C.c_int64() for _ in range(4)]
        L.check(L.lib().pgh_graph_info(self._h, *[C.byref(v) for v in vals]))
        return dict(n_rows=vals[0].value, n_cols=vals[1].value, nnz=vals[2].value, device_bytes=vals[3].value)

    def format(self):
        buf = C.create_string_buffer(512)
        L.check(L.lib().pgh_graph_format(self._h, buf, 512))
        return buf.value.decode()

    def degrees(self):
        out = DeviceVector.empty(self.shape[0])
        L.check(L.lib().pgh_graph_degrees(self._h, out._h))
        return out

    def download_transposed(self):
        """scipy CSR of the stored M^T (verification)."""
        import scipy.sparse as sp
        indptr = np.empty(self.shape[1] + 1, dtype=np.int64)
        indices = np.empty(self.nnz, dtype=np.int32)
        data = np.empty(self.nnz, dtype=np.float32)
        L.check(L.lib().pgh_graph_download(self._h, _ptr(indptr), _ptr(indices), _ptr(data)))
        return sp.csr_array((data, indices, indptr), shape=(self.shape[1], self.shape[0]))

    def conv(self, x):
        """M^T x (numpy.py:64-65); pure.  A DeviceMatrix [n, b] is propagated as one multi-seed pass (b <= 64)."""
        if isinstance(x, DeviceMatrix):
            if x.b > 64:
                return DeviceMatrix.from_columns([self.conv(c) for c in x.columns()])
            y = DeviceMatrix.empty(self.shape[1], x.b)
            L.check(L.lib().pgh_spmm(self._h, x._h, y._h))
            return y
        if LAZY and self._h is not None and len(x) == self.shape[0] and self._resident_lengths():
            return LazyVector.conv(self, x)         # evaluated when somebody looks -- fused with what the filter does to it next
        y = DeviceVector.empty(self.shape[1])
        L.check(L.lib().pgh_spmv(self._h, x._h, y._h))
        return y

    def tocoo(self):
        return self.download_transposed().T.tocoo()


class DroppedGraph:
    """graph_dropout(M, rate) with rate > 0 (pygrank/core/backend/pytorch.py:34-38): the same graph with every entry kept
    with probability 1 - rate and scaled by 1 / (1 - rate).  The mask is a pure function of (seed, entry index) that the
    SpMV kernel evaluates on the fly (include/pgh.h pgh_spmv_dropout)."""

    def __init__(self, base, rate, seed):
        if not 0 <= rate < 1:
            raise L.EngineError("graph_dropout: the rate must lie in [0, 1)")
        self.base, self.rate, self.seed = base, rate, seed
        self.shape = base.shape

    def conv(self, x):
        if isinstance(x, DeviceMatrix):
            if self.shape[0] != self.shape[1] or x.b > 64:
                return DeviceMatrix.from_columns([self.conv(c) for c in x.columns()])
            out = DeviceMatrix.empty(self.shape[1], x.b)                 # one pass over the adjacency for the whole slab
            L.check(L.lib().pgh_spmm_dropout(self.base._h, x._h, out._h, self.rate, self.seed))
            return out
        y = DeviceVector.empty(self.shape[1])
        L.check(L.lib().pgh_spmv_dropout(self.base._h, x._h, y._h, self.rate, self.seed))
        return y

    def degrees(self):
        """Row sums of the dropped M (pgh_graph_degrees_dropout): the same mask the convolutions apply."""
        out = DeviceVector.empty(self.shape[0])
        L.check(L.lib().pgh_graph_degrees_dropout(self.base._h, float(self.rate), int(self.seed), out._h))
        return out

    def format(self):
        return f"dropout {self.rate} (seed {self.seed}) over " + self.base.format()
