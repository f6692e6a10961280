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

    __slots__ = ("_h", "_n", "_keepalive", "uuid", "__weakref__")
    __array_priority__ = 1000      # numpy scalars/arrays defer to our reflected operators

    def __init__(self, handle, n, keepalive=None):
        self._h = handle
        self._n = int(n)
        self._keepalive = keepalive

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
            out = DeviceVector.empty(self._n)
            a, b = (other, self) if reflected else (self, other)
            L.check(L.lib().pgh_ewise_vv(op, a._h, b._h, out._h))
            return out
        if isinstance(other, (numbers.Number, np.generic)) or (isinstance(other, np.ndarray) and other.ndim == 0):
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
    def from_adjacency(W, normalization):
        """The preprocessor's "col" / "symmetric" / "both" / "none" normalisation of the raw adjacency W evaluated on the
        device (include/pgh.h pgh_graph_from_adjacency); unit weights travel as structure only."""
        import scipy.sparse as sp
        L.ensure_init()
        kind = {"col": L.NORM_COL, "symmetric": L.NORM_SYMMETRIC, "none": L.NORM_NONE, "both": L.NORM_BOTH}[normalization]
        W = sp.csr_array(W) if not sp.issparse(W) or W.format != "csr" else W
        indptr = np.ascontiguousarray(W.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(W.indices, dtype=np.int32)
        data = np.ascontiguousarray(W.data, dtype=np.float64)
        unit = len(data) > 0 and bool(np.all(data == 1.0))
        h = L.c_graph()
        L.check(L.lib().pgh_graph_from_adjacency(W.shape[0], W.shape[1], len(data), _ptr(indptr), _ptr(indices),
                                                 None if unit else _ptr(data), kind, 0, C.byref(h)))
        return DeviceGraph(h, W.shape, len(data))

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
