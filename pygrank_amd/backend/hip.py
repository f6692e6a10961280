"""The "hip" backend module: pygrank's 29-function backend contract on the MI355X engine.

Drop-in counterpart of pygrank/core/backend/numpy.py:1-86 / pytorch.py:1-114 for an HBM-resident fp32
engine; a pygrank maintainer would copy this file to ``pygrank/core/backend/hip.py`` and add "hip" to the
allow-list of pygrank/core/backend/__init__.py:41.  Every function launches hand-written HIP
kernels through the C-ABI of include/pgh.h; nothing here computes on the CPU except the two krylov-only dense
helpers ``eye``/``diag`` which the numpy and matvec engines of the reference also keep on the host
(numpy.py:3, matvec.py:41-42).
"""
import numpy as _np

from pygrank_amd import _lib as _L
from pygrank_amd.device import DroppedGraph, DeviceGraph, DeviceMatrix, DeviceVector


def backend_name():                       # specification.py:5-6
    return "hip"


def backend_init():                       # specification.py:9-10; fails loudly without an MI355X
    _L.ensure_init()


_dropout_calls = [0x5EED]                  # seed of the next dropout mask; set_dropout_seed() makes runs reproducible


def set_dropout_seed(seed):
    _dropout_calls[0] = int(seed)


def take_dropout_seeds(count):
    """Reserves `count` consecutive mask seeds (a fused batch run draws one per step) and returns the first."""
    first = _dropout_calls[0] + 1
    _dropout_calls[0] += int(count)
    return first


def peek_dropout_seed():
    """The seed the next graph_dropout call would draw (nothing is reserved)."""
    return _dropout_calls[0] + 1


def graph_dropout(M, dropout):            # specification.py:13; identity and O(1) for dropout == 0 (called 2 + #steps times)
    if dropout == 0:
        return M
    # pytorch.py:34-38: a new Bernoulli mask on the edge values at every call.  Here the mask is a hash of (seed, entry)
    # evaluated inside the SpMV (pgh_spmv_dropout): a dropped graph is a view, nothing is copied.
    base = getattr(M, "array", M)
    if isinstance(base, DroppedGraph):
        base = base.base
    _dropout_calls[0] += 1
    return DroppedGraph(base, float(dropout), _dropout_calls[0])


def separate_cols(x):                     # specification.py:17
    if isinstance(x, DeviceMatrix):
        return x.columns()
    x = _np.asarray(x)
    return [DeviceVector.from_host(x[:, j]) for j in range(x.shape[1])]


def combine_cols(cols):                   # specification.py:21
    return DeviceMatrix.from_columns(cols)


def _vec(x):
    return x if isinstance(x, (DeviceVector, DeviceMatrix)) else to_array(x)


def abs(x):                               # specification.py:25
    return _vec(x).__abs__()


def sum(x, axis=None):                    # specification.py:29
    if isinstance(x, DeviceGraph):
        if axis == 1:
            return x.degrees()
        raise NotImplementedError("only row sums (axis=1) of a device graph are available")
    if axis is not None:
        raise NotImplementedError("axis reductions are not part of the propagation path")
    return _vec(x).sum()


def mean(x, axis=None):                   # specification.py:33
    if axis is not None:
        raise NotImplementedError("axis reductions are not part of the propagation path")
    return _vec(x).mean()


def min(x, axis=None):                    # specification.py:37
    if axis is not None:
        raise NotImplementedError("axis reductions are not part of the propagation path")
    return _vec(x).min()


def max(x, axis=None):                    # specification.py:41
    if axis is not None:
        raise NotImplementedError("axis reductions are not part of the propagation path")
    return _vec(x).max()


def exp(x):                               # specification.py:45
    return _vec(x)._unary(_L.EXP)


def log(x):                               # specification.py:49
    return _vec(x)._unary(_L.LOG)


def ones(dims):                           # specification.py:53
    if isinstance(dims, int):
        return DeviceVector.full(dims, 1.0)
    if len(dims) == 1:
        return DeviceVector.full(dims[0], 1.0)
    return DeviceMatrix.from_columns([DeviceVector.full(dims[0], 1.0) for _ in range(dims[1])])


def eye(dims):                            # specification.py:57; host sparse identity as in numpy.py:3
    from scipy.sparse import eye as _eye
    return _eye(dims)


def diag(diagonal, offset=0):             # specification.py:61; krylov-only dense helper (host, as matvec.py:41-42)
    return _np.diag(_np.asarray(diagonal, dtype=_np.float64), offset)


def copy(x):                              # specification.py:65
    return x.copy() if isinstance(x, DeviceVector) else to_array(x, copy_array=True)


def scipy_sparse_to_backend(M):           # specification.py:69; preprocessing.py:144
    return DeviceGraph.from_scipy(M)


def to_array(obj, copy_array=False):      # specification.py:73; identity for own type (tests/test_core.py:30-32)
    if isinstance(obj, DeviceVector):
        return obj.copy() if copy_array else obj
    if isinstance(obj, DeviceMatrix):
        if obj.b == 1:
            return obj.column(0)
        raise _L.EngineError("cannot flatten an [n, b] slab with b > 1 into a vector")
    if obj.__class__.__module__ == "torch":
        obj = obj.detach().cpu().numpy()
    elif obj.__class__.__module__ == "tensorflow.python.framework.ops":
        obj = obj.numpy()
    return DeviceVector.from_host(_np.asarray(obj, dtype=_np.float64))


def to_primitive(obj):                    # specification.py:77
    if isinstance(obj, (DeviceVector, DeviceMatrix)):
        return obj
    if isinstance(obj, (float, int)):
        return float(obj)
    arr = _np.asarray(obj, dtype=_np.float64)
    if arr.ndim == 2:
        return DeviceMatrix.from_host(arr)
    return DeviceVector.from_host(arr)


def cast(obj):                            # specification.py:81; masks are already 0/1 f32 vectors
    return obj


def is_array(obj):                        # specification.py:85
    return isinstance(obj, (list, _np.ndarray, DeviceVector)) or obj.__class__.__module__ == "torch" \
        or obj.__class__.__module__ == "tensorflow.python.framework.ops"


def repeat(value, times):                 # specification.py:89
    return DeviceVector.full(times, value)


def self_normalize(obj):                  # specification.py:93; numpy.py:57-61
    s = obj.abssum()
    return obj / s if s != 0 else obj


def conv(signal, M):                      # specification.py:97; numpy.py:64-65: signal @ M = M^T signal
    return M.conv(_vec(signal))             # an [n, b] slab is propagated in one multi-seed pass


def length(x):                            # specification.py:101
    if isinstance(x, DeviceMatrix):
        return x.n * x.b
    return len(x)


def degrees(M):                           # specification.py:105; numpy.py:76-77
    return M.degrees()


def dot(x, y):                            # specification.py:109
    return _vec(x).dot(_vec(y))


def filter_out(x, exclude):               # specification.py:113
    return _vec(x).filter_out(_vec(exclude))


def epsilon():                            # specification.py:117; fp32 engine precedent pytorch.py:113-114
    return float(_np.finfo(_np.float32).eps)
