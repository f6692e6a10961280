"""Differentiable ``propagate`` for APPNP-style use (pygrank tests/test_gnn.py:22-28: a ranker's ``propagate`` inside a
model's forward pass; the reference gets its gradients from running the filter on a tensorflow / pytorch backend).

This backend's loops are not an autograd tape, but the filters a GNN propagates with are LINEAR maps of the feature matrix -- a fixed number
of steps (``error_type="iters"``), no L1 quotient: Y = F X with F = sum_k c_k (M^T)^k -- so the gradient of a loss with respect to the
features is the SAME filter run on the transposed operator: dL/dX = F^T dL/dY, F^T = sum_k c_k M^k.  ``differentiable_propagate`` is a
``torch.autograd.Function`` around ``ranker.propagate`` (the engine's multi-seed loop: pgh_ppr_run_batch for PageRank) whose backward pass
is one more ``propagate`` on the transposed graph (the same graph when the normalised matrix is symmetric; otherwise built once per graph
from the stored M^T).  ``graph_dropout`` is not supported here (a different mask per step would have to be replayed in reverse order)."""
import numpy as np

from pygrank_amd import backend
from pygrank_amd.device import DeviceGraph, DeviceMatrix
from pygrank_amd.preprocessing import Adjacency


def _is_linear(ranker):
    """A fixed number of steps and no L1 quotient (RecursiveGraphFilter.use_quotient, abstract_filters.py:133-134): then rank() is linear in
    the personalization -- the L1 normalisation of GraphFilter.rank (:52-55) and preserve_norm (:63-64) cancel."""
    counts_only = getattr(ranker.convergence, "_counts_only", lambda: False)()
    quotient = getattr(ranker, "use_quotient", False)
    return counts_only and (quotient is False or quotient == 0) and getattr(ranker, "preserve_norm", True) \
        and not getattr(ranker, "converge_to_eigenvectors", False)


def transposed_operator(ranker, graph):
    """The preprocessed graph whose conv multiplies by M instead of M^T (cached on the preprocessed graph); the graph itself when M = M^T."""
    M = ranker.preprocessor(graph)
    cached = getattr(M, "_pgh_transposed", None)
    if cached is not None:
        return cached
    g = getattr(M, "array", M)
    if not isinstance(g, DeviceGraph):
        raise Exception("differentiable_propagate needs a graph of the hip backend")
    MT = g.download_transposed()                              # scipy CSR of the stored M^T (f32 values)
    diff = MT - MT.T
    if diff.nnz == 0 or abs(diff).max() == 0:
        out = M
    else:
        out = Adjacency(DeviceGraph.from_scipy(MT))           # stores (M^T)^T = M: its conv is x -> M x
        out._pygrank_preprocessed = {backend.backend_name(): out}
        out._pygrank_node2id = M._pygrank_node2id
        out.is_directed = getattr(M, "is_directed", lambda: True)
    M._pgh_transposed = out
    return out


def _run(ranker, graph, array):
    F = DeviceMatrix.from_host(np.ascontiguousarray(array, dtype=np.float64))
    out = ranker.propagate(graph, F)
    if not isinstance(out, DeviceMatrix):                     # the per-column fallback of NodeRanking.propagate
        out = backend.combine_cols([getattr(col, "np", col) for col in out])
    return np.asarray(out, dtype=np.float64)


def differentiable_propagate(ranker, graph, features):
    """ranker.propagate(graph, features) as a differentiable torch operation.  features: torch tensor [n, B] (any device / float dtype);
    returns a tensor of the same device and dtype.  The ranker must be linear in its personalization: error_type="iters", use_quotient=False
    (what the reference's APPNP example configures, tests/test_gnn.py:24-25)."""
    import torch
    if not _is_linear(ranker):
        raise Exception("differentiable_propagate: the ranker must run a fixed number of steps without the L1 quotient "
                        "(error_type='iters', use_quotient=False): only then is propagate a linear map with a transposed-operator gradient")

    class _Propagate(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            y = _run(ranker, graph, x.detach().cpu().numpy())
            return torch.as_tensor(y, dtype=x.dtype, device=x.device)

        @staticmethod
        def backward(ctx, grad_out):
            gx = _run(ranker, transposed_operator(ranker, graph), grad_out.detach().cpu().numpy())
            return torch.as_tensor(gx, dtype=grad_out.dtype, device=grad_out.device)

    return _Propagate.apply(features)
