"""pygrank_amd -- MI355X-native engine for pygrank's graph-filter propagation path.

Drop-in for the hot path of MKLab-ITI/pygrank: the backend-module contract
(``pygrank_amd.backend.hip``: the 29 functions of pygrank/core/backend/specification.py), the
signal / preprocessor API around it (``to_signal``, ``preprocessor``, ``AdjacencyWrapper``), the convergence
manager with its residual measures, and the graph filters that drive the loop (``PageRank``, ``HeatKernel``,
``GenericGraphFilter``, ``AbsorbingWalks`` ...).  All arithmetic runs in hand-written HIP kernels for gfx950
behind the C-ABI of include/pgh.h; importing this package does not touch the GPU, the first backend call does
(and raises if no MI355X or no built engine library is present -- there is no CPU fallback).

Usage mirrors the reference::

    import pygrank_amd as pg
    ranks = pg.PageRank(alpha=0.85, tol=1e-6, error_type=pg.L1)(pg.AdjacencyWrapper(A, directed=True), seeds)
"""
from pygrank_amd import backend
from pygrank_amd.backend import Backend, load_backend, safe_div, safe_inv, backend_name
from pygrank_amd.backend import (graph_dropout, separate_cols, combine_cols, abs, sum, mean, min, max, exp, log, ones,
                                 eye, diag, copy, scipy_sparse_to_backend, to_array, to_primitive, cast, is_array,
                                 repeat, self_normalize, conv, length, degrees, dot, filter_out, epsilon, backend_init)
from pygrank_amd.signals import GraphSignal, NodeRanking, to_signal
from pygrank_amd.preprocessing import (Adjacency, AdjacencyWrapper, MethodHasher, obj2id, preprocessor,
                                       to_sparse_matrix)
from pygrank_amd.utils import call, ensure_used_args, remove_used_args
from pygrank_amd.measures import (AUC, L1, L2, MSQ, MSQRT, Cos, Dot, Euclidean, Mabs, MaxDifference, RMabs, Supervised)
from pygrank_amd.convergence import ConvergenceManager
from pygrank_amd.postprocess import (LinearSweep, Normalize, Ordinals, Postprocessor, Sweep, Tautology, Threshold, Top,
                                     Transformer)
from pygrank_amd.filters import (AbsorbingWalks, ClosedFormGraphFilter, GenericGraphFilter, GraphFilter, HeatKernel,
                                 ImpulseGraphFilter, LowPassRecursiveGraphFilter, PageRank, PageRankClosed,
                                 RecursiveGraphFilter, SymmetricAbsorbingRandomWalks)
from pygrank_amd.device import DeviceGraph, DeviceMatrix, DeviceVector

__version__ = "0.1.0"
