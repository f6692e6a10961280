"""The backend contract of the propagation path: the 29 functions every engine module must define.

Mirror of pygrank/core/backend/specification.py:5-118 (names only -- that file holds empty stubs)."""

API = (
    "backend_name", "backend_init", "graph_dropout", "separate_cols", "combine_cols", "abs", "sum", "mean", "min",
    "max", "exp", "log", "ones", "eye", "diag", "copy", "scipy_sparse_to_backend", "to_array", "to_primitive", "cast",
    "is_array", "repeat", "self_normalize", "conv", "length", "degrees", "dot", "filter_out", "epsilon",
)
assert len(API) == 29
