"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the pygrank graph-filter propagation path.

Nothing in ``pygrank_amd`` (the product) may import, call, link or execute anything under
``oracle/``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the CPU baseline.

Contents
--------
ref_loops.py   numpy/scipy fp64 restatement of the reference loops (PageRank, closed-form
               filters, AbsorbingWalks, convergence manager, normalisation).  Parity PINNED:
               checked against golden vectors produced by importing the reference itself
               (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
rmat_np.py     numpy restatement of the device RMAT generator (integer-exact, same hash).
spmv_oracle.c  plain-C restatement of scipy's csc_matvec (the arithmetic behind
               ``numpy.py:64-65`` ``signal @ M``) + an OpenMP pull variant for the all-core
               CPU baseline.  Built into oracle/_build/ by oracle/Makefile.
host_abi.c     host restatement of the include/pgh.h C-ABI (f32 storage, f64 accumulate) used
               as a test double so the host-side Python and the gloo multi-rank path can be
               exercised without a GPU.  Never loaded by the product.
"""
