"""Row-partitioned personalized PageRank across the GPUs of one node (BASELINE.json configs[4]).

The propagation path shards with ONE exchange step per iteration.  Vertices are relabelled by descending source
count and dealt round-robin to the ranks (pgh_graph_rmat_part), so every rank owns an equal-sized, statistically
nnz-balanced, hot-first slice of the rows of M^T.  Per iteration a rank runs the fused PageRank step on its slice
(pgh_ppr_step_dist: block partials -> combine with the alpha / (1 - alpha) epilogue), which also produces its slice
of the next gather vector; the slices are all-gathered (RCCL over xGMI through torch.distributed -- one process per
GPU, the engine enqueues on torch's current stream) and two f64 scalars are all-reduced (sum(y) for the L1 quotient,
the residual for ConvergenceManager).  The reference has no distributed counterpart ("none"); the loop
semantics are those of GraphFilter.rank + RecursiveGraphFilter._step + ConvergenceManager
(pygrank/algorithms/filters/abstract_filters.py:44-65,126-136; pygrank/algorithms/convergence.py:77-101).
"""
import ctypes as C
import math
import os
import time

import numpy as np

from pygrank_amd import _lib as L
from pygrank_amd.device import DeviceGraph, DeviceVector

_NORMALIZATIONS = {"col": 0, "symmetric": 1, "none": 2}


class PartitionedGraph:
    """This rank's slice of a globally relabelled RMAT graph."""

    def __init__(self, graph, rank, world):
        self.graph, self.rank, self.world = graph, rank, world
        self.n = graph.shape[0]                 # global number of nodes (length of the gather vector)
        self.n_local = graph.shape[1]           # rows of M^T held here
        self.row_begin = rank * self.n_local
        self._perm = None

    @property
    def perm(self):
        """new id -> original id (identical on every rank)."""
        if self._perm is None:
            out = np.empty(self.n, dtype=np.int32)
            rb = C.c_int64()
            L.check(L.lib().pgh_graph_perm(self.graph._h, out.ctypes.data_as(C.c_void_p), C.byref(rb)))
            self._perm = out
        return self._perm


def rmat_partitioned(scale, edge_factor, rank, world, a=0.57, b=0.19, c=0.19, seed=0, normalization="col", symmetrize=False):
    L.ensure_init()
    h = L.c_graph()
    L.check(L.lib().pgh_graph_rmat_part(int(scale), int(edge_factor), float(a), float(b), float(c), int(seed),
                                        _NORMALIZATIONS[normalization], 1 if symmetrize else 0, int(rank), int(world), C.byref(h)))
    vals = [C.c_int64() for _ in range(4)]
    L.check(L.lib().pgh_graph_info(h, *[C.byref(v) for v in vals]))
    return PartitionedGraph(DeviceGraph(h, (vals[0].value, vals[1].value), vals[2].value), rank, world)


class _Buffers:
    """torch tensors (device memory for RCCL, host memory for gloo) viewed as engine vectors."""

    def __init__(self, n, n_local, device):
        import torch
        self.torch = torch
        self.xg_full = torch.zeros(n, dtype=torch.float32, device=device)
        self.xg_local = torch.zeros(n_local, dtype=torch.float32, device=device)
        self.y = [torch.zeros(n_local, dtype=torch.float32, device=device) for _ in range(2)]
        self.scalar = torch.zeros(1, dtype=torch.float64, device=device)
        self.v_xg_full = DeviceVector.wrap(self.xg_full.data_ptr(), n, keepalive=self.xg_full)
        self.v_xg_local = DeviceVector.wrap(self.xg_local.data_ptr(), n_local, keepalive=self.xg_local)
        self.v_y = [DeviceVector.wrap(t.data_ptr(), n_local, keepalive=t) for t in self.y]


class DistributedPageRank:
    """PageRank(alpha) with ConvergenceManager(tol, error_type, max_iters, end_modulo) on a PartitionedGraph."""

    _KINDS = {"mabs": L.ERR_MABS, "l1": L.ERR_L1, "linf": L.ERR_LINF, "iters": L.ERR_ITERS}

    def __init__(self, alpha=0.85, tol=1e-6, error_type="mabs", max_iters=100, end_modulo=1, use_quotient=True,
                 preserve_norm=True, epsilon=float(np.finfo(np.float32).eps)):
        self.alpha, self.tol, self.error_type = alpha, tol, error_type
        self.max_iters, self.end_modulo, self.use_quotient = max_iters, end_modulo, use_quotient
        self.preserve_norm, self.epsilon = preserve_norm, epsilon
        self.iteration, self.spmv, self.elapsed = 0, 0, None
        self._buffers = None

    def _all_reduce(self, bufs, dist, value, op):
        bufs.scalar[0] = value
        dist.all_reduce(bufs.scalar, op=op)
        return float(bufs.scalar.item())

    def rank(self, pgraph, p_local):
        """p_local: this rank's slice of the personalization (DeviceVector, new id space).  Returns the slice of ranks."""
        import torch
        import torch.distributed as dist
        lib = L.lib()
        g = pgraph.graph
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        if self._buffers is None or self._buffers.xg_full.numel() != pgraph.n:
            self._buffers = _Buffers(pgraph.n, pgraph.n_local, device)
        bufs = self._buffers
        kind = self._KINDS[self.error_type]
        tol = 0.0 if self.tol is None else max(self.tol, self.epsilon)          # convergence.py:101
        # One stream for everything: the engine launches on it (pgh_set_stream) and torch.distributed orders its
        # RCCL collectives against it, so kernels and collectives are sequenced on the device without host syncs.
        if device.type == "cuda":
            if getattr(self, "_stream", None) is None:
                self._stream = torch.cuda.Stream()
            L.check(lib.pgh_set_stream(C.c_void_p(self._stream.cuda_stream)))
            stream_ctx = torch.cuda.stream(self._stream)
        else:
            import contextlib
            stream_ctx = contextlib.nullcontext()
        with stream_ctx:
            # ---- prologue of GraphFilter.rank (abstract_filters.py:52-56): global L1 norm, x0 = p / norm
            norm = self._all_reduce(bufs, dist, p_local.abssum(), dist.ReduceOp.SUM)
            if norm == 0:
                self.iteration = 0
                return p_local
            p = p_local / norm
            cur, scale, prev_scale = 0, 1.0, 1.0
            L.check(lib.pgh_vec_copy(bufs.v_y[cur]._h, p._h))
            L.check(lib.pgh_dist_prescale(g._h, bufs.v_y[cur]._h, bufs.v_xg_local._h))
            dist.all_gather_into_tensor(bufs.xg_full, bufs.xg_local)
            t0 = time.perf_counter()
            it, spmv, converged = 1, 0, False       # `it` = ConvergenceManager.iteration of the pending has_converged call
            err, s_local = C.c_double(), C.c_double()
            while it < self.max_iters:                                             # convergence.py:86
                nxt = 1 - cur
                L.check(lib.pgh_ppr_step_dist(g._h, bufs.v_xg_full._h, scale, p._h, self.alpha, bufs.v_y[nxt]._h,
                                              bufs.v_xg_local._h, C.byref(s_local)))
                total = self._all_reduce(bufs, dist, s_local.value, dist.ReduceOp.SUM)
                dist.all_gather_into_tensor(bufs.xg_full, bufs.xg_local)           # next gather vector over xGMI ...
                prev_scale, scale = scale, ((1.0 / total if total != 0 else 0.0) if self.use_quotient else 1.0)
                cur = nxt
                spmv += 1
                it += 1
                if it >= self.max_iters:
                    break
                if kind != L.ERR_ITERS and it % self.end_modulo == 0:              # ... overlapped with the residual
                    local_kind = L.ERR_LINF if kind == L.ERR_LINF else L.ERR_L1
                    L.check(lib.pgh_scaled_residual(local_kind, bufs.v_y[cur]._h, scale, bufs.v_y[1 - cur]._h, prev_scale,
                                                    C.byref(err)))
                    e = self._all_reduce(bufs, dist, err.value, dist.ReduceOp.MAX if kind == L.ERR_LINF else dist.ReduceOp.SUM)
                    if kind == L.ERR_MABS:
                        e /= pgraph.n
                    if e <= tol:
                        converged = True
                        break
            if device.type == "cuda":
                self._stream.synchronize()
            self.elapsed = time.perf_counter() - t0
            self.iteration, self.spmv, self.converged = it, spmv, converged
            if not converged and self.error_type != "iters" and it >= self.max_iters:
                raise Exception("Could not converge within " + str(self.max_iters) + " iterations")
            factor = scale * (norm if self.preserve_norm else 1.0)                 # abstract_filters.py:63-64
            out = bufs.v_y[cur] * factor
            L.check(lib.pgh_sync())
            return out


# ----------------------------------------------------------------------------------------------------------------
# bench.py --gpus N leg
# ----------------------------------------------------------------------------------------------------------------
def bench_row_partitioned(args, rmat, alpha, tol, max_iters, num_seeds, hbm_peak):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank)
    L.ensure_init(local_rank if use_cuda else 0)
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl" if use_cuda else "gloo")
    if args.scale is None:                      # weak scaling: fixed edges per GPU; configs[4] at 8 GPUs
        scale, ef = (27, 8) if world == 8 else (23 + int(math.log2(world)), 16)
    else:
        scale, ef = args.scale, (args.ef or 16)
    if args.ef is not None and args.scale is None:
        ef = args.ef
    t0 = time.time()
    pg = rmat_partitioned(scale, ef, rank, world, **rmat)
    L.check(L.lib().pgh_sync())
    build_s = time.time() - t0
    n, n_local, lo = pg.n, pg.n_local, pg.row_begin
    nnz_t = torch.tensor([pg.graph.nnz], dtype=torch.float64, device="cuda" if use_cuda else "cpu")
    dist.all_reduce(nnz_t)
    nnz_total = int(nnz_t.item())
    deg = np.asarray(pg.graph.degrees())                     # row sums of M for every (relabelled) source
    candidates = np.flatnonzero(deg > 0)
    total = args.warmup + args.steps
    personalizations = []
    for step in range(total):
        rng = np.random.default_rng(1 + step)
        seeds = np.sort(rng.choice(candidates, size=min(num_seeds, len(candidates)), replace=False))
        p = np.zeros(n_local)
        mine = seeds[(seeds >= lo) & (seeds < lo + n_local)] - lo
        p[mine] = 1.0
        personalizations.append(DeviceVector.from_host(p))
    ranker = DistributedPageRank(alpha=alpha, tol=tol, error_type="l1", max_iters=max_iters)
    for step in range(args.warmup):
        ranker.rank(pg, personalizations[step])
    dist.barrier()
    if use_cuda:
        torch.cuda.synchronize()
    spmv_total, iters = 0, []
    t0 = time.perf_counter()
    for step in range(args.warmup, total):
        ranker.rank(pg, personalizations[step])
        spmv_total += ranker.spmv
        iters.append(ranker.iteration)
    if use_cuda:
        torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if use_cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # roofline leg on rank 0: HIP-event time of the step kernels, per-GPU algorithmic bytes
    lib = L.lib()
    L.check(lib.pgh_profile_reset())
    L.check(lib.pgh_profile_enable(1))
    ranker.rank(pg, personalizations[total - 1])
    L.check(lib.pgh_profile_enable(0))
    prof = {}
    for kid, name in ((L.K_SPMV, "spmv"), (L.K_FIXUP, "fixup"), (L.K_COMBINE, "combine"), (L.K_RESIDUAL, "residual")):
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
        prof[name] = (ms.value / cnt.value * 1e3) if cnt.value else None
    step_us = sum(v for k, v in prof.items() if k in ("spmv", "fixup", "combine") and v)
    alg_bytes = 8 * pg.graph.nnz + 4 * n + 16 * n_local
    achieved = alg_bytes / (step_us * 1e-6) / 1e9 if step_us else None
    if rank != 0:
        return None
    return dict(
        metric="edges*iters/sec (GTEPS) for PPR alpha=0.85 to tol=1e-6", value=round(nnz_total * spmv_total / elapsed / 1e9, 2),
        unit="GTEPS", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload=f"row-partitioned PPR on RMAT scale-{scale} ef-{ef} over {world} GPUs (BASELINE.json configs[4] shape)",
                    n=n, nnz=nnz_total, alpha=alpha, tol=tol, error_type="L1", seeds=num_seeds, iterations_per_step=iters,
                    spmv_per_step=spmv_total / args.steps, graph_build_s=round(build_s, 2),
                    parallelism=f"1-D row partition x{world}, all-gather of the gather vector + 2 scalar all-reduces per iteration",
                    exchange_bytes_per_iteration_per_gpu=4 * n_local * (world - 1)),
        roofline=dict(bound="hbm", kernel="k_bsf_partial + k_bsf_fixup + k_bsf_combine<AXPBY> (one fused PPR step, rank 0 slice)",
                      achieved=round(achieved, 1) if achieved else None, peak=hbm_peak, unit="GB/s",
                      frac=round(achieved / hbm_peak, 4) if achieved else None, traffic=None,
                      algorithmic_bytes_per_launch=alg_bytes, avg_launch_us=round(step_us, 2), format=pg.graph.format(),
                      kernels_avg_us=prof),
        cpu_baseline=None, parity=None)
