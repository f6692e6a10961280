"""Worker of tests/test_distributed_cpu.py: one process per rank (gloo), engine = host test double.
Runs DistributedPageRank on a row-partitioned RMAT graph and writes this rank's slice to a .npz."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, scale, ef = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch.distributed as dist
    import pygrank_amd as pg
    from pygrank_amd import _lib
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import DistributedPageRank, rmat_partitioned
    on_gpu = os.environ.get("PGH_TEST_ENGINE") == "hip"
    if on_gpu:            # tests/test_gpu_parity.py: the real engine; RCCL with one rank, or (PGH_DIST_BACKEND=gloo)
        import torch      # several ranks sharing the single GPU of the box: the multi-rank DEVICE path, functionally
        device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(device)
        _lib.ensure_init(device)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import host_double
        host_double.install()
    # ONE rank's slice dense (its cold image numbers every live slot) beside compact peers -- the switch is read when the
    # slice is built, so it is set in this rank's process only
    if os.environ.get("PGH_TEST_DENSE_RANK") == os.environ.get("RANK", "0"):
        os.environ["PGH_DIST_NEED_LISTS"] = "0"
    pg.load_backend("hip")
    dist.init_process_group(backend=os.environ.get("PGH_DIST_BACKEND", "nccl" if on_gpu else "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    graph = rmat_partitioned(scale, ef, rank, world, seed=0)
    perm = graph.perm
    # personalization defined on ORIGINAL ids (same on every rank), mapped into this rank's slice of new ids
    rng = np.random.default_rng(1)
    p_old = np.zeros(graph.n)
    p_old[rng.choice(graph.n, 20, replace=False)] = rng.random(20) + 0.5
    lo = graph.row_begin
    p_local = p_old[perm[lo:lo + graph.n_local]]
    results = {}
    for name, kw in (("l1", dict(error_type="l1", tol=1e-6, max_iters=500)),
                     ("mabs", dict(error_type="mabs", tol=1e-7, max_iters=500)),
                     ("iters", dict(error_type="iters", max_iters=21)),
                     ("noquot", dict(error_type="linf", tol=1e-7, max_iters=500, use_quotient=False))):
        ranker = DistributedPageRank(alpha=0.85, **kw)
        out = ranker.rank(graph, DeviceVector.from_host(p_local))
        results[name + "_ranks"] = np.asarray(out)
        results[name + "_iters"] = ranker.iteration
        results[name + "_fused"] = int(bool(ranker.exchange.get("in_kernel_residual", False)))
        results[name + "_two_launches"] = int(bool(ranker.exchange.get("finish_in_two_launches", False)))
    # a personalization with NEGATIVE entries: the in-kernel residual cannot vouch for its bound and hands the step to the separate
    # kernel (paused once), the result is the oracle's all the same
    signs = np.where(np.arange(graph.n) % 3 == 0, -0.25, 1.0)
    ranker = DistributedPageRank(alpha=0.85, error_type="l1", tol=1e-6, max_iters=500)
    out = ranker.rank(graph, DeviceVector.from_host((p_old * signs)[perm[lo:lo + graph.n_local]]))
    results["signed_ranks"], results["signed_iters"] = np.asarray(out), ranker.iteration
    results["signed_paused"] = int(bool(ranker.exchange.get("paused_in_kernel_residual", False)))
    results["driver"] = str(ranker.exchange.get("driver"))
    results["split_regions"] = int(bool(ranker.exchange.get("split_regions")))
    # how the cold parts travelled, and what the run says it received per iteration against the host-side count: the hot prefixes of
    # the peers' blocks + the cold slots of the peers' blocks that THIS slice references (pgh_dist_need_counts)
    import ctypes as C
    need = np.zeros(8, dtype=np.int64)
    _lib.check(_lib.lib().pgh_dist_need_counts(graph.graph._h, need.ctypes.data_as(C.c_void_p)))
    nb, blk, hs = C.c_int32(), C.c_int64(), C.c_int32()
    _lib.check(_lib.lib().pgh_graph_gather_layout(graph.graph._h, C.byref(nb), C.byref(blk), None))
    _lib.check(_lib.lib().pgh_graph_hot_prefix(graph.graph._h, C.byref(hs)))
    bpr = nb.value // world
    results["exchange_kind"] = str(ranker.exchange.get("exchange"))
    results["exchange_bytes"] = int(ranker.exchange.get("exchange_bytes_per_iteration_per_gpu", -1))
    results["expected_list_bytes"] = int(4 * (hs.value * bpr * (world - 1) + need[:nb.value].sum() - need[rank * bpr:(rank + 1) * bpr].sum()))
    results["need_total"] = int(need[:nb.value].sum())
    from pygrank_amd.distributed import DistributedAbsorbingWalks
    absorbing = DistributedAbsorbingWalks(alpha=0.85, error_type="l1", tol=1e-6, max_iters=500)
    out = absorbing.rank(graph, DeviceVector.from_host(p_local))
    results["absorb_ranks"] = np.asarray(out)
    results["absorb_iters"] = absorbing.iteration
    from pygrank_amd.distributed import DistributedHeatKernel, DistributedPageRankClosed
    for name, algo in (("heat", DistributedHeatKernel(t=3, error_type="l1", tol=1e-7, max_iters=100)),
                       ("heat_mabs", DistributedHeatKernel(t=5, error_type="mabs", tol=1e-9, max_iters=100)),
                       ("closed", DistributedPageRankClosed(alpha=0.85, error_type="linf", tol=1e-5, max_iters=300))):
        out = algo.rank(graph, DeviceVector.from_host(p_local))
        results[name + "_ranks"] = np.asarray(out)
        results[name + "_iters"] = algo.iteration
    results["closed_form_driver"] = str(algo.exchange.get("driver"))
    results["closed_form_two_launches"] = int(bool(algo.exchange.get("finish_in_two_launches", False)))
    from pygrank_amd.distributed import PREFLIGHT
    results["preflight"] = str(PREFLIGHT.get((world, rank), "not run"))
    # the probe itself (it runs by itself only with more than one rank over RCCL): here with whatever ranks this run has
    results["preflight_selftest"] = "not run"
    if on_gpu and dist.get_backend() == "nccl" and os.environ.get("PGH_DIST_NATIVE", "auto") in ("auto", "1"):
        import torch
        from pygrank_amd import distributed as D
        device = torch.device("cuda", torch.cuda.current_device())
        if D._native_comm(dist, device) is not None:
            results["preflight_selftest"] = str(D._preflight(dist, device, rank, world, _lib.lib()))
    # the gather bases are state of the GRAPH: a Python-driven filter that keeps its buffers must find its own layout again after
    # an engine-driven run on the same graph has laid the gather vector out in two regions
    staged = DistributedHeatKernel(t=3, error_type="l1", tol=1e-7, max_iters=100)
    staged._native_formula = False
    first = np.asarray(staged.rank(graph, DeviceVector.from_host(p_local)))
    DistributedPageRank(alpha=0.85, error_type="l1", tol=1e-6, max_iters=500).rank(graph, DeviceVector.from_host(p_local))
    again = np.asarray(staged.rank(graph, DeviceVector.from_host(p_local)))
    results["interleaved_equal"] = int(np.array_equal(first, again))
    results["interleaved_vs_engine"] = float(np.max(np.abs(first - results["heat_ranks"]), initial=0.0))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), perm=perm, lo=lo, n_local=graph.n_local, nnz=graph.graph.nnz, **results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
