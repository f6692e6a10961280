"""Worker of tests/test_distributed_cpu.py::test_partitioned_upload_of_a_scipy_graph: one process per rank (gloo), engine = host
test double (or the real engine with PGH_TEST_ENGINE=hip).  Every rank builds the SAME scipy graph, partitions it with
pygrank_amd.distributed.partition_scipy and runs DistributedPageRank on its slice."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_graph(n=3001, m=40000, seed=5):
    """directed, weighted (non-integer weights: the valued layout), a few dangling rows, n not a multiple of anything"""
    rng = np.random.default_rng(seed)
    rows = (rng.pareto(1.2, m) * 40).astype(np.int64) % n          # skewed sources
    cols = rng.integers(0, n, m)
    keep = rows % 17 != 3
    A = sp.coo_array((rng.random(m)[keep] + 0.25, (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    p = np.zeros(n)
    p[rng.choice(n, 25, replace=False)] = rng.random(25) + 0.5
    return A, p


def main():
    out_dir = sys.argv[1]
    import torch.distributed as dist
    import pygrank_amd as pg
    from pygrank_amd import _lib
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import DistributedPageRank, partition_scipy
    from oracle import ref_loops as orc
    on_gpu = os.environ.get("PGH_TEST_ENGINE") == "hip"
    if on_gpu:
        import torch
        device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(device)
        _lib.ensure_init(device)
    else:
        import host_double
        host_double.install()
    pg.load_backend("hip")
    dist.init_process_group(backend=os.environ.get("PGH_DIST_BACKEND", "nccl" if on_gpu else "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    A, p = make_graph()
    M = orc.normalize(A, "col", True)
    part = partition_scipy(M, rank, world)
    results = {}
    for name, kw in (("l1", dict(error_type="l1", tol=1e-6, max_iters=500)), ("mabs", dict(error_type="mabs", tol=1e-7, max_iters=500))):
        ranker = DistributedPageRank(alpha=0.85, **kw)
        out = ranker.rank(part, DeviceVector.from_host(part.local_slice(p)))
        full = np.zeros(A.shape[0])
        part.scatter_slice(np.asarray(out), full)
        results[name + "_ranks"] = full                    # zeros outside this rank's slice
        results[name + "_iters"] = ranker.iteration
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), n_pad=part.n, n_local=part.n_local, nnz=part.graph.nnz, **results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
