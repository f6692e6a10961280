"""Worker of the replica-split tests (tests/test_distributed_cpu.py, tests/test_gpu_parity.py): one process per rank, every rank
holds the WHOLE graph and the same [n, B] feature matrix, ReplicatedPropagation hands each rank its share of the columns
(NodeRanking.propagate, pygrank/core/signals.py:225-226, as multi-seed batches with zero communication)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def features(n, width, seed=3):
    """[n, width] personalizations: 5 weighted seeds per column; column 2 (when present) is all zeros (abstract_filters.py:53-54)."""
    rng = np.random.default_rng(seed)
    F = np.zeros((n, width))
    for j in range(width):
        if j != 2:
            F[rng.choice(n, 5, replace=False), j] = rng.random(5) + 0.5
    return F


def main():
    out_dir, scale, ef, width = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch.distributed as dist
    import pygrank_amd as pg
    from pygrank_amd import _lib
    on_gpu = os.environ.get("PGH_TEST_ENGINE") == "hip"
    if on_gpu:
        import torch
        device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(device)
        _lib.ensure_init(device)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import host_double
        host_double.install()
    pg.load_backend("hip")
    dist.init_process_group(backend=os.environ.get("PGH_DIST_BACKEND", "nccl" if on_gpu else "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    from pygrank_amd.distributed import ReplicatedPropagation, replica_columns
    from pygrank_amd.synthetic import rmat_graph
    adj = rmat_graph(scale, ef, seed=0, normalization="col")
    n = adj.array.shape[0]
    F = features(n, width)
    results = {}
    for name, kw in (("l1", dict(error_type=pg.L1, tol=1e-6, max_iters=500)), ("mabs", dict(tol=1e-7, max_iters=500))):
        split = ReplicatedPropagation(pg.PageRank(alpha=0.85, **kw))
        whole = split.propagate(adj, F)                            # [n, width] on every rank
        lo, hi = split.columns
        assert (lo, hi) == replica_columns(width, rank, world)
        results[name + "_ranks"] = np.asarray(whole)
        iters = np.full(width, -1, dtype=np.int64)                  # this rank knows the counts of its own columns
        at = lo
        for batch in split.last_batches if hi > lo else []:
            for col in batch:
                iters[at] = col["iterations"]
                at += 1
        assert at == hi or hi == lo
        results[name + "_iters"] = iters
        mine = split.propagate(adj, F, gather=False)                # the share alone: no collective at all
        results[name + "_share_equal"] = int(mine is None if hi == lo else np.array_equal(np.asarray(mine), results[name + "_ranks"][:, lo:hi]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
