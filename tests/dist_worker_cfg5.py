"""Worker of tests/test_gpu_fullsize.py::test_cfg5_*: ONE rank over RCCL on the real engine.

mode "big":   the graph of BASELINE.json configs[4] (RMAT scale 27, edge factor 8: 134 M nodes, ~1.07 G edges) through the
              row-partitioned path (DistributedPageRank over pgh_dist_*), then -- after the slice has been given up -- the same
              graph and personalizations through the single-GPU engine; properties of both and their difference go to a .npz.
mode "slice": the partitioned path with the 8-way layout of the 8-GPU run (PGH_BLOCKS=8) at a scale the oracle finishes in
              seconds; the un-permuted result goes to a .npz for the comparison with the oracle in the test process."""
import gc
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, mode, scale, ef = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist
    import pygrank_amd as pg
    from pygrank_amd import _lib
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import DistributedPageRank, rmat_partitioned
    torch.cuda.set_device(0)
    _lib.ensure_init(0)
    pg.load_backend("hip")
    dist.init_process_group(backend="nccl")
    assert dist.get_world_size() == 1
    graph = rmat_partitioned(scale, ef, 0, 1, seed=0)
    n, nnz = graph.n, graph.graph.nnz
    perm = np.array(graph.perm)
    live = perm >= 0
    deg = np.asarray(graph.graph.degrees())               # row sums of M per (relabelled) source
    cand = np.flatnonzero(deg > 0)
    rng = np.random.default_rng(5)
    seeds = [np.sort(rng.choice(cand, 100, replace=False)) for _ in range(2)]
    weights = [rng.random(100) + 0.5 for _ in range(2)]

    def personalization(parts):
        p = np.zeros(n)
        for k, c in parts:
            p[seeds[k]] += c * weights[k]
        return p

    kw = dict(alpha=0.85, tol=1e-6, error_type="l1", max_iters=1000)
    fmt = graph.graph.format()
    out = {}
    runs = {"a": [(0, 1.0)], "b": [(1, 1.0)], "ab": [(0, 2.0), (1, 3.0)]}
    part = {}
    for name, parts in runs.items():
        ranker = DistributedPageRank(**kw)
        p_new = personalization(parts)
        got = np.asarray(ranker.rank(graph, DeviceVector.from_host(p_new)), dtype=np.float64)
        part[name] = (got, int(ranker.iteration), p_new)
        out[name + "_iters_part"] = int(ranker.iteration)
        out[name + "_sum_part"] = float(got.sum())
        out[name + "_psum"] = float(p_new.sum())
    # the grouped ncclSend / ncclRecv exchange of the N-rank run, executed by this lone rank TO ITSELF (PGH_DIST_P2P_ALONE=1: compact
    # numbering of the slice's cold sources -> pack launch -> point-to-point transfers -> finish; without the switch a rank alone writes
    # its slice in place and exchanges nothing).  The compact numbering of a rank that references every live slot is the identity, so the
    # run must reproduce the in-place run bit for bit.
    os.environ["PGH_DIST_P2P_ALONE"] = "1"
    ranker = DistributedPageRank(**kw)
    again = np.asarray(ranker.rank(graph, DeviceVector.from_host(part["a"][2])), dtype=np.float64)
    del os.environ["PGH_DIST_P2P_ALONE"]
    out["p2p_exchange"] = str(ranker.exchange.get("exchange"))
    out["p2p_driver"] = str(ranker.exchange.get("driver"))
    out["p2p_iters"] = int(ranker.iteration)
    out["p2p_bits_equal"] = int(np.array_equal(again, part["a"][0]))
    out["p2p_max_abs_diff"] = float(np.max(np.abs(again - part["a"][0])))
    del again
    DistributedPageRank(**kw).rank(graph, DeviceVector.from_host(part["a"][2]))       # back to the in-place layout for what follows
    if mode == "big":
        # linearity in the personalization: runs of a FIXED number of iterations (a stopping rule cuts the three runs at
        # different points of their tails) WITHOUT the L1 quotient (renormalising an iterate that lost mass to dangling nodes is
        # not linear): rank(2 a + 3 b) = 2 rank(a) + 3 rank(b) up to f32 rounding
        fixed = {}
        for name, parts in runs.items():
            ranker = DistributedPageRank(alpha=0.85, error_type="iters", max_iters=13, use_quotient=False)
            fixed[name] = np.asarray(ranker.rank(graph, DeviceVector.from_host(personalization(parts))), dtype=np.float64)
            assert ranker.iteration == 13
        mix = 2.0 * fixed["a"] + 3.0 * fixed["b"]
        out["linearity_rel_linf"] = float(np.max(np.abs(fixed["ab"] - mix)) / np.max(np.abs(mix)))
        del fixed, mix
    if mode == "slice":
        got_old = np.zeros(int(live.sum()))
        p_old = np.zeros(int(live.sum()))
        got_old[perm[live]] = part["a"][0][live]
        p_old[perm[live]] = part["a"][2][live]
        np.savez(os.path.join(out_dir, "slice.npz"), ranks=got_old, p=p_old, iters=part["a"][1], nnz=nnz, format=fmt, **out)
    else:
        # the partitioned path gives its memory back before the single-GPU engine builds the same graph
        del ranker
        graph.graph.destroy()
        gc.collect()
        torch.cuda.empty_cache()
        from pygrank_amd.synthetic import rmat_graph
        adj = rmat_graph(scale, ef, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
        assert adj.array.nnz == nnz and adj.array.shape[0] == int(live.sum())
        n_old = adj.array.shape[0]
        for name in runs:
            got_new, iters_part, p_new = part[name]
            p_old = np.zeros(n_old)
            p_old[perm[live]] = p_new[live]
            single = pg.PageRank(alpha=0.85, error_type=pg.L1, tol=1e-6, max_iters=1000)
            ref = np.asarray(single.rank(adj, pg.to_signal(adj, p_old)).np, dtype=np.float64)
            got_old = np.zeros(n_old)
            got_old[perm[live]] = got_new[live]
            out[name + "_iters_single"] = int(single.convergence.iteration)
            out[name + "_rel_linf"] = float(np.max(np.abs(got_old - ref)) / np.max(np.abs(ref)))
            out[name + "_sum_single"] = float(ref.sum())
            out[name + "_pad_mass"] = float(np.abs(got_new[~live]).sum())
        out["format_single"] = adj.array.format()
        np.savez(os.path.join(out_dir, "big.npz"), n=n_old, nnz=nnz, format=fmt, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
