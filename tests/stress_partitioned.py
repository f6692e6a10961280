"""Randomised parity harness of the row-partitioned ENGINE loop (pgh_dist_ppr_run / pgh_dist_poly_run) with several ranks sharing one
GPU: collectives through the host (pgh_comm_create_external over gloo; RCCL refuses two ranks on one device).  Every case draws a
graph (generated slices or a caller's matrix with a ragged id count), a layout (column blocks, cold image forced or not, regions split
or not, the finish kernel in one or two launches), a filter and its stopping rule, and a personalization (seed set, dense, signed),
runs it on every rank and compares the assembled result with the oracle on the un-partitioned graph: <= 2e-6 and equal iteration
counts (a residual within f32 rounding of the tolerance may stop a step apart).  Not collected by pytest (a GPU-minutes sink by design):

    (PGH_TEST_ENGINE=host: the Python-driven loop on the host double, no GPU)
    PGH_DIST_NATIVE=external PGH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 \
        --master-addr 127.0.0.1 --master-port 29655 tests/stress_partitioned.py --seconds 120 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

EPS32 = float(np.finfo(np.float32).eps)
LAYOUT_KEYS = ("PGH_BLOCKS", "PGH_PB", "PGH_PB_FORCE", "PGH_PB_HEAVY", "PGH_PB_HUBMAX", "PGH_DIST_SPLIT", "PGH_DIST_FINISH_SPLIT")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-scale", type=int, default=15)
    ap.add_argument("--only", type=int, default=-1, help="run this case of the stream alone (the draws of the others are made and dropped)")
    args = ap.parse_args()
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    import pygrank_amd as pg
    from pygrank_amd import _lib
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import (DistributedAbsorbingWalks, DistributedHeatKernel, DistributedPageRank, DistributedPageRankClosed,
                                         partition_scipy, rmat_partitioned)
    from oracle import ref_loops as orc
    from oracle import rmat_np
    if os.environ.get("PGH_TEST_ENGINE", "hip") == "hip":
        device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(device)
        _lib.ensure_init(device)
    else:                                                  # PGH_TEST_ENGINE=host: the Python-driven loop on the host double (no GPU)
        import host_double
        host_double.install()
    pg.load_backend("hip")
    dist.init_process_group(backend=os.environ.get("PGH_DIST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    ctl = dist.new_group(backend="gloo")                   # the harness's own traffic (host tensors), whatever the filters exchange over
    rng = np.random.default_rng(args.seed)                 # the SAME stream on every rank: every draw below is collective
    deadline = time.time() + args.seconds
    done, borderline, by_kind, drivers, stats = 0, 0, {}, set(), {}
    while True:
        go = torch.tensor([1 if time.time() < deadline else 0])
        dist.broadcast(go, 0, group=ctl)                              # rank 0's clock decides for everybody
        if int(go.item()) == 0:
            break
        # ---- the graph and its layout
        scale, ef, gseed = int(rng.integers(8, args.max_scale + 1)), int(rng.choice([4, 8, 16])), int(rng.integers(0, 1000))
        layout = {}
        if rng.random() < 0.6:                             # a cold image whatever the size, with small hub / heavy-row marks
            layout.update(PGH_PB="1", PGH_PB_FORCE="1", PGH_PB_HEAVY=str(int(rng.choice([16, 64, 256]))), PGH_PB_HUBMAX=str(int(rng.choice([100, 500, 4000]))))
        if rng.random() < 0.5:
            layout["PGH_BLOCKS"] = str(int(rng.choice([b for b in (2, 4, 8) if b >= world])))
        if rng.random() < 0.25:
            layout["PGH_DIST_SPLIT"] = "0"
        layout["PGH_DIST_FINISH_SPLIT"] = str(int(rng.choice([0, 2])))
        for key in LAYOUT_KEYS:
            os.environ.pop(key, None)
        os.environ.update(layout)
        from_matrix = rng.random() < 0.4
        cut = int(rng.integers(0, 200))
        shape, kind, rule = int(rng.integers(0, 4)), int(rng.integers(0, 4)), str(rng.choice(["l1", "mabs", "linf", "iters"]))
        case_rng = np.random.default_rng([args.seed, done])          # the case's own stream: a case can be re-run alone (--only)
        if args.only >= 0 and done != args.only:
            done += 1
            if done > args.only:
                break
            continue
        rng_main, rng = rng, case_rng
        A = rmat_np.rmat_csr(scale, ef, seed=gseed)
        if from_matrix:                                    # a caller's matrix: ragged id count (rows cut off), padded by the partition
            n = int(A.shape[0] - cut)
            A = sp.csr_array(A[:n, :n])
            M = sp.csr_array(orc.normalize(A, "col", True))
            graph = partition_scipy(M, rank, world)
        else:
            n = A.shape[0]
            M = sp.csr_array(orc.normalize(A, "col", True))
            graph = rmat_partitioned(scale, ef, rank, world, seed=gseed)
        perm = np.asarray(graph.perm)                      # new id -> original id (-1: padding)
        lo, n_local = graph.row_begin, graph.n_local
        mine = perm[lo:lo + n_local]
        # ---- the personalization (original ids)
        p_old = np.zeros(n)
        if shape == 0:
            k = int(rng.integers(1, 40))
            p_old[rng.choice(n, min(k, n), replace=False)] = rng.random(min(k, n)) + 0.5
        elif shape == 1:
            p_old[:] = rng.random(n)
        elif shape == 2:
            p_old[:] = np.where(rng.random(n) < 0.05, rng.random(n), 0.0)
            p_old[int(rng.integers(0, n))] = 1.0
        else:                                              # signed: the in-kernel residual must pause, the result is the oracle's
            k = int(rng.integers(2, 60))
            # (a quarter of the entries negative and small: sum(p) stays well away from zero -- the L1 quotient divides by sum(y), and
            # a sum that cancels makes any f32 evaluation ill-conditioned, partitioned or not)
            p_old[rng.choice(n, min(k, n), replace=False)] = (rng.random(min(k, n)) + 0.5) * np.where(rng.random(min(k, n)) < 0.25, -0.25, 1.0)
        p_local = np.where(mine >= 0, p_old[np.maximum(mine, 0)], 0.0)
        # ---- the filter
        kw = dict(error_type=rule, max_iters=int(rng.integers(5, 40)) if rule == "iters" else 600)
        if rule != "iters":
            kw["tol"] = float(10.0 ** rng.uniform(-8.0 if rule == "mabs" else -6.5, -3.0))
        if kind == 0:
            alpha = float(rng.uniform(0.5, 0.95))
            use_quotient = bool(rng.random() < 0.8)
            algo = DistributedPageRank(alpha=alpha, use_quotient=use_quotient, **kw)
            ref = lambda **k: orc.pagerank(M, p_old, alpha=alpha, eps=EPS32, use_quotient=use_quotient, **k)   # noqa: E731
        elif kind == 1:
            alpha = float(rng.uniform(0.5, 0.95))
            algo = DistributedAbsorbingWalks(alpha=alpha, **kw)
            ref = lambda **k: orc.absorbing_walks(M, p_old, alpha=alpha, eps=EPS32, **k)                          # noqa: E731
        elif kind == 2:
            t = float(rng.uniform(1.0, 6.0))
            kw["max_iters"] = min(kw["max_iters"], 100)
            algo = DistributedHeatKernel(t=t, **kw)
            ref = lambda **k: orc.heat_kernel(M, p_old, t=t, eps=EPS32, **k)                                     # noqa: E731
        else:
            alpha = float(rng.uniform(0.5, 0.9))
            kw["max_iters"] = min(kw["max_iters"], 300)
            algo = DistributedPageRankClosed(alpha=alpha, **kw)
            ref = lambda **k: orc.pagerank_closed(M, p_old, alpha=alpha, eps=EPS32, **k)                          # noqa: E731
        what = f"case {done}: seed {args.seed} scale {scale} ef {ef} gseed {gseed} matrix {from_matrix} n {n} layout {layout} p {shape} " \
               f"filter {type(algo).__name__} {kw} world {world}"
        try:
            failure = None
            try:
                out = np.asarray(algo.rank(graph, DeviceVector.from_host(p_local)), dtype=np.float64)
                iters = int(algo.iteration)
            except Exception as exc:                       # the oracle may refuse the same case (out of iterations): compared below
                out, iters, failure = None, -1, f"{type(exc).__name__}: {str(exc)[:200]}"
            drivers.add(str(getattr(algo, "exchange", {}).get("driver")))
            # every rank's slice -> rank 0 (gloo), un-permuted into original ids
            sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([n_local if out is not None else 0], dtype=torch.int64), group=ctl)
            ok_everywhere = all(int(s.item()) > 0 for s in sizes)
            want, want_iters, want_failure = None, -1, None
            try:
                want, want_iters = ref(**kw)
            except Exception as exc:
                want_failure = f"{type(exc).__name__}: {str(exc)[:200]}"
            if want_failure is not None or not ok_everywhere:
                assert (want_failure is not None) == (failure is not None), (what, failure, want_failure)
            else:
                slices = [torch.zeros(int(s.item()), dtype=torch.float64) for s in sizes]
                dist.all_gather(slices, torch.from_numpy(np.ascontiguousarray(out)), group=ctl)
                got = np.zeros(n)
                for r, piece in enumerate(slices):
                    ids = perm[r * n_local:(r + 1) * n_local]
                    got[ids[ids >= 0]] = piece.numpy()[ids >= 0]
                delta = iters - want_iters
                if iters != want_iters:
                    # a residual within f32 rounding of the tolerance may stop a step apart (tests/stress_filters.py has the same
                    # rule): the result is then held against the oracle stopped after the engine's number of steps
                    assert rule != "iters" and abs(iters - want_iters) <= max(1, want_iters // 25), (what, iters, want_iters)
                    want, _ = ref(error_type="iters", max_iters=iters)
                    borderline += 1
                top = float(np.max(np.abs(want)))
                if top > 0:
                    err = float(np.max(np.abs(got - want))) / top
                    # what the harness proves: the distribution per leg, and north_star's 1e-6 for every case that is
                    # not a signed personalization (it loses digits to cancellation in ANY f32 evaluation -- the host double, an
                    # independent f32 implementation, misses the oracle by 2.7e-6 where the engine misses it by 2.6e-6, seed 24 case
                    # 2395 -- so those are held to 6e-6)
                    leg = type(algo).__name__ + (", signed p" if shape == 3 else "")
                    row = stats.setdefault(leg, dict(runs=0, le_1e6=0, le_2e6=0, above=0, worst=0.0, deltas={}))
                    row["runs"] += 1
                    row["le_1e6" if err <= 1e-6 else ("le_2e6" if err <= 2e-6 else "above")] += 1
                    row["worst"] = max(row["worst"], err)
                    row["deltas"][delta] = row["deltas"].get(delta, 0) + 1
                    assert err <= (6e-6 if shape == 3 else 1e-6), (what, err)
                else:
                    assert not np.any(got), what
        except AssertionError:
            sys.stderr.write(f"[rank {rank}] FAILED {what}\n")
            raise
        by_kind[type(algo).__name__] = by_kind.get(type(algo).__name__, 0) + 1
        graph.graph.destroy()
        rng = rng_main
        done += 1
    for key in LAYOUT_KEYS:
        os.environ.pop(key, None)
    if rank == 0:
        for leg in sorted(stats):
            row = stats[leg]
            deltas = " ".join(f"{k:+d}:{v}" for k, v in sorted(row["deltas"].items()))
            print(f"  {leg:40s} runs {row['runs']:6d}  rel-Linf <= 1e-6: {row['le_1e6']:6d}  (1e-6, 2e-6]: {row['le_2e6']:4d}  above: {row['above']:4d}  "
                  f"worst {row['worst']:.2e}  iterations engine - oracle: {deltas}")
        print(f"partitioned stress ok: {done} cases in {args.seconds:.0f} s (seed {args.seed}, world {world}, drivers {sorted(drivers)}, {borderline} stopped a step apart): {by_kind}")
    dist.barrier(group=ctl)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
