"""GPU parity tests proper: every check calls libpgh_hip.so through the C-ABI on a real MI355X and is compared
with the oracle (oracle/ref_loops.py, pinned to the reference by tests/test_oracle_golden.py) and with the
committed golden vectors of the reference itself (tests/golden/golden.npz).

Bars: bit-exact for format conversion / integer work (upload + transposition, filter_out, iteration
counts); <= 1e-6 relative L-inf for floating point (BASELINE.json north_star), tolerance stated per test.
"""
import numpy as np
import pytest

import cases
import core_checks
import kernel_checks
from parity_common import EPS32, rel_linf, run_engine, run_oracle, runs_in_f64, tol_is_fp32_safe, tolerance_for

pytestmark = pytest.mark.gpu


def test_engine_is_the_hip_library(gpu_engine):
    from pygrank_amd import _lib
    assert _lib.runtime_name() == "hip:gfx950"
    import ctypes
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().pgh_device_name(buf, 256))
    assert b"gfx950" in buf.value


@pytest.mark.parametrize("check", kernel_checks.ALL, ids=[c.__name__ for c in kernel_checks.ALL])
def test_kernels(gpu_engine, check):
    check(gpu_engine)


@pytest.mark.parametrize("check", core_checks.ALL, ids=[c.__name__ for c in core_checks.ALL])
def test_core(gpu_engine, check):
    check(gpu_engine)


@pytest.mark.parametrize("name,gkey,algo,kwargs", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_filters_match_reference(gpu_engine, golden, graphs, name, gkey, algo, kwargs):
    A, directed, p = graphs(gkey)
    got, iters, ranker = run_engine(gpu_engine, A, directed, p, algo, kwargs)
    # engine epsilon() is fp32 (convergence.py:101) -- except that a tolerance below it sends the whole-loop routes to the f64 image, where the
    # REFERENCE's count and result are reproduced (round 6)
    f64 = runs_in_f64(algo, kwargs)
    want, want_iters = run_oracle(A, directed, p, algo, kwargs, **({} if f64 else dict(eps=EPS32)))
    assert iters == want_iters
    assert rel_linf(got, want) <= tolerance_for(kwargs)
    if tol_is_fp32_safe(kwargs) or f64:
        assert iters == int(golden[name + "|iters"])
        assert rel_linf(got, golden[name + "|ranks"]) <= tolerance_for(kwargs)
    if algo != "lowpass" and not kwargs.get("converge_to_eigenvectors"):
        assert hasattr(ranker, "last_loop")


@pytest.mark.parametrize("name,gkey,algo,kwargs", [c for c in cases.CASES if c[1] in ("rmat10_dir", "weighted300")],
                         ids=[c[0] for c in cases.CASES if c[1] in ("rmat10_dir", "weighted300")])
def test_generic_route_matches_reference(gpu_engine, graphs, name, gkey, algo, kwargs):
    """Same cases with the fused device loop disabled: per-step backend primitives (the route the unmodified
    reference filters would take through the backend module)."""
    A, directed, p = graphs(gkey)
    got, iters, ranker = run_engine(gpu_engine, A, directed, p, algo, kwargs, _fused_loop=lambda *a, **k: False,
                                    _fused_rank=lambda *a, **k: None)
    assert not hasattr(ranker, "last_loop")            # no device loop ran: one engine call per backend primitive
    want, want_iters = run_oracle(A, directed, p, algo, kwargs, eps=EPS32)
    assert iters == want_iters                         # the same bar as the fused route
    assert rel_linf(got, want) <= tolerance_for(kwargs)


def test_hip_matches_host_double_entry_points(gpu_engine, oracle_build_dir):
    """C-ABI cross-check: the HIP library and the independent host restatement of the same ABI agree on the
    device-loop results for a seeded RMAT graph (iterations equal, ranks within 1e-6 rel L-inf)."""
    import ctypes as C
    import os
    import scipy.sparse as sp
    from oracle import ref_loops as orc, rmat_np
    from pygrank_amd import _lib as L
    A = rmat_np.rmat_csr(15, 16, seed=4)
    M = sp.csr_array(orc.normalize(A, "col", True))
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 100, seed=1)] = 1.0 / 100
    dbl = L._bind(C.CDLL(os.path.join(oracle_build_dir, "libpgh_host_oracle.so")))
    results = []
    for lib in (L.lib(), dbl):
        g, vp, vr = L.c_graph(), L.c_vec(), L.c_vec()
        ip, idx, dat = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64)
        assert lib.pgh_graph_from_csr(M.shape[0], M.shape[1], M.nnz, ip.ctypes.data, idx.ctypes.data, dat.ctypes.data, 0, C.byref(g)) == 0
        assert lib.pgh_vec_alloc(len(p), C.byref(vp)) == 0 and lib.pgh_vec_alloc(len(p), C.byref(vr)) == 0
        assert lib.pgh_vec_h2d_f64(vp, p.ctypes.data, len(p)) == 0 and lib.pgh_vec_h2d_f64(vr, p.ctypes.data, len(p)) == 0
        cfg = L.LoopCfg(alpha=0.85, use_quotient=1, err_kind=L.ERR_L1, tol=1e-6, max_iters=1000, end_modulo=1, out_scale=1.0)
        res = L.LoopResult()
        assert lib.pgh_ppr_run(g, vp, vr, C.byref(cfg), C.byref(res)) == 0, lib.pgh_last_error()
        out = np.empty(len(p))
        assert lib.pgh_vec_d2h_f64(vr, out.ctypes.data, len(p)) == 0
        results.append((out, res.iterations, res.converged, res.spmv_count))
        lib.pgh_vec_free(vp), lib.pgh_vec_free(vr), lib.pgh_graph_destroy(g)
    (a, ia, ca, sa), (b, ib, cb, sb) = results
    assert (ia, ca, sa) == (ib, cb, sb) and ca == 1
    assert rel_linf(a, b) <= 1e-6
    want, want_iters = orc.pagerank(M, p, alpha=0.85, error_type="l1", tol=1e-6, max_iters=1000)
    assert ia == want_iters and rel_linf(a, want) <= 1e-6


def test_full_size_properties_rmat20(gpu_engine):
    """Size-independent properties on a larger graph than the oracle is run on in the other tests
    (RMAT scale 20, ~16M edges): linearity of conv, conservation of mass for column-normalised graphs without
    dangling rows in the seed set's reach, equality of single-shot steps and the device loop, idempotence of
    a converged run under warm start."""
    import scipy.sparse as sp
    from oracle import ref_loops as orc, rmat_np
    pg = gpu_engine
    A = rmat_np.rmat_csr(20, 16, seed=0)
    n = A.shape[0]
    M = sp.csr_array(orc.normalize(A, "col", True))
    g = pg.scipy_sparse_to_backend(M)
    rng = np.random.default_rng(5)
    x = rng.random(n)
    y = rng.random(n)
    dx, dy = pg.to_array(x), pg.to_array(y)
    # linearity: conv(2x + 3y) == 2 conv(x) + 3 conv(y) (fp32 rounding only)
    lhs = np.asarray(pg.conv(dx * 2.0 + dy * 3.0, g))
    rhs = np.asarray(pg.conv(dx, g) * 2.0 + pg.conv(dy, g) * 3.0)
    assert np.max(np.abs(lhs - rhs)) <= 8 * EPS32 * np.max(np.abs(rhs))
    # checksum: sum(conv(x)) == x . degrees(M)  (column sums of M^T are the row sums of M)
    s1 = pg.sum(pg.conv(dx, g))
    s2 = pg.dot(dx, pg.degrees(g))
    assert abs(s1 - s2) <= 1e-6 * abs(s2)
    # against scipy at full size (this is the cfg2-shaped workload at 1/8 scale; oracle finishes in ~1 s)
    ref = x.astype(np.float32).astype(np.float64) @ M
    got = np.asarray(pg.conv(dx, g))
    assert np.max(np.abs(got - ref)) <= 1e-6 * np.max(np.abs(ref))
    # headline stopping rule on the device loop vs the oracle loop
    p = np.zeros(n)
    p[rmat_np.seed_nodes(A, 100, seed=1)] = 1.0
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(assume_immutability=True)
    ranker = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=1000)
    ranks = ranker.rank(graph, p.copy())
    want, want_iters = orc.pagerank(M, p, alpha=0.85, error_type="l1", tol=1e-6, max_iters=1000)
    assert ranker.convergence.iteration == want_iters
    assert rel_linf(np.asarray(ranks.np), want) <= 1e-6
    # idempotence: restarting from the converged ranks stops at the first comparison
    again = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-5, max_iters=1000)
    again.rank(graph, p.copy(), warm_start=np.asarray(ranks.np) / np.asarray(ranks.np).sum())
    assert again.convergence.iteration == 2


@pytest.mark.parametrize("world,backend,scale,mode", [(1, "nccl", 14, ""), (2, "gloo", 14, ""), (4, "gloo", 14, ""), (2, "gloo", 18, ""),
                                                      (2, "gloo", 18, "single_queue"), (1, "nccl", 14, "single_queue"),
                                                      (1, "nccl", 18, ""), (1, "nccl", 18, "python_driver"),
                                                      (2, "gloo", 18, "engine_loop"), (4, "gloo", 14, "engine_loop"),
                                                      (2, "gloo", 18, "engine_loop_single_queue"), (1, "nccl", 18, "three_queues"),
                                                      (2, "gloo", 18, "engine_loop_no_a2a"), (2, "gloo", 18, "python_allgather"),
                                                      (2, "gloo", 18, "dense_images"), (4, "gloo", 18, "engine_loop"),
                                                      (1, "nccl", 18, "p2p_alone"), (1, "nccl", 18, "p2p_alone_three_queues"),
                                                      (2, "gloo", 18, "mixed_python"), (2, "gloo", 18, "engine_loop_mixed")],
                         ids=["rccl_x1", "gloo_x2", "gloo_x4", "gloo_x2_cold_image", "gloo_x2_cold_image_single_queue", "rccl_x1_single_queue",
                              "rccl_x1_cold_image_split_regions", "rccl_x1_cold_image_python_driver",
                              "gloo_x2_cold_image_engine_loop", "gloo_x4_engine_loop", "gloo_x2_cold_image_engine_loop_single_queue",
                              "rccl_x1_cold_image_three_queues", "gloo_x2_cold_image_engine_loop_without_all_to_all",
                              "gloo_x2_cold_image_python_driver_all_gather", "gloo_x2_cold_image_dense_layout", "gloo_x4_cold_image_engine_loop",
                              "rccl_x1_cold_image_send_recv_to_itself", "rccl_x1_cold_image_send_recv_to_itself_three_queues",
                              "gloo_x2_one_dense_one_compact_slice_python_driver", "gloo_x2_one_dense_one_compact_slice_engine_loop"])
def test_row_partitioned_path_on_one_gpu(gpu_engine, tmp_path, world, backend, scale, mode):
    """The N > 1 code path on the real engine: relabelled slice generation, the device-driven pgh_dist_* loop, in-place
    collectives on device scalars, the trimmed all-gather -- against the oracle.  World size 1 runs over RCCL; world sizes
    2 and 4 share the single GPU of this box and exchange through gloo (functional coverage of the multi-rank device
    path: RCCL refuses two ranks on one GPU).  The same logic runs on the CPU in tests/test_distributed_cpu.py; 8 GPUs
    over RCCL/xGMI are the driver's scaling run.  The scale-18 case has sources outside the LDS hot cache and forces the
    propagation-blocking image of the cold entries onto every rank's slice."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ef = 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), os.path.join(root, "tests", "dist_worker.py"), str(tmp_path), str(scale), str(ef)]
    env = dict(os.environ, PYTHONPATH=root, PGH_TEST_ENGINE="hip", PGH_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # RCCL runs go through the engine's own loop (pgh_dist_ppr_run: RCCL, streams and events driven from C++, hot prefixes and
    # cold parts of the gather vector exchanged as two contiguous regions); gloo runs and "python_driver" through the staged
    # pgh_dist_* calls from pygrank_amd/distributed.py
    if backend == "nccl" and (mode or scale > 14):
        # a rank alone has nothing to exchange and nothing to add: the engine skips its collectives (the slice is written in place)
        # -- all but the plain scale-14 case make the RCCL calls all the same, so that ncclAllGather (in place) and ncclAllReduce run
        env.update(PGH_DIST_GATHER_ALONE="1", PGH_DIST_REDUCE_ALONE="1")
    if mode == "single_queue":           # the conservative switches of a first multi-GPU run: one communicator, one stream
        env.update(PGH_DIST_SINGLE_COMM="1", PGH_DIST_SINGLE_STREAM="1")
    if mode == "python_driver":
        env.update(PGH_DIST_NATIVE="0")
    if mode == "engine_loop_no_a2a":     # a host with all-gather / all-reduce only: compact slices copy their slots out of the gathered vector
        env.update(PGH_DIST_EXTERNAL_A2A="0")
    if mode == "python_allgather":       # the same fallback in the Python-driven loop
        env.update(PGH_DIST_EXCHANGE="allgather")
    if mode == "dense_images":           # rounds 1-4: the cold image numbers every live cold slot, the exchange is the all-gather alone
        env.update(PGH_DIST_NEED_LISTS="0")
    if mode.startswith("engine_loop"):   # the ENGINE's loop with several ranks (region offsets of every rank, split exchange, ...):
        env.update(PGH_DIST_NATIVE="external")      # the collectives come back to the host (pgh_comm_create_external over gloo)
        env.update(PGH_DIST_FINISH_SPLIT="2")       # ... and the finish kernel in two launches, as a large exchange has it
    if mode == "engine_loop_single_queue":          # ... on ONE queue: the whole packed slice of a rank as a single all-gather
        env.update(PGH_DIST_SINGLE_STREAM="1")
    if mode == "three_queues":           # one rank runs on one queue by default (nobody to overlap an exchange with): force the three,
        env.update(PGH_DIST_SINGLE_STREAM="0", PGH_DIST_FINISH_SPLIT="2")      # and the finish kernel in two launches (N > 1 does by itself)
    if mode.endswith("mixed") or mode.startswith("mixed"):      # rank 1's slice dense, rank 0's compact -- the all-gather stays,
        env.update(PGH_TEST_DENSE_RANK="1")                     # the compact slice copies its slots out of the gathered vector
    if mode.startswith("p2p_alone"):     # the N-rank exchange on one GPU: compact numbering -> pack launch -> grouped ncclSend / ncclRecv of the
        env.update(PGH_DIST_P2P_ALONE="1")           # rank's stretches TO ITSELF (without the switch a lone rank writes its slice in place)
    if mode == "p2p_alone_three_queues":
        env.update(PGH_DIST_SINGLE_STREAM="0", PGH_DIST_FINISH_SPLIT="2")
    if scale > 14:                       # four column blocks over two ranks: two all-gathers per exchange, like bench.py --gpus 2
        env.update(PGH_PB="1", PGH_PB_FORCE="1", PGH_DEBUG="1", PGH_BLOCKS="4", PGH_PB_HEAVY="64", PGH_PB_HUBMAX="500")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    if scale > 14:
        assert "[pgh] pb:" in res.stderr, res.stderr[-3000:]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    driver = {"": "engine (RCCL)" if backend == "nccl" else "python (torch.distributed)", "single_queue": "engine (RCCL)" if backend == "nccl"
              else "python (torch.distributed)", "python_driver": "python (torch.distributed)", "engine_loop": "engine (host collectives)",
              "engine_loop_single_queue": "engine (host collectives)", "three_queues": "engine (RCCL)",
              "engine_loop_no_a2a": "engine (host collectives)", "python_allgather": "python (torch.distributed)",
              "dense_images": "python (torch.distributed)", "p2p_alone": "engine (RCCL)", "p2p_alone_three_queues": "engine (RCCL)",
              "mixed_python": "python (torch.distributed)", "engine_loop_mixed": "engine (host collectives)"}[mode]
    assert all(str(part["driver"]) == driver for part in parts), [str(part["driver"]) for part in parts]
    # how the cold parts travelled (need lists wherever every slice has a cold image), and the received bytes a run reports
    # against the host-side count of the slots this slice references in its peers' blocks
    if scale > 14:
        kind = {"engine_loop_no_a2a": "all-gather + local compaction", "python_allgather": "all-gather", "dense_images": "all-gather"}.get(mode, "need lists")
        if driver == "engine (RCCL)" and not mode.startswith("p2p_alone"):        # one rank alone references every live slot of its own blocks: its compact numbering is the dense one,
            kind = "all-gather"              # the engine writes the slice in place and packs nothing
        if "mixed" in mode:                  # rank 0 compact, rank 1 dense: the dense all-gather on both, rank 0 compacts its own copy
            assert [int(part["need_total"]) > 0 for part in parts] == [True, False]
            kinds = [str(part["exchange_kind"]) for part in parts]
            assert kinds == (["all-gather + local compaction", "all-gather"] if mode == "engine_loop_mixed" else ["all-gather", "all-gather"]), kinds
            kind = "mixed"
        else:
            assert all(str(part["exchange_kind"]).startswith(kind) for part in parts), [str(part["exchange_kind"]) for part in parts]
            assert all((int(part["need_total"]) > 0) == (mode != "dense_images") for part in parts)
        if kind == "need lists":
            for part in parts:
                assert int(part["exchange_bytes"]) == int(part["expected_list_bytes"]), (int(part["exchange_bytes"]), int(part["expected_list_bytes"]))
    else:
        assert all(int(part["need_total"]) == 0 and str(part["exchange_kind"]) == "all-gather" for part in parts)
    if scale > 14 and driver.startswith("engine"):
        assert all(int(part["split_regions"]) == 1 for part in parts)      # hot prefixes and cold parts exchanged as two regions
    from parity_common import check_partition_against_oracle
    check_partition_against_oracle(parts, scale, ef)
    assert all(str(part["closed_form_driver"]) == driver for part in parts), [str(part["closed_form_driver"]) for part in parts]
    if driver == "engine (RCCL)":        # the probe that guards N > 1 (engine loop vs Python-driven loop on slices with a cold image)
        assert all(str(part["preflight_selftest"]) == "ok" for part in parts), [str(part["preflight_selftest"]) for part in parts]
    if scale > 14 and driver.startswith("engine"):
        # slices with a cold image: the residual of the L1 / Mabs rules is evaluated inside the finish kernel (ONE 4-scalar all-reduce
        # per iteration), and a personalization with negative entries makes it hand one step to the separate kernel
        for part in parts:
            assert int(part["l1_fused"]) == 1 and int(part["mabs_fused"]) == 1 and int(part["noquot_fused"]) == 0
            assert int(part["signed_paused"]) == 1
            # more than one rank on three queues (and the forced one-rank case): the finish kernel in two launches, exchanged rows first
            two = mode.startswith("engine_loop") or mode.endswith("three_queues")          # forced by the test (PGH_DIST_FINISH_SPLIT=2)
            assert int(part["l1_two_launches"]) == int(two) and int(part["noquot_two_launches"]) == int(two), (mode, world)
            assert int(part["closed_form_two_launches"]) == int(two)


@pytest.mark.parametrize("world,backend,width", [(2, "gloo", 5), (1, "nccl", 6)], ids=["gloo_x2", "rccl_x1_device_gather"])
def test_replica_split_of_a_seed_batch_on_one_gpu(gpu_engine, tmp_path, world, backend, width):
    """SURVEY.md 8e, last sentence: a batch of seed sets split across ranks that each hold the whole graph (NodeRanking.propagate,
    pygrank/core/signals.py:225-226 -> pgh_ppr_run_batch on every rank's share, no communication before the final all-gather of the
    result slabs).  Two ranks share this box's GPU over gloo (slabs staged through the host); one rank over RCCL runs the
    device-to-device all-gather.  Against the oracle: <= 1e-6 per column, equal iteration counts."""
    import os
    import subprocess
    import sys
    from test_distributed_cpu import check_replica_split
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scale, ef = 13, 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29671 + world), os.path.join(root, "tests", "dist_worker_replicas.py"), str(tmp_path), str(scale), str(ef), str(width)]
    env = dict(os.environ, PYTHONPATH=root, PGH_TEST_ENGINE="hip", PGH_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0",
               PGH_REPLICA_GATHER_ALONE="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "staging through the host" not in res.stderr, res.stderr[-2000:]       # the RCCL leg gathers device to device
    check_replica_split([np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)], scale, ef, width)


def test_two_gpus_over_rccl(gpu_engine, tmp_path):
    """More than one rank over RCCL (boxes with >= 2 GPUs only; this pool's boxes have one, the driver's scaling node has eight):
    the engine-driven loop must pass its own probe against the Python-driven loop (pygrank_amd.distributed._preflight), run every
    partitioned filter against the oracle and say that it was the engine that drove RCCL."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: RCCL refuses two ranks on one device (the 2- and 4-rank engine loop runs through host collectives above)")
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    scale, ef = 18, 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dist_worker.py"), str(tmp_path), str(scale), str(ef)]
    env = dict(os.environ, PYTHONPATH=root, PGH_TEST_ENGINE="hip", PGH_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PGH_PB="1", PGH_PB_FORCE="1", PGH_BLOCKS="4", PGH_PB_HEAVY="64", PGH_PB_HUBMAX="500", PGH_DIST_TIMEOUT_S="120",
               PGH_DIST_NATIVE="auto")         # (with more than one rank the engine loop is opt-in: probe, then the engine drives RCCL)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2)]
    from parity_common import check_partition_against_oracle
    check_partition_against_oracle(parts, scale, ef)
    assert [str(part["preflight"]) for part in parts] == ["ok", "ok"]
    assert all(str(part["driver"]) == "engine (RCCL)" and str(part["closed_form_driver"]) == "engine (RCCL)" for part in parts)
    assert all(int(part["l1_fused"]) == 1 and int(part["signed_paused"]) == 1 for part in parts)


def test_graph_dropout_on_the_cold_image(gpu_engine):
    """graph_dropout of single vectors on the production layout -- hot-only 16-bit stream + propagation-blocking image of the cold
    entries -- forced onto a scale-18 graph (kernel_checks.dropout_on_cold_image_check); the small-graph layouts run in
    test_kernels[check_graph_dropout], the bench graph in tests/test_gpu_fullsize.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r, %r]\nimport pygrank_amd as pg\npg.load_backend('hip')\nimport kernel_checks\n"
            "kernel_checks.dropout_on_cold_image_check(pg)\nprint('dropout cold image ok')\n"
            % (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")))
    env = dict(os.environ, PYTHONPATH=root, PGH_PB="1", PGH_PB_FORCE="1", PGH_BLOCKS="4", PGH_PB_HEAVY="64", PGH_PB_HUBMAX="500")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0 and "dropout cold image ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


def test_randomised_stress(gpu_engine):
    """tools/stress_gpu.py for 20 s: random graph shapes (empty / tiny / hub rows / power-law / uniform / banded, integer and
    real weights) x layout switches (1-8 column blocks, relabelling, trimmed gather vector, propagation-blocking image)
    against scipy in fp64.  Kernels are serialised so that an out-of-allocation access faults where it happens (that is how
    the unconditional row prefetch of k_bsf_partial was caught)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (PGH_SLAB_MB=0: every buffer an allocation of its own, so that an access past its end leaves the allocation and faults)
    env = dict(os.environ, PYTHONPATH=root, AMD_SERIALIZE_KERNEL="3", HIP_LAUNCH_BLOCKING="1", PGH_SLAB_MB="0")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_gpu.py"), "--seconds", "20", "--seed", "7"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0 and "stress ok" in res.stdout, res.stdout[-1500:] + res.stderr[-2500:]


def test_randomised_filters(gpu_engine):
    """tests/stress_filters.py for 20 s: random graphs x layout switches x {PageRank (3 residuals, with / without quotient),
    AbsorbingWalks, HeatKernel taylor / chebyshev, propagate} against the oracle on the engine's stored matrix, including
    agreement on non-convergence; kernels serialised."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, AMD_SERIALIZE_KERNEL="3", HIP_LAUNCH_BLOCKING="1", PGH_SLAB_MB="0")
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_filters.py"), "--seconds", "20", "--seed", "11"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0 and "filters stress ok" in res.stdout, res.stdout[-1500:] + res.stderr[-2500:]


def test_bench_two_ranks_on_one_gpu(gpu_engine):
    """The driver's N > 1 launch of bench.py on the real engine: two ranks sharing this box's GPU (gloo transport, see
    test_row_partitioned_path_on_one_gpu), one JSON line on stdout from rank 0, whole-job value."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "16"]
    env = dict(os.environ, PYTHONPATH=root, PGH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["unit"] == "GTEPS"
    assert out["config"]["exchange_bytes_per_iteration_per_gpu"] > 0 and len(out["config"]["iterations_per_step"]) == 2


def test_bench_eight_ranks_on_one_gpu(gpu_engine):
    """the world == 8 branch of bench.py (BASELINE.json configs[4]'s rank count) on the real engine -- eight supervised
    ranks sharing this box's GPU over gloo, one column block per rank, need-list exchange between all pairs, parity of the 8-way
    partition against the oracle, the same graph on one GPU, the replica split of 8 x 64 seed sets -- through the driver's launch line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--scale", "17"]
    env = dict(os.environ, PYTHONPATH=root, PGH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["fallback_rung"]["rung"] == 0
    assert out["parity"]["rel_linf"] <= 1e-6 and out["parity"]["gpu_iterations"] == out["parity"]["cpu_iterations"]
    same = out["same_graph_1gpu"]
    assert same["rel_linf_partitioned_vs_1gpu"] <= 1e-6 and same["iterations_1gpu"] == same["iterations_partitioned"]
    replicas = out["secondary"]["batch_of_64_seeds_replicas"]
    assert "error" not in replicas and replicas["ranks"] == 8 and replicas["seed_sets"] == 512, replicas


def test_bench_ladder_on_the_gpu(gpu_engine):
    """VERDICT r5 item 1b on the real engine: rank 1's child of rung 0 stops answering AFTER it has initialised the GPU and joined the
    first collectives; at the rung's deadline the supervisors (which never touch the GPU) end both children with SIGKILL and start a
    fresh child tree one rung down on the same GPU; the line names the rung."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29643", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "16",
           "--no-secondary"]
    env = dict(os.environ, PYTHONPATH=root, PGH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PGH_BENCH_TEST_FAIL="0:1:hang:after_timing", PGH_BENCH_RUNG_S="90")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    rung = out["config"]["fallback_rung"]
    assert rung["rung"] == 1 and rung["env"] == dict(PGH_DIST_SINGLE_COMM="1", PGH_DIST_SINGLE_STREAM="1"), rung
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["parity"]["rel_linf"] <= 1e-6 and "incomplete" not in out["config"]
    # (the child's own watchdog -- 20 s inside the rung's deadline -- ends a rank that is still alive; the supervisor's SIGKILL ends one that is not)
    assert "did not finish within its watchdog" in res.stderr or "no result within 90 s" in res.stderr, res.stderr[-3000:]
    assert "rung 0 (default) failed on ranks" in res.stderr, res.stderr[-3000:]


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo")], ids=["rccl_x1", "gloo_x2"])
def test_partitioned_upload_of_a_scipy_graph_on_gpu(gpu_engine, tmp_path, world, backend):
    """pgh_graph_from_csr_part on the real engine (a caller's weighted scipy graph, relabelled and sliced per rank) driven
    through the device partitioned loop, against the oracle.  Same worker as tests/test_distributed_cpu.py."""
    import os
    import subprocess
    import sys
    import scipy.sparse as sp
    from oracle import ref_loops as orc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import dist_worker_csr
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29650 + world), os.path.join(root, "tests", "dist_worker_csr.py"), str(tmp_path)]
    env = dict(os.environ, PYTHONPATH=root, PGH_TEST_ENGINE="hip", PGH_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    A, p = dist_worker_csr.make_graph()
    M = sp.csr_array(orc.normalize(A, "col", True))
    assert sum(int(part["nnz"]) for part in parts) == M.nnz
    for name, kw in (("l1", dict(error_type="l1", tol=1e-6, max_iters=500)), ("mabs", dict(error_type="mabs", tol=1e-7, max_iters=500))):
        want, want_iters = orc.pagerank(M, p, alpha=0.85, eps=EPS32, **kw)
        got = sum(part[name + "_ranks"] for part in parts)
        assert all(int(part[name + "_iters"]) == want_iters for part in parts), name
        assert rel_linf(got, want) <= 1e-6, name
