"""N > 1 path on the CPU: world_size 2 and 4 gloo runs of the row-partitioned PageRank (pygrank_amd/distributed.py)
against the host test double, compared with the single-process oracle on the same graph (ranks un-permuted).
Covers the global relabelling, equal-sized slices, the all-gather of the gather vector and the scalar all-reduces."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ref_loops as orc, rmat_np
from parity_common import check_partition_against_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS32 = float(np.finfo(np.float32).eps)


@pytest.mark.parametrize("world", [2, 4])
def test_row_partitioned_pagerank_gloo(tmp_path, oracle_build_dir, world):
    scale, ef = 10, 8
    port = 29500 + world + (os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(scale), str(ef)]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    n = 1 << scale
    assert [int(part["n_local"]) for part in parts] == [n // world] * world     # equal slices: unpadded all-gather
    nnz_per_rank = [int(part["nnz"]) for part in parts]
    assert max(nnz_per_rank) <= 1.05 * np.mean(nnz_per_rank), nnz_per_rank      # hot-first round-robin deal: balanced work
    check_partition_against_oracle(parts, scale, ef)
    assert all(str(part["driver"]) == "python (torch.distributed)" for part in parts)


def test_bench_two_ranks_prints_one_json_line(oracle_build_dir):
    """The driver's N > 1 launch of bench.py (torch.distributed.run, one rank per GPU): rank 0 prints exactly one JSON
    line on stdout (RCCL/gloo chatter goes to stderr), the value is the whole-job rate."""
    import json
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "bench_dist_worker.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--scale", "11", "--ef", "8"]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    # (the host double at scale 11 can round to 0.00 GTEPS on a busy box: the unrounded factors of `value` must be positive)
    assert out["value"] >= 0 and out["config"]["spmv_per_step"] > 0 and out["ms_per_step"] > 0
    assert out["unit"] == "GTEPS" and out["vs_baseline"] is None
    assert out["config"]["nnz"] == rmat_np.rmat_csr(11, 8, seed=0).nnz
    assert len(out["config"]["iterations_per_step"]) == 2
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"])


def test_bench_launcherless_spawns_its_ranks(oracle_build_dir):
    """`python bench.py --gpus 2` with NO launcher (the way the driver starts the N = 1 run): the parent process spawns the
    ranks as children, relays rank 0's single JSON line and exits with their status.  The N > 1 line carries parity (the
    partitioned path vs the oracle), a CPU baseline, the nnz balance of the partition and the same-graph single-GPU leg."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_dist_worker.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--scale", "11", "--ef", "8"]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["value"] >= 0 and out["config"]["spmv_per_step"] > 0 and out["ms_per_step"] > 0     # (see above)
    assert out["parity"]["rel_linf"] <= 1e-6 and out["parity"]["gpu_iterations"] == out["parity"]["cpu_iterations"]
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1
    assert out["config"]["nnz_per_rank_max_over_mean"] <= 1.10       # scale 11: 2 K rows, statistical balance
    same = out["same_graph_1gpu"]
    assert same["rel_linf_partitioned_vs_1gpu"] <= 1e-6 and same["iterations_1gpu"] == same["iterations_partitioned"]
    assert same["speedup_vs_1gpu_same_graph"] > 0
    # SURVEY.md 8e, last sentence: the multi-seed batch as a replica split beside the partitioned headline
    replicas = out["secondary"]["batch_of_64_seeds_replicas"]
    assert "error" not in replicas, replicas
    assert replicas["ranks"] == 2 and replicas["seed_sets"] == 2 * replicas["seed_sets_per_rank"] and replicas["edge_vector_products_per_s_G"] > 0


def _bench_ranks(world, extra_env, scale="11", launcher=True, timeout=900):
    import json
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    tail = [os.path.join(ROOT, "tests", "bench_dist_worker.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--scale", scale, "--ef", "8"]
    cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
            "--master-port", str(port)] if launcher else [sys.executable]) + tail
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", **extra_env)
    if not launcher:
        for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(key, None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    return res, [json.loads(ln) for ln in lines]


def test_bench_ladder_restarts_a_failed_rung_as_a_fresh_child_tree(oracle_build_dir):
    """VERDICT r5 item 1b.  The driver's N > 1 launch (torch.distributed.run -> bench.py): the launcher's ranks are supervisors that never
    touch the GPU and run the work as a child.  Rank 1's child of rung 0 ends with the watchdog's code 3 at its start while rank 0's child
    waits in its first collective: rank 0's supervisor ends that child (no waiting for a collective that cannot complete), both start a
    FRESH child tree one rung down (one communicator, one stream), and the line says which rung produced it."""
    res, lines = _bench_ranks(2, dict(PGH_BENCH_TEST_FAIL="0:1:3:start"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert len(lines) == 1, res.stdout
    rung = lines[0]["config"]["fallback_rung"]
    assert rung["rung"] == 1 and rung["env"] == dict(PGH_DIST_SINGLE_COMM="1", PGH_DIST_SINGLE_STREAM="1"), rung
    assert lines[0]["n_gpus"] == 2 and lines[0]["parity"]["rel_linf"] <= 1e-6
    # (rank 0's child is ended by its supervisor -- code 5 -- or, over gloo, sees the connection of the dead peer reset first)
    assert "rung 0 (default) failed on ranks" in res.stderr and "starting a fresh child tree one rung down" in res.stderr, res.stderr[-3000:]
    assert "rung 0 (default): child ended with code 3" in res.stderr, res.stderr[-3000:]


def test_bench_ladder_survives_a_hung_rank_and_reaches_the_last_rung(oracle_build_dir):
    """A rank that stops answering (a stalled collective) is ended at the rung's deadline together with its peers; two failed rungs lead
    to the last one: dense all-gather, the Python-driven loop, one communicator and one stream -- through the launcher-less entry."""
    res, lines = _bench_ranks(2, dict(PGH_BENCH_TEST_FAIL="0:0:hang:start,1:1:9:after_timing", PGH_BENCH_RUNG_S="25"), launcher=False)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert len(lines) == 1, res.stdout
    rung = lines[0]["config"]["fallback_rung"]
    assert rung["rung"] == 2 and rung["env"]["PGH_DIST_EXCHANGE"] == "allgather" and rung["env"]["PGH_DIST_NEED_LISTS"] == "0", rung
    assert "no result within 25 s" in res.stderr and "rung 1 (one communicator, one stream) failed on ranks" in res.stderr, res.stderr[-3000:]
    assert "incomplete" not in lines[0]["config"]


def test_bench_ladder_keeps_the_measured_headline_when_every_rung_fails_later(oracle_build_dir):
    """Every rung's child dies AFTER its timed region (in the legs that follow: roofline, parity, same graph, replicas): the headline
    the first rung measured is printed, marked incomplete, with exit code 0 -- a measurement is never thrown away."""
    res, lines = _bench_ranks(2, dict(PGH_BENCH_TEST_FAIL="0:0:9:after_timing,1:0:9:after_timing,2:0:9:after_timing"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert len(lines) == 1, res.stdout
    out = lines[0]
    assert out["config"]["fallback_rung"]["rung"] == 0 and "incomplete" in out["config"]
    assert out["n_gpus"] == 2 and out["config"]["spmv_per_step"] > 0 and out["ms_per_step"] > 0 and out["roofline"] is None


def test_bench_eight_ranks_gloo(oracle_build_dir):
    """The world == 8 branch of the bench (BASELINE.json configs[4]'s rank count) end to end on the CPU: eight supervised ranks,
    one block per rank, parity of the 8-way partition against the oracle, the same-graph leg, the replica split of 8 x 4 seed sets."""
    res, lines = _bench_ranks(8, {}, scale="12", timeout=1500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert len(lines) == 1, res.stdout
    out = lines[0]
    assert out["n_gpus"] == 8 and out["config"]["fallback_rung"]["rung"] == 0
    assert out["parity"]["rel_linf"] <= 1e-6 and out["parity"]["gpu_iterations"] == out["parity"]["cpu_iterations"]
    assert out["same_graph_1gpu"]["rel_linf_partitioned_vs_1gpu"] <= 1e-6
    assert out["secondary"]["batch_of_64_seeds_replicas"]["ranks"] == 8


@pytest.mark.parametrize("world", [2, 4])
def test_partitioned_upload_of_a_scipy_graph(tmp_path, oracle_build_dir, world):
    """pgh_graph_from_csr_part / distributed.partition_scipy: a caller's (weighted, non-power-of-two) scipy graph is relabelled
    identically on every rank, each rank uploads its rows of M^T only, and the partitioned PageRank equals the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker_csr
    port = 29700 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_csr.py"), str(tmp_path)]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    A, p = dist_worker_csr.make_graph()
    M = sp.csr_array(orc.normalize(A, "col", True))
    assert sum(int(part["nnz"]) for part in parts) == M.nnz            # every entry lives on exactly one rank
    assert len({int(part["n_local"]) for part in parts}) == 1 and int(parts[0]["n_pad"]) >= A.shape[0]
    for name, kw in (("l1", dict(error_type="l1", tol=1e-6, max_iters=500)), ("mabs", dict(error_type="mabs", tol=1e-7, max_iters=500))):
        want, want_iters = orc.pagerank(M, p, alpha=0.85, eps=EPS32, **kw)
        got = sum(part[name + "_ranks"] for part in parts)             # slices are disjoint
        assert all(int(part[name + "_iters"]) == want_iters for part in parts), name
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), name


def test_partition_balance_eight_ways(host_engine):
    """nnz per rank of the 8-way partition (configs[4] shape at a CPU-sized scale): max / mean <= 1.05.  The slices are equal in
    rows by construction; the entries follow because ids are dealt round-robin in descending order of their source counts."""
    from pygrank_amd.distributed import rmat_partitioned
    for world in (2, 4, 8):
        nnz = [rmat_partitioned(16, 16, r, world).graph.nnz for r in range(world)]
        assert max(nnz) <= 1.05 * np.mean(nnz), (world, nnz)
        assert sum(nnz) == rmat_np.rmat_csr(16, 16, seed=0).nnz


def test_randomised_partitions_against_the_oracle(oracle_build_dir):
    """tests/stress_partitioned.py for a few seconds with two ranks (gloo, host double, the Python-driven loop): random graphs (generated
    slices and callers' matrices with ragged id counts), layouts, filters, stopping rules and personalizations against the oracle on the
    un-partitioned graph.  (Its first minute found the Mabs rule dividing by the PADDED id count of a caller's matrix -- one iteration
    early on a 483-node graph; PartitionedGraph.n_nodes.)  The GPU run of the same harness drives the engine's own loop."""
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "stress_partitioned.py"), "--seconds", "12", "--seed", "5", "--max-scale", "11"]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", PGH_TEST_ENGINE="host", PGH_DIST_BACKEND="gloo")
    env.pop("PGH_DIST_NATIVE", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "partitioned stress ok" in res.stdout, res.stdout[-2000:]
    cases = int(res.stdout.split("partitioned stress ok:")[1].split()[0])
    assert cases >= 50, res.stdout[-500:]


def check_replica_split(parts, scale, ef, width):
    """What tests/dist_worker_replicas.py wrote against the oracle: every rank holds the same [n, width] result, column j of it is the
    oracle's PageRank of feature column j (<= 1e-6, the rank that ran it reports the oracle's iteration count), an all-zero column comes
    back as it went in (abstract_filters.py:53-54)."""
    import dist_worker_replicas
    world = len(parts)
    n = 1 << scale
    M = sp.csr_array(orc.normalize(rmat_np.rmat_csr(scale, ef, seed=0), "col", True))
    F = dist_worker_replicas.features(n, width)
    shares = [(width * r // world, width * (r + 1) // world) for r in range(world)]
    assert [(int(part["lo"]), int(part["hi"])) for part in parts] == shares
    for name, kw in (("l1", dict(error_type="l1", tol=1e-6, max_iters=500)), ("mabs", dict(error_type="mabs", tol=1e-7, max_iters=500))):
        whole = parts[0][name + "_ranks"]
        assert whole.shape == (n, width)
        for part in parts:
            assert np.array_equal(part[name + "_ranks"], whole), name          # the same bits on every rank after the all-gather
            assert int(part[name + "_share_equal"]) == 1
        for j in range(width):
            owner = [r for r, (lo, hi) in enumerate(shares) if lo <= j < hi][0]
            if not F[:, j].any():
                assert not whole[:, j].any()
                continue
            want, want_iters = orc.pagerank(M, F[:, j], alpha=0.85, eps=EPS32, **kw)
            assert np.max(np.abs(whole[:, j] - want)) <= 1e-6 * np.max(np.abs(want)), (name, j)
            assert int(parts[owner][name + "_iters"][j]) == want_iters, (name, j)


@pytest.mark.parametrize("world,width", [(2, 5), (4, 6), (4, 3)])
def test_replica_split_of_a_seed_batch_gloo(tmp_path, oracle_build_dir, world, width):
    """SURVEY.md 8e, last sentence: multi-seed batches split across the ranks with zero communication (every rank holds the whole
    graph) -- uneven shares, a rank without columns (width 3 on 4 ranks), an all-zero column."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    scale, ef = 10, 8
    port = 29600 + world * 7 + width + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_replicas.py"), str(tmp_path), str(scale), str(ef), str(width)]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    check_replica_split(parts, scale, ef, width)
