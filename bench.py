"""bench.py -- headline benchmark of the propagation path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (SURVEY.md 8d / BASELINE.json configs[1]): personalized PageRank, alpha = 0.85, to tol = 1e-6 with the
L1 residual (the "headline" stopping rule), on a synthetic RMAT graph (a, b, c, d) =
(0.57, 0.19, 0.19, 0.05), seed 0, duplicates summed, "col" normalisation, fp32 values, one personalization
vector of 100 seed nodes per step.  N = 1: scale 23, edge factor 16 (8.4 M nodes, ~131 M edges).
N > 1: the CSR is 1-D row-partitioned across the ranks (nnz-balanced), every iteration all-gathers the rank
vector over RCCL/xGMI; per-GPU work is held fixed (weak scaling): scale 23 + log2(N) at edge factor 16, and the
BASELINE.json configs[4] graph (scale 27, edge factor 8, ~1.07 B edges) at N = 8.

A "step" is one full PPR run (personalization resident in HBM -> converged ranks in HBM).  The timed region
covers exactly K steps; value = nnz * (SpMV launches of all steps) / time  [edges*iterations/s, reported in
GTEPS].  Extra objects on the JSON line: "roofline" (HIP-event time of EVERY launch of one PPR iteration -- block
partial sums, the cold image's phase A, phase B + epilogue, residual, close -- vs the 8*nnz + 16*n algorithmic
bytes of one iteration) and "cpu_baseline" (the oracle's scipy loop -- exactly the reference's numpy-backend
arithmetic -- on the same graph, one core; the all-core OpenMP pull kernel beside it), plus the in-run parity of
GPU vs CPU ranks.

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself: the parent (which never
touches the GPU) runs `python -m torch.distributed.run --nproc-per-node N <this script> ...` as a child process, relays
rank 0's JSON line and exits with the child's status.
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RMAT = dict(a=0.57, b=0.19, c=0.19)
ALPHA, TOL, MAX_ITERS, SEEDS = 0.85, 1e-6, 1000, 100
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
# HBM traffic of the step kernels: rocprofv3 --pmc passes of this same command (tools/gpu_bench_call.sh), summarised by
# tools/summarize_pmc.py with the guide's gfx950 corrections; counters cannot be read from inside the timed process
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06", "bench_n1_pmc.json")
# the launches of one PPR iteration, in stream order (HIP-event ids of include/pgh.h)
STEP_KERNELS = ("spmv", "fixup", "pb_gather", "pb_finish", "combine", "residual", "close")


def csrc_sha16():
    """Hash of the kernel sources: ties the committed PMC summary to the code that produced it."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "pygrank_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pygrank_amd", "csrc", "*.h"))):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def seeds_for(step, candidates, count=SEEDS):
    rng = np.random.default_rng(1 + step)           # RNG seed 1 (+ step for further draws)
    return np.sort(rng.choice(candidates, size=min(count, len(candidates)), replace=False))


def measured_traffic(scale, ef, blocked):
    """Per-iteration HBM bytes (read + written) of the PPR iteration's kernels from the committed PMC summary of this same
    command (counters cannot be read from inside the timed process), or None when the summary does not describe this
    workload or was collected with different kernel sources (it carries the hash of pygrank_amd/csrc it was made from)."""
    if not blocked or (scale, ef) != (23, 16) or not os.path.exists(PMC_SUMMARY):
        return None, None
    with open(PMC_SUMMARY) as f:
        pmc = json.load(f)
    if pmc.get("_meta", {}).get("csrc_sha16") != csrc_sha16():
        return None, "stale: " + os.path.relpath(PMC_SUMMARY, ROOT) + " was collected from other kernel sources"
    total = 0.0
    steps = max([row.get("dispatches", 0) for name, row in pmc.items() if name.startswith("k_bsf_partial")] or [0])
    for name, row in pmc.items():
        # the PPR iteration's launches: the AXPBY epilogue is MODE 1 (the PMC run also holds the secondary filters' <2,.> / <3,.>);
        # a kernel that does not run in every iteration (the separate residual: first step of a run only, the close: once per
        # run) counts with its share of the launches
        if name.startswith(("k_bsf_partial", "k_bsf_fixup", "k_bsf_combine<1,", "k_pb_gather", "k_pb_finish<1,", "k_step_residual", "k_step_close")):
            share = min(1.0, row.get("dispatches", steps) / steps) if steps else 1.0
            total += share * (row["hbm_read_bytes_corrected"] + row["hbm_write_bytes"])
    return (int(total), os.path.relpath(PMC_SUMMARY, ROOT)) if total > 0 else (None, None)


def measured_batch_traffic(scale, ef, width, batch_steps=11):
    """HBM bytes of one batch step of the multi-seed loop (k_mm_partial + fix-up + combine + residual) from the committed PMC
    summary of tools/probe_batch_kernels.py (profiles/r06/spmm_final_pmc.json; same hash rule as the headline's traffic)."""
    path = os.path.join(ROOT, "profiles", "r06", "spmm_final_pmc.json")
    if (scale, ef, width) != (23, 16, 64) or not os.path.exists(path):
        return {}
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get("_meta", {}).get("csrc_sha16") != csrc_sha16():
        return dict(measured_gb_per_step=None, measured_traffic_source="stale: " + os.path.relpath(path, ROOT))
    # per batch step: every launch of the loop's kernels in the profiled run over the steps that did work (the gather pass in its two
    # forms, the fix-up, the epilogue in its two forms, the separate residual of the first step, folds and closes)
    # (the profiled tool runs the batch `runs` times -- warm-up + timed -- and the run-ahead leaves a few no-op iterations behind every
    # stop: they move nothing)
    runs = max(int(pmc.get("k_mm_state_init", {}).get("dispatches", 1)), 1)
    working = max(int(batch_steps), 1) * runs
    total = sum((row["hbm_read_bytes_corrected"] + row["hbm_write_bytes"]) * row.get("dispatches", 1) for name, row in pmc.items()
                if name.startswith(("k_mm_partial", "k_mm_fixup", "k_mm_step", "k_mm_residual2", "k_mm_fold", "k_mm_close2")))
    return dict(measured_gb_per_step=round(total / working / 1e9, 2), measured_traffic_source=os.path.relpath(path, ROOT))


def build_profile(lib, L):
    """{phase: ms} of the last graph build (wall time per phase with the engine's stream drained behind each)."""
    buf = C.create_string_buffer(4096)
    L.check(lib.pgh_last_build_profile(buf, 4096))
    return {k: round(float(v), 2) for k, v in (item.split("=") for item in buf.value.decode().split(";") if item)}


def stream_ceiling_gbs(lib, L):
    """What a plain streaming copy reaches on this box right now (read + written bytes per second, engine's pgh_vec_copy on
    256 MB vectors): the practical HBM ceiling beside the 8 TB/s datasheet peak."""
    m = 1 << 26
    va, vb = L.c_vec(), L.c_vec()
    L.check(lib.pgh_vec_alloc(m, C.byref(va)))
    L.check(lib.pgh_vec_alloc(m, C.byref(vb)))
    L.check(lib.pgh_vec_fill(va, 1.0))
    t = L.c_timer()
    L.check(lib.pgh_timer_create(C.byref(t)))
    best = 0.0
    for _ in range(3):
        L.check(lib.pgh_vec_copy(vb, va))
    for _ in range(3):
        L.check(lib.pgh_timer_start(t))
        for _ in range(4):
            L.check(lib.pgh_vec_copy(vb, va))
        L.check(lib.pgh_timer_stop(t))
        ms = C.c_double()
        L.check(lib.pgh_timer_elapsed_ms(t, C.byref(ms)))
        best = max(best, 4 * 2 * 4.0 * m / (ms.value * 1e-3) / 1e9)
    L.check(lib.pgh_timer_destroy(t))
    L.check(lib.pgh_vec_free(va))
    L.check(lib.pgh_vec_free(vb))
    return round(best, 1)


def allcore_cpu_gteps(MT, nnz, repeats=3):
    """The oracle's OpenMP pull kernel over CSR(M^T) (oracle/spmv_oracle.c) on every host core: the "best-effort CPU" line
    of SURVEY.md 8d beside the reference's single-threaded scipy path."""
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_spmv.so")
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    n = MT.shape[0]
    indptr = np.ascontiguousarray(MT.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(MT.indices, dtype=np.int32)
    data = np.ascontiguousarray(MT.data, dtype=np.float64)
    x = np.full(MT.shape[1], 1.0 / max(MT.shape[1], 1))
    y = np.empty(n)
    args = (C.c_int64(n), indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
            x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    lib.oracle_pull_spmv_omp(*args)
    t0 = time.perf_counter()
    for _ in range(repeats):
        lib.oracle_pull_spmv_omp(*args)
    dt = (time.perf_counter() - t0) / repeats
    return dict(value=round(nnz / dt / 1e9, 3), unit="GTEPS", cores=os.cpu_count(), kind="port",
                sample=f"{repeats} SpMV of the same graph, OpenMP pull kernel over CSR(M^T), fp64")


def single_gpu(args):
    import pygrank_amd as pg
    from pygrank_amd import _lib as L
    from pygrank_amd.synthetic import rmat_graph
    pg.load_backend("hip")
    lib = L.lib()
    scale, ef = args.scale, args.ef
    t0 = time.time()
    adj = rmat_graph(scale, ef, seed=0, normalization="col", **RMAT)
    L.check(lib.pgh_sync())
    build_s = time.time() - t0
    build_ms = build_profile(lib, L)                 # where that first build of the process spent its time (pgh_last_build_profile)
    # ... and the same build again: what a rank() pays from its second call on when the caller does not promise immutability
    # (the reference's default: every rank() re-normalises and re-uploads, pygrank/core/utils/preprocessing.py:233-287)
    t0 = time.time()
    again = rmat_graph(scale, ef, seed=0, normalization="col", **RMAT)
    L.check(lib.pgh_sync())
    rebuild_s = time.time() - t0
    del again
    g = adj.array
    n, nnz = g.shape[0], g.nnz
    out_deg = np.asarray(pg.degrees(g))              # row sums of M: > 0 <=> out-degree > 0
    candidates = np.flatnonzero(out_deg > 0)
    total = args.warmup + args.steps
    personalizations = []
    for step in range(total):                        # inputs resident in HBM before the timed region
        p = np.zeros(n)
        p[seeds_for(step, candidates)] = 1.0
        personalizations.append(pg.to_signal(adj, p))
    ranker = pg.PageRank(alpha=ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS)

    def run(step):
        ranks = ranker.rank(adj, personalizations[step])
        return ranks, ranker.last_loop

    for step in range(args.warmup):
        run(step)
    L.check(lib.pgh_sync())
    spmv_total, loop_ms_total, iters, paused = 0, 0.0, [], 0
    t0 = time.perf_counter()
    for step in range(args.warmup, total):
        ranks, info = run(step)
        spmv_total += info["spmv"]
        loop_ms_total += info["loop_ms"]
        iters.append(info["iterations"])
        paused += info.get("flags", 0) & 1
    L.check(lib.pgh_sync())
    elapsed = time.perf_counter() - t0
    gteps = nnz * spmv_total / elapsed / 1e9

    # ---- roofline leg: HIP events around every launch of the dominant kernel (one extra, untimed run)
    L.check(lib.pgh_profile_reset())
    L.check(lib.pgh_profile_enable(1))
    _, prof_info = run(total - 1)
    L.check(lib.pgh_profile_enable(0))
    prof = {}
    for kid, name in ((L.K_SPMV, "spmv"), (L.K_FIXUP, "fixup"), (L.K_PB_GATHER, "pb_gather"), (L.K_PB_ACCUM, "pb_finish"),
                      (L.K_COMBINE, "combine"), (L.K_RESIDUAL, "residual"), (L.K_FINAL, "close")):
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
        prof[name] = dict(launches=cnt.value, avg_us=(ms.value / cnt.value * 1e3) if cnt.value else None, total_us=ms.value * 1e3)
    alg_bytes = 8 * nnz + 16 * n                     # PPR step + quotient + residual, per iteration
    # One PPR iteration = every launch between two iterates: block partial sums, (cross-tile fix-up,) the cold image's
    # phase A, phase B + epilogue (or the combine of graphs without a cold image), residual, close.  Its duration is the sum
    # of their HIP-event times per iteration that did work = per SpMV of the profiled run: the <= 2 no-op iterations the
    # run-ahead loop leaves behind after convergence are charged to the working ones, and the close of a step rides in the
    # first kernel of the next one on the blocked layout (k_step_close then runs once per run, not once per iteration).
    step_kernels = [k for k in STEP_KERNELS if prof[k]["avg_us"]]
    iterations = max(int(prof_info["spmv"]), 1)
    step_us = sum(prof[k]["total_us"] for k in step_kernels) / iterations
    achieved = alg_bytes / (step_us * 1e-6) / 1e9
    blocked = g.format().startswith("bsf")
    traffic, traffic_source = measured_traffic(scale, ef, blocked)
    ceiling = stream_ceiling_gbs(lib, L)
    names = dict(spmv="k_bsf_partial" if blocked else "k_spmv_merge<AXPBY>", fixup="k_bsf_fixup" if blocked else "k_spmv_fixup",
                 pb_gather="k_pb_gather", pb_finish="k_pb_finish<AXPBY>", combine="k_bsf_combine<AXPBY>", residual="k_step_residual",
                 close="k_step_close")
    roofline = dict(bound="hbm", kernel=" + ".join(names[k] for k in step_kernels) + " (one PPR iteration: step + quotient + residual)",
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                    traffic=traffic, traffic_source=traffic_source, algorithmic_bytes_per_launch=alg_bytes,
                    avg_launch_us=round(step_us, 2), measured_copy_gbs=ceiling,
                    frac_of_measured_copy=round(achieved / ceiling, 4) if ceiling else None,
                    # bytes the counters saw (not the nominal ones) over the same time, against the measured copy rate
                    traffic_frac_of_measured_copy=round(traffic / (step_us * 1e-6) / 1e9 / ceiling, 4) if traffic and ceiling else None,
                    format=g.format(),
                    kernels_avg_us={k: (round(v["avg_us"], 2) if v["avg_us"] else None) for k, v in prof.items()},
                    launches_per_iteration={k: round(v["launches"] / iterations, 3) for k, v in prof.items() if v["launches"]})

    # ---- the other filters of the path on the same resident graph (8 nnz + 20 n per polynomial term,
    # 8 nnz + 24 n per absorbing step); reported beside the headline, not part of `value`
    secondary = {}
    if not args.no_secondary:
        def side(label, other, graph, per_step, edges, runs=3, signals=None):
            signals = personalizations if signals is None else signals
            other.rank(graph, signals[0])
            L.check(lib.pgh_sync())
            t1 = time.perf_counter()
            count, loop = 0, 0.0
            for step in range(runs):
                other.rank(graph, signals[step % len(signals)])
                count += other.last_loop["spmv"]
                loop += other.last_loop["loop_ms"]
            L.check(lib.pgh_sync())
            dt = time.perf_counter() - t1
            secondary[label] = dict(gteps=round(edges * count / dt / 1e9, 2), device_step_us=round(loop / count * 1e3, 1),
                                    nominal_gbs=round(per_step / (loop / count * 1e-3) / 1e9, 1), spmv_per_run=count // runs,
                                    iterations=int(other.convergence.iteration))
        # the three stopping rules (the headline above is (ii)), both polynomial recurrences, the absorbing
        # walk, and the symmetrised graph; 8 nnz + 20 n per polynomial term, 8 nnz + 24 n per absorbing step
        side("ppr_mabs_default_tol1e-6", pg.PageRank(alpha=ALPHA, tol=TOL, max_iters=MAX_ITERS), adj, alg_bytes, nnz)
        side("ppr_50_iterations", pg.PageRank(alpha=ALPHA, error_type="iters", max_iters=51), adj, alg_bytes, nnz)
        side("heat_kernel_t5_31_iterations", pg.HeatKernel(5, error_type="iters", max_iters=31), adj, 8 * nnz + 20 * n, nnz)
        side("heat_kernel_t5_31_iterations_chebyshev", pg.HeatKernel(5, coefficient_type="chebyshev", error_type="iters", max_iters=31),
             adj, 8 * nnz + 20 * n, nnz)
        side("absorbing_walks_a085_l1_1e-6", pg.AbsorbingWalks(ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS), adj,
             8 * nnz + 24 * n, nnz)
        # a tolerance below fp32 eps: f64 iterates on the f64 image, chosen by the filter itself (the reference's iteration counts)
        side("ppr_l1_1e-9_f64_iterates", pg.PageRank(alpha=ALPHA, error_type=pg.L1, tol=1e-9, max_iters=MAX_ITERS), adj, 8 * nnz + 32 * n, nnz, runs=2)
        # The backend-primitive route -- what north_star names ("plugs in as a pygrank.core.backends module so PageRank / HeatKernel /
        # AbsorbingWalks ... are unchanged"): the filters reach the engine ONE PRIMITIVE AT A TIME (conv, *, +, sum, /, abs, -; pygrank/core/
        # backend/__init__.py:59-80), the whole-loop entry points are switched off.  Lazy vectors (pygrank_amd/device.py) keep the iterate in
        # the engine's id space: one resident step per formula, one residual launch, two scalars to the host per iteration.  The result is
        # looked at (the way out of the id space) inside the timed region.
        def primitives(label, make, per_step):
            other = make()
            other._fused_loop = lambda *a, **k: False
            other._fused_rank = lambda *a, **k: None
            np.asarray(other.rank(adj, personalizations[0]).np[0])
            L.check(lib.pgh_sync())
            t1 = time.perf_counter()
            steps = 0
            runs = 3
            for step in range(runs):
                out = other.rank(adj, personalizations[step % len(personalizations)])
                float(out.np[0])                         # somebody looks at the ranks
                steps += int(other.convergence.iteration) - 1
            L.check(lib.pgh_sync())
            dt = time.perf_counter() - t1
            secondary[label] = dict(gteps=round(nnz * steps / dt / 1e9, 2), wall_step_us=round(dt / steps * 1e6, 1),
                                    nominal_gbs=round(per_step / (dt / steps) / 1e9, 1), spmv_per_run=steps // runs,
                                    iterations=int(other.convergence.iteration), route="one engine call per backend primitive (lazy vectors)")
        try:
            primitives("ppr_l1_1e-6_backend_primitives", lambda: pg.PageRank(alpha=ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS), alg_bytes)
            primitives("heat_kernel_t5_31_iterations_backend_primitives", lambda: pg.HeatKernel(5, error_type="iters", max_iters=31), 8 * nnz + 20 * n)
            primitives("absorbing_walks_a085_l1_1e-6_backend_primitives",
                       lambda: pg.AbsorbingWalks(ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS), 8 * nnz + 24 * n)
            secondary["ppr_l1_1e-6_backend_primitives"]["vs_fused_route"] = round(secondary["ppr_l1_1e-6_backend_primitives"]["gteps"] / gteps, 3)
        except Exception as exc:                     # a side measurement never takes the headline down
            secondary["ppr_l1_1e-6_backend_primitives"] = dict(error=str(exc)[:300])
        # configs[2]: 64 personalizations at once (NodeRanking.propagate -> pgh_ppr_run_batch); edge-vector products per second,
        # nominal bytes 8 nnz + 4 n + 12 n b per batch step
        try:
            from pygrank_amd.device import DeviceMatrix
            width = 64
            feats = DeviceMatrix.empty(n, width)
            for j in range(width):
                col = np.zeros(n)
                col[seeds_for(100 + j, candidates)] = 1.0
                feats.set_column(j, pg.to_signal(adj, col).np)
            batch_ranker = pg.PageRank(alpha=ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS)
            batch_ranker.propagate(adj, feats)
            L.check(lib.pgh_sync())
            t1 = time.perf_counter()
            batch_ranker.propagate(adj, feats)
            L.check(lib.pgh_sync())
            dt = time.perf_counter() - t1
            info = batch_ranker.last_batches[0]
            products = sum(c["spmv"] for c in info)
            steps = max(c["spmv"] for c in info)
            secondary["ppr_l1_1e-6_batch_of_64_seeds"] = dict(
                edge_vector_products_per_s_G=round(nnz * products / dt / 1e9, 1), device_step_us=round(info[0]["loop_ms"] / steps * 1e3, 1),
                nominal_gbs=round((8 * nnz + 4 * n + 12 * n * width) / (info[0]["loop_ms"] / steps * 1e-3) / 1e9, 1),
                batch_steps=steps, width=width, **measured_batch_traffic(scale, ef, width, steps))
            del feats
        except Exception as exc:                     # a side measurement never takes the headline down
            secondary["ppr_l1_1e-6_batch_of_64_seeds"] = dict(error=str(exc))
        # real-valued edge weights (nx ... weight="weight", preprocessing.py:103): the same structure with weights in [0.1, 1.1),
        # "col" normalisation on the device -> the VALUED stream (2-byte index + f32 value per entry); nominal bytes as the headline
        if not getattr(args, "no_weighted", False):
            try:
                import scipy.sparse as sp
                MTw = g.download_transposed()
                MTw = sp.csr_array((np.random.default_rng(7).random(MTw.nnz) + 0.1, MTw.indices, MTw.indptr), shape=MTw.shape)
                Ww = sp.csr_array(MTw.T)
                Ww.sort_indices()
                del MTw
                wadj = pg.preprocessor(normalization="col", assume_immutability=True)(pg.AdjacencyWrapper(Ww, directed=True))
                del Ww
                side("ppr_l1_1e-6_real_weights", pg.PageRank(alpha=ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS), wadj, alg_bytes, nnz,
                     signals=[pg.to_signal(wadj, sig.np) for sig in personalizations[:3]])
                secondary["ppr_l1_1e-6_real_weights"]["format"] = next((part.strip() for part in wadj.array.format().split(",") if "B/edge" in part),
                                                                       wadj.array.format()[:60])
                del wadj
            except Exception as exc:
                secondary["ppr_l1_1e-6_real_weights"] = dict(error=str(exc)[:300])
        if not args.no_symmetric:
            sym = rmat_graph(scale, ef, seed=0, symmetrize=True, **RMAT)          # A + A^T, "symmetric" normalisation
            nnz_s = sym.array.nnz
            side("ppr_l1_1e-6_symmetrised_graph", pg.PageRank(alpha=ALPHA, error_type=pg.L1, tol=TOL, max_iters=MAX_ITERS), sym,
                 8 * nnz_s + 16 * n, nnz_s, signals=[pg.to_signal(sym, sig.np) for sig in personalizations[:3]])
            del sym

    # ---- CPU baseline + parity: the oracle's scipy loop (= reference numpy backend), same graph, same seeds
    cpu = None
    parity = None
    if not args.no_cpu:
        import scipy.sparse as sp
        from oracle import ref_loops as orc
        MT = g.download_transposed()
        M = sp.csr_array(MT.T.astype(np.float64))     # fp32-rounded values of the same normalised matrix
        p = np.asarray(personalizations[total - 1].np, dtype=np.float64)
        t1 = time.perf_counter()
        want, cpu_iters = orc.pagerank(M, p, alpha=ALPHA, error_type="l1", tol=TOL, max_iters=MAX_ITERS)
        cpu_s = time.perf_counter() - t1
        got = np.asarray(run(total - 1)[0].np, dtype=np.float64)
        cpu = dict(value=round(nnz * (cpu_iters - 1) / cpu_s / 1e9, 4), unit="GTEPS", cores=1, kind="port",
                   sample=f"1 full PPR run ({cpu_iters - 1} SpMV) on the same scale-{scale} graph, scipy x @ M fp64 "
                          f"single thread, {cpu_s:.1f} s",
                   all_cores=allcore_cpu_gteps(MT, nnz))
        parity = dict(rel_linf=float(np.max(np.abs(got - want)) / np.max(np.abs(want))), gpu_iterations=int(iters[-1]),
                      cpu_iterations=int(cpu_iters), bound=1e-6)
    return dict(
        metric="edges*iters/sec (GTEPS) for PPR alpha=0.85 to tol=1e-6", value=round(gteps, 2), unit="GTEPS", n_gpus=1,
        steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True,
        scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload=f"single-GPU PPR on RMAT scale-{scale} ef-{ef} (BASELINE.json configs[1])", n=n, nnz=nnz,
                    alpha=ALPHA, tol=TOL, error_type="L1", seeds=SEEDS, iterations_per_step=iters,
                    spmv_per_step=spmv_total / args.steps, device_loop_ms_per_step=round(loop_ms_total / args.steps, 4),
                    graph_build_s=round(build_s, 3), graph_rebuild_s=round(rebuild_s, 3), graph_build_ms=build_ms, parallelism="1 GPU",
                    runs_with_a_paused_in_kernel_residual=paused),
        roofline=roofline, cpu_baseline=cpu, parity=parity, secondary=secondary)


_REAL_STDOUT = None


def _stdout_to_stderr():
    """RCCL prints its version banner and warnings on fd 1.  The contract is ONE JSON line on stdout, so during the
    multi-GPU run fd 1 points at stderr; the JSON goes to the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def multi_gpu(args):
    os.environ["NCCL_DEBUG"] = os.environ.get("PGH_NCCL_DEBUG", "WARN")
    _stdout_to_stderr()
    # A rank that waits on a peer for ever would leave the driver without a line AND without an exit code: the whole N-GPU
    # run is bounded (PGH_BENCH_WATCHDOG_S, default 1500 s); past it the rank says so and ends its process with code 3
    # (pygrank_amd.distributed bounds its own waits on collectives the same way, PGH_DIST_TIMEOUT_S).
    import threading

    def _expired():
        sys.stderr.write(f"[bench] rank {os.environ.get('RANK', '0')}: the N-GPU run did not finish within its watchdog; exiting with code 3 "
                         "(retry with PGH_DIST_SINGLE_COMM=1 PGH_DIST_SINGLE_STREAM=1)\n")
        sys.stderr.flush()
        os._exit(3)
    watchdog = threading.Timer(float(os.environ.get("PGH_BENCH_WATCHDOG_S", "1500")), _expired)
    watchdog.daemon = True
    watchdog.start()
    try:
        return bench_row_partitioned(args, RMAT, ALPHA, TOL, MAX_ITERS, SEEDS, HBM_PEAK_GBS)
    finally:
        watchdog.cancel()


# ----------------------------------------------------------------------------------------------------------------
# --gpus N: the row-partitioned path (pygrank_amd.distributed), its parity leg against the oracle and the same graph on ONE GPU
# ----------------------------------------------------------------------------------------------------------------
def bench_row_partitioned(args, rmat, alpha, tol, max_iters, num_seeds, hbm_peak):
    import math
    import torch
    import torch.distributed as dist
    from pygrank_amd import _lib as L
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import DistributedPageRank, rmat_partitioned
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # PGH_DIST_BACKEND=gloo with several ranks on ONE GPU: a functional test of the multi-rank device path (streams,
    # events, in-place collectives on device scalars) on boxes with a single GPU; never a measurement
    device_index = local_rank % torch.cuda.device_count() if use_cuda else 0
    if use_cuda:
        torch.cuda.set_device(device_index)
    L.ensure_init(device_index)
    if not dist.is_initialized():
        dist.init_process_group(backend=os.environ.get("PGH_DIST_BACKEND", "nccl" if use_cuda else "gloo"))
    if args.scale is None:                      # weak scaling: fixed edges per GPU; configs[4] at 8 GPUs
        scale, ef = (27, 8) if world == 8 else (23 + int(math.log2(world)), 16)
    else:
        scale, ef = args.scale, (args.ef or 16)
    if args.ef is not None and args.scale is None:
        ef = args.ef
    _test_failure("start", rank)
    t0 = time.time()
    pg = rmat_partitioned(scale, ef, rank, world, **rmat)
    L.check(L.lib().pgh_sync())
    build_s = time.time() - t0
    n, n_local, lo = pg.n, pg.n_local, pg.row_begin
    nnz_t = torch.tensor([pg.graph.nnz], dtype=torch.float64, device="cuda" if use_cuda else "cpu")
    dist.all_reduce(nnz_t)
    nnz_total = int(nnz_t.item())
    nnz_m = torch.tensor([float(pg.graph.nnz)], dtype=torch.float64, device="cuda" if use_cuda else "cpu")
    dist.all_reduce(nnz_m, op=dist.ReduceOp.MAX)
    balance = round(float(nnz_m.item()) * world / max(nnz_total, 1), 4)          # max over ranks / mean
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
    rung = dict(rung=int(os.environ.get("PGH_BENCH_RUNG", "0")), settings=os.environ.get("PGH_BENCH_RUNG_LABEL", "default"),
                env={k: os.environ[k] for k in ("PGH_DIST_SINGLE_COMM", "PGH_DIST_SINGLE_STREAM", "PGH_DIST_EXCHANGE", "PGH_DIST_NEED_LISTS",
                                                "PGH_DIST_NATIVE") if k in os.environ})
    headline = dict(
        metric="edges*iters/sec (GTEPS) for PPR alpha=0.85 to tol=1e-6", value=round(nnz_total * spmv_total / elapsed / 1e9, 2),
        unit="GTEPS", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload=f"row-partitioned PPR on RMAT scale-{scale} ef-{ef} over {world} GPUs (BASELINE.json configs[4] shape)",
                    n=n, nnz=nnz_total, alpha=alpha, tol=tol, error_type="L1", seeds=num_seeds, iterations_per_step=iters,
                    spmv_per_step=spmv_total / args.steps, graph_build_s=round(build_s, 2), nnz_per_rank_max_over_mean=balance,
                    fallback_rung=rung, **dict(ranker.exchange)))
    # the timed region is over on every rank: what has been measured is kept where the rank's supervisor finds it, whatever happens in
    # the legs below (supervise_rank prints it, marked incomplete, when no rung of the ladder gets through all of them)
    if rank == 0 and os.environ.get("PGH_BENCH_PARTIAL"):
        _write_atomic(os.environ["PGH_BENCH_PARTIAL"], json.dumps(dict(headline, roofline=None, cpu_baseline=None)))
    _test_failure("after_timing", rank)
    # roofline leg on rank 0: HIP-event time of the step kernels, per-GPU algorithmic bytes
    lib = L.lib()
    L.check(lib.pgh_profile_reset())
    L.check(lib.pgh_profile_enable(1))
    ranker.rank(pg, personalizations[total - 1])
    L.check(lib.pgh_profile_enable(0))
    prof = {}
    for kid, name in ((L.K_SPMV, "spmv"), (L.K_PB_GATHER, "pb_gather"), (L.K_PB_ACCUM, "pb_finish"), (L.K_FIXUP, "fixup"),
                      (L.K_COMBINE, "combine"), (L.K_RESIDUAL, "residual"), (L.K_FINAL, "close")):
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
        if name == "spmv":
            steps_profiled = max(cnt.value, 1)
        prof[name] = (ms.value / steps_profiled * 1e3) if cnt.value else None     # us per iteration (a kind may launch twice)
    # every launch of one iteration counts (as in the single-GPU line): step + residual + the scalar folds / closes
    step_us = sum(v for v in prof.values() if v)
    names = dict(spmv="k_bsf_partial", pb_gather="k_pb_gather", pb_finish="k_pb_finish<AXPBY>", fixup="k_bsf_fixup",
                 combine="k_bsf_combine<AXPBY>", residual="k_step_residual (first step only with the in-kernel residual)",
                 close="k_dist_fold4 + k_dist_close_fused (k_dist_close_sum + k_dist_fold + k_dist_close_err without the in-kernel residual)")
    step_kernels = " + ".join(names[k] for k, v in prof.items() if v)
    alg_bytes = 8 * pg.graph.nnz + 4 * n + 16 * n_local
    achieved = alg_bytes / (step_us * 1e-6) / 1e9 if step_us else None
    # ---- parity + CPU baseline of the N-rank line: the same partitioned code path on a graph the oracle finishes in
    # seconds (every rank takes part; rank 0 compares the un-permuted slices with the oracle's scipy loop and times it)
    parity, cpu = None, None
    if not args.no_cpu:
        parity, cpu = _parity_leg(dist, rank, world, min(scale, 20), ef, rmat, alpha, tol, max_iters, num_seeds, use_cuda)
    # what the line says about this rank's slice, read BEFORE the same-graph leg gives the slice up
    graph_format = pg.graph.format()
    nnz_local = pg.graph.nnz
    exchange = dict(ranker.exchange)
    # ---- the same graph on ONE GPU (rank 0, single-GPU engine) while the other ranks wait: the same-graph speed-up the
    # north star quotes (>= 6x at 8 GPUs on the 1 B-edge graph) next to the weak-scaling value, and a full-size cross-check
    same_graph = None
    if world > 1 and not getattr(args, "no_same_graph", False):
        same_graph = _same_graph_leg(dist, rank, world, pg, ranker, personalizations[total - 1], scale, ef, rmat, alpha, tol,
                                     max_iters, num_seeds, total, use_cuda, nnz_total * spmv_total / elapsed / 1e9)
    # ---- configs[2] across the ranks as a REPLICA SPLIT (last sentence): every rank holds the whole scale-23 graph
    # and runs 64 of the 64 x N seed sets; nothing is exchanged.  Last of all: every rank gives its slice up first.
    replicas = None
    if not args.no_secondary:
        replicas = _replica_batch_leg(dist, rank, world, pg, ranker, min(scale, 23), 16 if scale >= 23 else ef, rmat, alpha, tol, max_iters,
                                      use_cuda)
    if rank != 0:
        return None
    headline["config"].update(
        parallelism=f"1-D row partition x{world}, all-gather of the gather vector + " +
                    ("ONE 4-scalar all-reduce" if exchange.get("in_kernel_residual") else "2 scalar all-reduces") + " per iteration",
        engine_loop_probe=_probe_verdict(world, rank), **exchange)
    return dict(
        headline,
        roofline=dict(bound="hbm", kernel=step_kernels + " (one PPR iteration of rank 0's slice, exchange excluded)",
                      achieved=round(achieved, 1) if achieved else None, peak=hbm_peak, unit="GB/s",
                      frac=round(achieved / hbm_peak, 4) if achieved else None, traffic=None,
                      algorithmic_bytes_per_launch=alg_bytes, avg_launch_us=round(step_us, 2), format=graph_format,
                      kernels_avg_us=prof),
        cpu_baseline=cpu, parity=parity, same_graph_1gpu=same_graph, secondary=dict(batch_of_64_seeds_replicas=replicas))


def _test_failure(where, rank):
    """Test hook of the supervisors' ladder (tests/test_distributed_cpu.py): PGH_BENCH_TEST_FAIL="rung:rank:code:where[,...]" ends this
    rank's child with `code` at the named point of the named rung ("hang": it stops answering instead)."""
    for spec in filter(None, os.environ.get("PGH_BENCH_TEST_FAIL", "").split(",")):
        f_rung, f_rank, f_code, f_where = spec.split(":")
        if int(f_rung) == int(os.environ.get("PGH_BENCH_RUNG", "0")) and int(f_rank) == rank and f_where == where:
            sys.stderr.write(f"[bench] rank {rank}: test hook: failing {where} of rung {f_rung} with {f_code}\n")
            sys.stderr.flush()
            if f_code == "hang":
                time.sleep(10 ** 6)
            os._exit(int(f_code))


def _probe_verdict(world, rank):
    """What pygrank_amd.distributed found when it probed the engine-driven RCCL loop on a small graph before trusting it with N > 1
    ("ok", what failed -- the run then used the Python-driven loop -- or "not run": one rank, PGH_DIST_NATIVE=0/1, gloo)."""
    from pygrank_amd.distributed import PREFLIGHT
    return str(PREFLIGHT.get((world, rank), "not run"))


def _device_of(use_cuda):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")


def _gather_slices(dist, local, world, use_cuda):
    """Every rank's equal-sized f32 slice -> one array on every rank (new id order)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32)).to(_device_of(use_cuda))
    out = torch.empty(t.numel() * world, dtype=torch.float32, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


def _parity_leg(dist, rank, world, scale, ef, rmat, alpha, tol, max_iters, num_seeds, use_cuda):
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import DistributedPageRank, rmat_partitioned
    from oracle import ref_loops as orc, rmat_np          # the checker: imported by the bench's parity leg only
    import scipy.sparse as sp
    pgp = rmat_partitioned(scale, ef, rank, world, **rmat)
    perm = pgp.perm
    A = rmat_np.rmat_csr(scale, ef, seed=0) if rank == 0 else None
    # personalization on ORIGINAL ids, identical on every rank (the seeds are drawn from a rank-independent rule)
    rng = np.random.default_rng(1)
    p_old = np.zeros(pgp.n)
    p_old[rng.choice(pgp.n, size=min(num_seeds, pgp.n), replace=False)] = 1.0
    lo = pgp.row_begin
    ranker = DistributedPageRank(alpha=alpha, tol=tol, error_type="l1", max_iters=max_iters)
    out = ranker.rank(pgp, DeviceVector.from_host(p_old[perm[lo:lo + pgp.n_local]]))
    new_order = _gather_slices(dist, np.asarray(out), world, use_cuda)
    if rank != 0:
        return None, None
    got = np.zeros(pgp.n)
    got[perm] = new_order
    M = sp.csr_array(orc.normalize(A, "col", True))
    t0 = time.perf_counter()
    want, cpu_iters = orc.pagerank(M, p_old, alpha=alpha, error_type="l1", tol=tol, max_iters=max_iters,
                                   eps=float(np.finfo(np.float32).eps))
    cpu_s = time.perf_counter() - t0
    parity = dict(rel_linf=float(np.max(np.abs(got - want)) / np.max(np.abs(want))), gpu_iterations=int(ranker.iteration),
                  cpu_iterations=int(cpu_iters), bound=1e-6,
                  workload=f"the same {world}-rank partitioned path on RMAT scale-{scale} ef-{ef} vs the oracle's scipy loop")
    cpu = dict(value=round(M.nnz * (cpu_iters - 1) / cpu_s / 1e9, 4), unit="GTEPS", cores=1, kind="port",
               sample=f"1 full PPR run ({cpu_iters - 1} SpMV) on the RMAT scale-{scale} ef-{ef} graph of the parity leg, scipy x @ M "
                      f"fp64 single thread, {cpu_s:.2f} s")
    return parity, cpu


def _same_graph_leg(dist, rank, world, pg, ranker, p_last, scale, ef, rmat, alpha, tol, max_iters, num_seeds, total, use_cuda,
                    value_n):
    """Rank 0 runs the same graph / same personalization on its GPU alone (single-GPU engine) while the other ranks wait;
    every rank contributes its slice of the partitioned result and of the personalization for a full-size comparison.
    Call it LAST: rank 0 gives up its slice (graph image, exchange buffers) before it builds the whole graph, so the two
    never sit in HBM together."""
    from pygrank_amd import _lib as L
    out = ranker.rank(pg, p_last)                           # the partitioned result of the last personalization, again
    full_new = _gather_slices(dist, np.asarray(out), world, use_cuda)          # new id order
    p_full_new = _gather_slices(dist, np.asarray(p_last), world, use_cuda)
    iterations_partitioned = int(ranker.iteration)
    result = None
    if rank == 0:
        try:
            import gc
            import pygrank_amd as pgm
            from pygrank_amd.synthetic import rmat_graph
            perm = np.array(pg.perm)                        # host copy: the slice goes away next
            n_all = pg.n
            ranker._buffers = None
            ranker._buffers_for = None
            from pygrank_amd.distributed import release_native_comms
            release_native_comms()                          # the engine's RCCL communicators hold the run's gather buffers
            pg.graph.destroy()
            del out
            gc.collect()
            if use_cuda:
                import torch
                torch.cuda.empty_cache()
            pgm.load_backend("hip")
            adj = rmat_graph(scale, ef, seed=0, normalization="col", **rmat)
            p_old = np.zeros(n_all)
            p_old[perm] = p_full_new                        # back to ORIGINAL ids
            sig = pgm.to_signal(adj, p_old)
            rk = pgm.PageRank(alpha=alpha, error_type=pgm.L1, tol=tol, max_iters=max_iters)
            rk.rank(adj, sig)                               # warm-up
            L.check(L.lib().pgh_sync())
            t0 = time.perf_counter()
            spmv, runs, ranks1 = 0, 3, None
            for _ in range(runs):
                ranks1 = rk.rank(adj, sig)
                spmv += rk.last_loop["spmv"]
            L.check(L.lib().pgh_sync())
            dt = time.perf_counter() - t0
            one = adj.array.nnz * spmv / dt / 1e9
            got = np.zeros(n_all)
            got[perm] = full_new
            ref = np.asarray(ranks1.np, dtype=np.float64)
            result = dict(gteps_1gpu=round(one, 2), speedup_vs_1gpu_same_graph=round(value_n / one, 3) if one > 0 else None,
                          iterations_1gpu=int(rk.convergence.iteration), iterations_partitioned=iterations_partitioned,
                          rel_linf_partitioned_vs_1gpu=float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))),
                          workload=f"the bench graph (RMAT scale-{scale} ef-{ef}) on rank 0's GPU alone, same personalization")
        except Exception as exc:                            # the line then SAYS that the number is missing, and why
            result = dict(error=str(exc)[:300], speedup_vs_1gpu_same_graph=None)
    dist.barrier()
    return result


def _replica_batch_leg(dist, rank, world, pg, ranker, scale, ef, rmat, alpha, tol, max_iters, use_cuda):
    """NodeRanking.propagate (pygrank/core/signals.py:225-226) of 64 x N seed sets, 64 per rank, every rank on its own copy of
    the whole graph (pygrank_amd.distributed.ReplicatedPropagation; zero communication).  Whole-job edge-vector products per
    second over the slowest rank's time.  A failure is reported on the line, it never takes the headline down."""
    import torch
    result = None
    try:
        import gc
        import pygrank_amd as pgm
        from pygrank_amd import _lib as L
        from pygrank_amd.device import DeviceMatrix
        from pygrank_amd.distributed import ReplicatedPropagation, release_native_comms
        from pygrank_amd.synthetic import rmat_graph
        ranker._buffers = None
        ranker._buffers_for = None
        release_native_comms()
        if pg.graph._h is not None:
            pg.graph.destroy()
        gc.collect()
        if use_cuda:
            torch.cuda.empty_cache()
        pgm.load_backend("hip")
        adj = rmat_graph(scale, ef, seed=0, normalization="col", **rmat)
        g = adj.array
        n, nnz = g.shape[0], g.nnz
        candidates = np.flatnonzero(np.asarray(pgm.degrees(g)) > 0)
        per_rank = 64 if scale >= 16 else 4                       # (CPU tests of the contract run a scale-11 graph on the host double)
        width = per_rank * world
        feats = DeviceMatrix.empty(n, width)                      # the same [n, 64 N] feature matrix on every rank
        for j in range(width):
            col = np.zeros(n)
            col[seeds_for(100 + j, candidates)] = 1.0
            feats.set_column(j, pgm.to_signal(adj, col).np)
        split = ReplicatedPropagation(pgm.PageRank(alpha=alpha, error_type=pgm.L1, tol=tol, max_iters=max_iters))
        split.propagate(adj, feats, gather=False)                 # warm-up: builds the multi-seed image
        setup_ok = True
    except Exception as exc:
        setup_ok, result = False, dict(error=str(exc)[:300])
    # Every rank runs the SAME sequence of collectives whatever happened above (a rank that failed while building its replica
    # used to skip to the final barrier while the healthy ones waited in all_reduce): agree first, skip the leg on all ranks together
    ok = torch.tensor([1.0 if setup_ok else 0.0], dtype=torch.float64, device=_device_of(use_cuda))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 0.0:
        return result if result is not None else dict(error="another rank could not build its replica of the graph")
    timed = None
    try:
        t0 = time.perf_counter()
        split.propagate(adj, feats, gather=False)
        L.check(L.lib().pgh_sync())
        dt = time.perf_counter() - t0
        info = split.last_batches[0]
        timed = [dt, float(sum(c["spmv"] for c in info))]
    except Exception as exc:
        result = dict(error=str(exc)[:300])
    t = torch.tensor(timed if timed is not None else [-1.0, 0.0], dtype=torch.float64, device=_device_of(use_cuda))
    worst, least = t.clone(), t.clone()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    dist.all_reduce(least, op=dist.ReduceOp.MIN)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if float(least[0].item()) < 0:
        return result if result is not None else dict(error="another rank failed inside the timed batch")
    try:
        steps = max(c["spmv"] for c in info)
        result = dict(edge_vector_products_per_s_G=round(nnz * float(t[1].item()) / float(worst[0].item()) / 1e9, 1),
                      seed_sets=width, seed_sets_per_rank=per_rank, ranks=world, slowest_rank_ms=round(float(worst[0].item()) * 1e3, 2),
                      rank0_device_step_us=round(info[0]["loop_ms"] / steps * 1e3, 1), rank0_columns=list(split.columns),
                      workload=f"64 x {world} seed sets on RMAT scale-{scale} ef-{ef}, one whole-graph replica per rank, no exchange")
        del feats, adj
    except Exception as exc:
        result = dict(error=str(exc)[:300])
    return result


# ----------------------------------------------------------------------------------------------------------------
# The N-rank run is SUPERVISED: the process the launcher starts for a rank never touches the GPU; it runs the rank's work as a child
# process, and when any rank's child fails -- an exception, a crash inside RCCL, a watchdog exit (code 3), a stalled collective past the
# rung's deadline -- every supervisor ends its child (SIGKILL of the child's process group: the kernel driver then tears the queues of a
# spinning collective down) and all of them start a FRESH child tree one rung further down a ladder of more conservative settings.
# Nothing is ever re-executed from a process that has initialised the GPU.  The rung that produced the line is in config.fallback_rung.
# Supervisors of one node agree through files in a scratch directory (the ranks of this bench share one node by contract).
# ----------------------------------------------------------------------------------------------------------------
LADDER = (
    ("default", {}),
    ("one communicator, one stream", dict(PGH_DIST_SINGLE_COMM="1", PGH_DIST_SINGLE_STREAM="1")),
    ("one communicator, one stream, dense all-gather, Python-driven loop",
     dict(PGH_DIST_SINGLE_COMM="1", PGH_DIST_SINGLE_STREAM="1", PGH_DIST_EXCHANGE="allgather", PGH_DIST_NEED_LISTS="0", PGH_DIST_NATIVE="0")),
)


def _sync_dir():
    """A directory every supervisor of THIS launch derives alike: the launcher (their common parent) and its start time."""
    import tempfile
    if os.environ.get("PGH_BENCH_SYNC_DIR"):
        path = os.environ["PGH_BENCH_SYNC_DIR"]
    else:
        ppid = os.getppid()
        try:
            with open(f"/proc/{ppid}/stat") as f:
                started = f.read().rsplit(")", 1)[1].split()[19]          # field 22: start time of the launcher in clock ticks
        except Exception:
            started = "0"
        path = os.path.join(tempfile.gettempdir(), f"pgh_bench_{os.environ.get('MASTER_PORT', '0')}_{ppid}_{started}")
    os.makedirs(path, exist_ok=True)
    return path


def _write_atomic(path, text):
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read_or_none(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def supervise_rank(args):
    """One rank's supervisor (see the block comment above).  Returns the exit code of this process."""
    import signal
    import socket
    rank, world = int(os.environ["RANK"]), int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    entry = os.path.abspath(getattr(sys.modules["__main__"], "__file__", __file__))
    sync = _sync_dir()
    rung_s = float(os.environ.get("PGH_BENCH_RUNG_S", "540"))                # deadline of one rung's child tree
    first = int(os.environ.get("PGH_BENCH_FIRST_RUNG", "0"))
    partials = []

    def note(text):
        sys.stderr.write(f"[bench supervisor] rank {rank}: {text}\n")
        sys.stderr.flush()

    for rung in range(first, len(LADDER)):
        label, rung_env = LADDER[rung]
        port_file = os.path.join(sync, f"rung{rung}.port")
        if rank == 0:
            with socket.socket() as sock:                                     # the child tree's own rendezvous (rank 0's child hosts the store)
                sock.bind(("127.0.0.1", 0))
                _write_atomic(port_file, str(sock.getsockname()[1]))
        deadline = time.time() + 120
        while _read_or_none(port_file) is None and time.time() < deadline:
            time.sleep(0.05)
        port = _read_or_none(port_file)
        if port is None:
            note(f"rung {rung}: rank 0's supervisor never published a port")
            return 4
        env = dict(os.environ, PGH_BENCH_WORKER="1", PGH_BENCH_RUNG=str(rung), PGH_BENCH_RUNG_LABEL=label, MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port.strip(), PGH_BENCH_PARTIAL=os.path.join(sync, f"rung{rung}.partial.json"),
                   PGH_BENCH_WATCHDOG_S=str(max(rung_s - 20, 30)))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)                         # (the launcher's store belongs to the supervisors' processes)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(rung_env)
        proc = subprocess.Popen([sys.executable, entry] + sys.argv[1:], stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True,
                                start_new_session=True)
        lines = []
        import threading

        def pump():
            for out in proc.stdout:
                if out.lstrip().startswith("{"):
                    lines.append(out.strip())
                else:
                    sys.stderr.write(out)
        reader = threading.Thread(target=pump, daemon=True)
        reader.start()
        t_end = time.time() + rung_s
        killed = None
        while proc.poll() is None:
            time.sleep(0.2)
            # a peer's child has failed: this rung is lost on every rank -- do not wait for a collective that will never complete
            failed_peer = next((r for r in range(world) if r != rank and (_read_or_none(os.path.join(sync, f"rung{rung}.rank{r}.rc")) or "0").strip() != "0"), None)
            if failed_peer is not None:
                killed = f"rank {failed_peer}'s child failed"
            elif time.time() > t_end:
                killed = f"no result within {rung_s:.0f} s"
            if killed is not None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
        reader.join(timeout=10)
        rc = proc.returncode if killed is None else (3 if "within" in killed else 5)
        if rc == 0 and rank == 0 and not lines:
            rc = 6                                                              # rank 0's child ended without a line
        _write_atomic(os.path.join(sync, f"rung{rung}.rank{rank}.rc"), str(rc))
        if rc != 0:
            note(f"rung {rung} ({label}): child ended with code {rc}" + (f" ({killed})" if killed else ""))
        # every supervisor learns every child's outcome (a peer still waits for its child at most until the rung's deadline)
        t_wait = time.time() + rung_s + 60
        codes = None
        while time.time() < t_wait:
            got = [_read_or_none(os.path.join(sync, f"rung{rung}.rank{r}.rc")) for r in range(world)]
            if all(v is not None for v in got):
                codes = [int(v.strip() or "1") for v in got]
                break
            time.sleep(0.1)
        if codes is None:
            note(f"rung {rung}: a peer's supervisor never reported")
            return 4
        if all(c == 0 for c in codes):
            if rank == 0:
                print(lines[-1], flush=True)
            return 0
        partial = _read_or_none(os.path.join(sync, f"rung{rung}.partial.json"))
        if partial:
            partials.append(partial.strip())
        if rank == 0:
            note(f"rung {rung} ({label}) failed on ranks {[r for r, c in enumerate(codes) if c != 0]} (codes {codes})"
                 + ("; starting a fresh child tree one rung down" if rung + 1 < len(LADDER) else "; no rung left"))
    # no rung produced a complete line: the headline of the first rung whose TIMED REGION completed on every rank (its later legs --
    # parity, the same graph on one GPU, the replica split -- did not) is still a measurement; the line says what is missing
    if rank == 0 and partials:
        out = json.loads(partials[0])
        out["config"]["incomplete"] = "the run failed after its timed region: parity / same-graph / replica legs are missing"
        print(json.dumps(out), flush=True)
        return 0
    return 0 if partials else 3


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree (torch.distributed.run, one
    rank per GPU) and relay rank 0's JSON line.  This parent has not touched the GPU (no HIP call, no torch.cuda query) and
    never replaces itself with another program.  The ranks the launcher starts are supervisors (supervise_rank): the fallback
    ladder lives there, so the driver's own `torch.distributed.run ... bench.py` launch has it too."""
    import socket
    entry = os.path.abspath(getattr(sys.modules["__main__"], "__file__", __file__))     # tests enter through a wrapper
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    rc, line = 1, None
    for attempt in range(2):                    # the port is free when it is picked, not necessarily when the launcher binds it
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), entry] + sys.argv[1:]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True)
        line = None
        for out in proc.stdout:
            if out.lstrip().startswith("{"):
                line = out.strip()
            else:
                sys.stderr.write(out)
        rc = proc.wait()
        if line is not None or rc == 3:         # a result, or every rung of the ranks' own ladder failed
            break
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=None, help="override the RMAT scale (default: 23 + log2(gpus))")
    ap.add_argument("--ef", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side measurements (other stopping rules / filters)")
    ap.add_argument("--no-weighted", action="store_true", help="skip the real-valued-weights side measurement")
    ap.add_argument("--no-symmetric", action="store_true", help="skip the symmetrised-graph side measurement")
    ap.add_argument("--no-same-graph", action="store_true", help="N > 1: skip the single-GPU run of the same graph on rank 0")
    ap.add_argument("--force-partitioned", action="store_true", help="run the row-partitioned path even with one rank")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:       # no launcher around us: become the launcher's parent
        sys.exit(spawn_ranks(args))
    if (args.gpus > 1 or world > 1) and os.environ.get("PGH_BENCH_WORKER") != "1" and os.environ.get("PGH_BENCH_LADDER", "1") != "0":
        sys.exit(supervise_rank(args))                   # a launcher's rank: supervise the work as a child (fallback ladder)
    if args.force_partitioned and args.gpus == 1 and "RANK" not in os.environ:      # plain `python bench.py --force-partitioned`
        for key, val in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29655")):
            os.environ.setdefault(key, val)
    if args.gpus > 1 or world > 1 or args.force_partitioned:
        result = multi_gpu(args)
    else:
        args.scale = 23 if args.scale is None else args.scale
        args.ef = 16 if args.ef is None else args.ef
        result = single_gpu(args)
    if args.gpus > 1 or world > 1 or args.force_partitioned:
        import torch.distributed as dist
        if dist.is_initialized():
            from pygrank_amd.distributed import release_native_comms
            dist.barrier()
            release_native_comms()
            dist.destroy_process_group()
    if result is not None:
        _emit(json.dumps(result))


if __name__ == "__main__":
    main()
