"""Step-kernel times of ONE rank's slice of a row-partitioned graph, alone on the GPU (no exchange): what a rank of the
N-GPU bench computes per iteration.  Usage: python tools/probe_partition.py [--worlds 2 4 8]   (PGH_PB=0 for the A/B)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pygrank_amd as pg  # noqa: E402
from pygrank_amd import _lib as L  # noqa: E402
from pygrank_amd.device import DeviceVector  # noqa: E402
from pygrank_amd.distributed import _HOT_PAD, rmat_partitioned  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--seeds", action="store_true", help="a seed-set personalization (100 hot rows) and the isolated rows watched, as the "
                                                       "partitioned PageRank loop does, instead of a dense random p")
ap.add_argument("--lib", default=None, help="a diagnostic build of the engine (tools/build_variants.sh) instead of the in-tree one")
args = ap.parse_args()
if args.lib:
    L._lib = L.load_library(os.path.abspath(args.lib))
pg.load_backend("hip")
lib = L.lib()
for world in args.worlds:
    scale, ef = (27, 8) if world == 8 else (23 + int(np.log2(world)), 16)      # bench.py's weak-scaling workloads
    part = rmat_partitioned(scale, ef, 0, world, seed=0)
    g = part.graph
    nb, blk = C.c_int32(), C.c_int64()
    live = np.zeros(8, dtype=np.int32)
    L.check(lib.pgh_graph_gather_layout(g._h, C.byref(nb), C.byref(blk), live.ctypes.data_as(C.c_void_p)))
    top = min(blk.value, (int(live.max()) + 63) // 64 * 64)
    bpr = nb.value // world
    bases = np.zeros(8, dtype=np.int64)
    need = np.zeros(8, dtype=np.int64)
    L.check(lib.pgh_dist_need_counts(g._h, need.ctypes.data_as(C.c_void_p)))
    hs = C.c_int32()
    L.check(lib.pgh_graph_hot_prefix(g._h, C.byref(hs)))
    rng = np.random.default_rng(0)
    if need.sum() > 0:
        # need lists: [block][hot] | [block][the cold slots this slice references] -- what a rank of the N-GPU run gathers from
        hot = hs.value
        cold_bases = np.zeros(8, dtype=np.int64)
        for b in range(nb.value):
            bases[b] = b * hot
        cold_bases[:nb.value] = nb.value * hot + np.concatenate(([0], np.cumsum(need[:nb.value])))[:-1]
        L.check(lib.pgh_graph_set_gather_bases_split(g._h, bases.ctypes.data_as(C.c_void_p), cold_bases.ctypes.data_as(C.c_void_p)))
        xg_len = nb.value * hot + int(need.sum())
        received = 4 * (hot * bpr * (world - 1) + int(need[bpr:nb.value].sum()))
        print(f"world={world}: need lists -- this rank references {int(need.sum())} of {nb.value * (top - hot)} live cold slots "
              f"({need.sum() / max(nb.value * (top - hot), 1):.3f}); received per iteration {received / 1e6:.1f} MB "
              f"(all-gather of the live prefixes: {4 * top * bpr * (world - 1) / 1e6:.1f} MB)", flush=True)
    else:
        for b in range(nb.value):
            r, j = divmod(b, bpr)
            bases[b] = (j * world + r) * top
        L.check(lib.pgh_graph_set_gather_bases(g._h, bases.ctypes.data_as(C.c_void_p)))
        xg_len = nb.value * top
    xg = DeviceVector.from_host(rng.random(xg_len + _HOT_PAD).astype(np.float32))
    if args.seeds:
        p_host = np.zeros(part.n_local, dtype=np.float32)
        p_host[:100] = 1.0
    else:
        p_host = rng.random(part.n_local).astype(np.float32)
    p = DeviceVector.from_host(p_host)
    y = DeviceVector.from_host(np.zeros(part.n_local, dtype=np.float32))
    if args.seeds:
        L.check(lib.pgh_dist_watch_isolated(g._h, p._h, p._h))
    out = DeviceVector.from_host(np.zeros(part.n_local, dtype=np.float32))
    for it in range(args.steps + 2):
        if it == 2:
            L.check(lib.pgh_profile_reset())
            L.check(lib.pgh_profile_enable(1))
        L.check(lib.pgh_ppr_step_dist(g._h, xg._h, 1.0, p._h, 0.85, y._h, out._h, None))
    pack_note = ""
    if need.sum() > 0:
        # the pack launch of the need-list exchange: every destination's stretch in one kernel.  What the peers ask of this rank is, by
        # symmetry, what it asks of them: its own list of its own blocks stands in for each of the `world` destinations
        mine = []
        for j in range(bpr):
            arr = np.zeros(int(need[j]), dtype=np.uint32)
            if len(arr):
                L.check(lib.pgh_dist_need_list(g._h, j, arr.ctypes.data_as(C.c_void_p)))
            mine.append(arr)
        slots = np.concatenate(mine * world)
        seg_counts = [len(mine[j]) for _ in range(world) for j in range(bpr)]
        seg_off = np.concatenate(([0], np.cumsum(seg_counts))).astype(np.int64)
        seg_block = np.array([j for _ in range(world) for j in range(bpr)], dtype=np.int32)
        L.check(lib.pgh_dist_set_send_lists(g._h, slots.ctypes.data_as(C.c_void_p), seg_block.ctypes.data_as(C.c_void_p),
                                            seg_off.ctypes.data_as(C.c_void_p), len(seg_counts)))
        send = DeviceVector.from_host(np.zeros(len(slots), dtype=np.float32))
        for _ in range(args.steps):
            L.check(lib.pgh_dist_pack(g._h, out._h, send._h))
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(L.K_PACK, C.byref(cnt), C.byref(ms)))
        pack_note = f" | pack launch (all {world} destinations, {len(slots)} slots = {len(slots) * 4 / 1e6:.1f} MB): {ms.value / max(cnt.value, 1) * 1e3:.1f} us"
        del send
    L.check(lib.pgh_profile_enable(0))
    if args.seeds:
        L.check(lib.pgh_dist_release_isolated(g._h))
    prof = {}
    for kid, name in ((L.K_SPMV, "spmv"), (L.K_PB_GATHER, "pbA"), (L.K_PB_ACCUM, "pbB"), (L.K_FIXUP, "fixup"), (L.K_COMBINE, "combine")):
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
        prof[name] = ms.value / cnt.value * 1e3 if cnt.value else 0.0
    step = sum(prof.values())
    print(f"world={world} scale={scale} ef={ef} slice nnz={g.nnz} n_local={part.n_local} step={step:7.1f} us "
          f"({g.nnz / step / 1e3:6.1f} GTEPS per rank) " + " ".join(f"{k}={v:.1f}" for k, v in prof.items()) + pack_note + f" | {g.format()}", flush=True)
    del xg, p, y, out, part, g
