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
import sys
import time

import numpy as np

from pygrank_amd import _lib as L
from pygrank_amd.device import DeviceGraph, DeviceVector

_NORMALIZATIONS = {"col": 0, "symmetric": 1, "none": 2}


class PartitionedGraph:
    """This rank's slice of a globally relabelled RMAT graph."""

    def __init__(self, graph, rank, world):
        self.graph, self.rank, self.world = graph, rank, world
        self.n = graph.shape[0]                 # global id count (length of the gather vector; a caller's matrix is padded to it)
        self.n_nodes = self.n                   # nodes of the caller's graph: what the Mabs rule averages over (partition_scipy sets it)
        self.n_local = graph.shape[1]           # rows of M^T held here
        self.row_begin = rank * self.n_local
        self._perm = None

    def local_slice(self, vec_original):
        """This rank's slice (new id order) of a vector given in ORIGINAL ids; padding ids get 0."""
        perm = self.perm[self.row_begin:self.row_begin + self.n_local]
        out = np.zeros(self.n_local)
        ok = perm >= 0
        out[ok] = np.asarray(vec_original, dtype=np.float64)[perm[ok]]
        return out

    def scatter_slice(self, slice_new, into_original):
        """Writes this rank's slice (new id order) into a vector indexed by ORIGINAL ids."""
        perm = self.perm[self.row_begin:self.row_begin + self.n_local]
        ok = perm >= 0
        into_original[perm[ok]] = np.asarray(slice_new)[ok]

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


def _auto_blocks(n):
    """Column blocks the engine's layout would choose for n sources (bsf_auto_blocks, csrc/pgh_bsf.hip)."""
    blocks = 1
    while blocks < 4 and n * 4 > blocks * (8 << 20):
        blocks <<= 1
    return 8 if n * 4 > (16 << 20) else blocks


def partition_scipy(M, rank, world):
    """Row-partitioned upload of a caller's matrix: M = the preprocessor's normalised scipy CSR (n x n, what
    scipy_sparse_to_backend receives, preprocessing.py:144).  Every rank calls this with the SAME matrix and derives the same
    relabelling: sources sorted by descending entry count (stable), dealt round-robin to B = max(world, auto) hot-first column
    blocks, the id space padded to a multiple of 64 B; rank r then uploads only the columns of its slice of the new ids
    (its rows of M^T).  Padding ids have no entries: with a zero personalization there they stay zero."""
    import scipy.sparse as sp
    L.ensure_init()
    assert world in (1, 2, 4, 8) and 0 <= rank < world, "1, 2, 4 or 8 ranks"
    M = sp.csr_array(M)
    n = M.shape[0]
    assert M.shape[0] == M.shape[1], "square matrices only"
    blocks = max(world, _auto_blocks(n))
    unit = 64 * blocks
    n_pad = (n + unit - 1) // unit * unit
    blk = n_pad // blocks
    order = np.argsort(-np.diff(M.indptr), kind="stable")               # rank by descending source count
    r = np.arange(n, dtype=np.int64)
    new_of_rank = (r % blocks) * blk + r // blocks                       # partitions deal the ranks one by one (nnz balance between the ranks)
    perm = np.full(n_pad, -1, dtype=np.int32)
    perm[new_of_rank] = order
    iperm = np.empty(n, dtype=np.int64)
    iperm[order] = new_of_rank
    n_local = n_pad // world
    lo = rank * n_local
    coo = M.tocoo()
    cols = iperm[coo.col]
    keep = (cols >= lo) & (cols < lo + n_local)
    local = sp.csr_array((coo.data[keep], (iperm[coo.row][keep], cols[keep] - lo)), shape=(n_pad, n_local))
    local.sort_indices()
    indptr = np.ascontiguousarray(local.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(local.indices, dtype=np.int32)
    data = np.ascontiguousarray(local.data, dtype=np.float64)
    h = L.c_graph()
    L.check(L.lib().pgh_graph_from_csr_part(n_pad, n_local, local.nnz, indptr.ctypes.data_as(C.c_void_p),
                                            indices.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p), lo, blocks,
                                            perm.ctypes.data_as(C.c_void_p), C.byref(h)))
    out = PartitionedGraph(DeviceGraph(h, (n_pad, n_local), local.nnz), rank, world)
    out.n_nodes = n                     # the padding ids are not nodes: Mabs divides by the caller's count (measures: sum / len)
    return out


_HOT_PAD = 32768          # the engine's LDS hot cache reads up to this many leading slots of a block's slice


class _Buffers:
    """torch tensors (device memory for RCCL, host memory for gloo) viewed as engine vectors.

    Gather-vector layout.  The engine cuts the source id space into `nb` hot-first column blocks of `blk` slots; rank r
    owns blocks r*bpr .. r*bpr + bpr - 1 (bpr = nb / world).  Only the first L slots of a block can be referenced by
    anybody (pgh_graph_gather_layout; L = max over ranks), so the DENSE exchange is bpr all-gathers of L floats per rank and
    the gather vector is stored as [j][rank][L]: block r*bpr + j starts at (j * world + r) * L.

    Need lists ("grouped ncclSend/ncclRecv for exact uneven slices"; pgh_dist_need_counts).  A slice whose cold entries
    all live in the propagation-blocking image numbers its cold sources compactly: it wants, per block, only the cold slots it
    references (a rank of an 8-way partition: ~43 % of the live ones).  When EVERY rank's slice does, the exchange of a step is: the hot
    prefixes all-gathered as before ([j][rank][hot], a few hundred KB), then ONE pack launch (pgh_dist_pack: every destination's
    stretch) and ONE all_to_all of the packed stretches into the compact cold region [block][its referenced slots].  When only some
    ranks' slices are compact, the dense all-gather stays and those ranks compact their copy locally (pgh_dist_compact_from_dense)."""

    def __init__(self, pgraph, device, dist):
        import torch
        self.torch = torch
        lib = L.lib()
        g = pgraph.graph
        n_local, world = pgraph.n_local, pgraph.world
        rank = pgraph.rank
        nb, blk = C.c_int32(), C.c_int64()
        live = np.zeros(8, dtype=np.int32)
        L.check(lib.pgh_graph_gather_layout(g._h, C.byref(nb), C.byref(blk), live.ctypes.data_as(C.c_void_p)))
        self.nb, self.blk = nb.value, blk.value
        assert self.nb % world == 0 and self.blk * self.nb == pgraph.n
        self.bpr = self.nb // world
        top = torch.tensor([int(live.max())], dtype=torch.int64, device=device)
        dist.all_reduce(top, op=dist.ReduceOp.MAX)
        self.live = min(self.blk, (int(top.item()) + 63) // 64 * 64)
        self.world = world
        # Hot / cold split of the exchange: when the slice's stream is hot-only (every cold entry lives in the
        # propagation-blocking image) the block partial sums read just the first `hot` slots of every block, so those
        # are exchanged first (a few hundred KB) and the bulk of the gather vector travels while they run.
        hs = C.c_int32()
        L.check(lib.pgh_graph_hot_prefix(g._h, C.byref(hs)))
        self._engine_hot = hs.value
        self.engine_streams = None          # (exchange stream, compute stream) while a three-queue run is on: where pack / compaction launch
        counts = np.zeros(8, dtype=np.int64)
        L.check(lib.pgh_dist_need_counts(g._h, counts.ctypes.data_as(C.c_void_p)))
        self.compact = bool(counts[:self.nb].sum() > 0)                          # this slice numbers its cold sources compactly
        agree = torch.tensor([hs.value, 1 if self.compact else 0], dtype=torch.int64, device=device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)                             # every rank must split the same way
        hot = int(agree[0].item())
        self.hot = (min(hot, self.live) + 63) // 64 * 64 if 0 < hot < self.live else 0
        self.lists = bool(int(agree[1].item())) and self.hot > 0 and os.environ.get("PGH_DIST_EXCHANGE", "lists") != "allgather"
        # (a compact slice beside a peer whose slice is NOT hot-only -- no cold image below the entry gate, a heavy row kept in the stream,
        # PGH_DIST_NEED_LISTS=0 on that rank -- sees hot == 0 here: one dense region, the split bases are the dense block starts, the compact
        # copy lies behind them and _compact_own_copy reads the cold slots at bases[b] + the ENGINE's hot prefix)
        self.need_counts = counts[:self.nb].copy()
        self.need_prefix = np.concatenate(([0], np.cumsum(self.need_counts)))
        need_total = int(self.need_prefix[-1])
        self.bases = np.zeros(8, dtype=np.int64)
        self.cold_bases = None
        if self.lists:
            # [j][rank][hot] | [block][referenced cold slots]
            dense_len = self.nb * self.hot
            for b in range(self.nb):
                r, j = divmod(b, self.bpr)
                self.bases[b] = (j * world + r) * self.hot
            self.cold_bases = np.zeros(8, dtype=np.int64)
            self.cold_bases[:self.nb] = dense_len + self.need_prefix[:-1]
            self.cold_at = dense_len
        else:
            dense_len = self.nb * self.live
            for b in range(self.nb):
                r, j = divmod(b, self.bpr)
                self.bases[b] = (j * world + r) * self.live
            if self.compact:
                # the dense exchange lands as before; this rank's image reads its cold sources from a compact copy behind it
                self.cold_bases = np.zeros(8, dtype=np.int64)
                self.cold_bases[:self.nb] = dense_len + self.need_prefix[:-1]
                self.cold_at = dense_len
        n_xg = dense_len + (need_total if self.compact else 0) + _HOT_PAD
        self.apply_bases(g)
        self.xg_full = torch.zeros(n_xg, dtype=torch.float32, device=device)
        self.xg_local = torch.zeros(n_local, dtype=torch.float32, device=device)
        self.y = [torch.zeros(n_local, dtype=torch.float32, device=device) for _ in range(2)]
        self.scalar = torch.zeros(1, dtype=torch.float64, device=device)
        self.state = torch.zeros(8, dtype=torch.float64, device=device)          # pgh_dist_* state (include/pgh.h)
        self.state_host = torch.zeros(8, dtype=torch.float64)
        if device.type == "cuda":
            self.state_host = self.state_host.pin_memory()
        self.v_xg_full = DeviceVector.wrap(self.xg_full.data_ptr(), n_xg, keepalive=self.xg_full)
        self.v_xg_local = DeviceVector.wrap(self.xg_local.data_ptr(), n_local, keepalive=self.xg_local)
        self.v_y = [DeviceVector.wrap(t.data_ptr(), n_local, keepalive=t) for t in self.y]
        self.exchange_bytes = 4 * self.live * self.bpr * (world - 1)            # received per rank and iteration
        # the big exchange and the scalar reductions use communicators of their own so that neither queues behind the other
        # (one exchange communicator per process, not per graph: PGH_DIST_SINGLE_COMM=1 keeps everything on the default one)
        self.pg_exchange = _exchange_group(dist, world)
        if self.lists:
            self._setup_lists(pgraph, device, dist, lib, g, rank, world)

    def _setup_lists(self, pgraph, device, dist, lib, g, rank, world):
        """Once per graph: every rank tells the owner of each block which of its cold slots it references."""
        torch = self.torch
        bpr, nb = self.bpr, self.nb
        mine = torch.from_numpy(self.need_counts.astype(np.int64)).to(device)
        everyone = torch.zeros(world * nb, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(everyone, mine)
        counts_all = everyone.cpu().numpy().reshape(world, nb)                  # [asking rank][block]
        lists = []
        for b in range(nb):
            arr = np.zeros(int(self.need_counts[b]), dtype=np.uint32)
            if len(arr):
                L.check(lib.pgh_dist_need_list(g._h, b, arr.ctypes.data_as(C.c_void_p)))
            lists.append(arr)
        ask = np.concatenate(lists) if lists else np.zeros(0, dtype=np.uint32)   # block-major = owner-major
        ask_splits = [int(self.need_counts[s * bpr:(s + 1) * bpr].sum()) for s in range(world)]
        asked_splits = [int(counts_all[r, rank * bpr:(rank + 1) * bpr].sum()) for r in range(world)]
        asked = torch.zeros(sum(asked_splits), dtype=torch.int32)
        self._all_to_all(dist, asked, torch.from_numpy(ask.view(np.int32).copy()), asked_splits, ask_splits, device)
        seg_counts = [int(counts_all[r, rank * bpr + j]) for r in range(world) for j in range(bpr)]       # destination-major
        seg_off = np.concatenate(([0], np.cumsum(seg_counts))).astype(np.int64)
        seg_block = np.array([j for _ in range(world) for j in range(bpr)], dtype=np.int32)
        slots = np.ascontiguousarray(asked.numpy().view(np.uint32))
        L.check(lib.pgh_dist_set_send_lists(g._h, slots.ctypes.data_as(C.c_void_p), seg_block.ctypes.data_as(C.c_void_p),
                                            seg_off.ctypes.data_as(C.c_void_p), len(seg_counts)))
        self.send_splits, self.recv_splits = asked_splits, ask_splits
        self.send_buf = torch.zeros(max(sum(asked_splits), 1), dtype=torch.float32, device=device)
        self.v_send = DeviceVector.wrap(self.send_buf.data_ptr(), self.send_buf.numel(), keepalive=self.send_buf)
        self.exchange_bytes = 4 * (self.hot * bpr * (world - 1) + sum(ask_splits) - ask_splits[rank])
        # what the host-side count says this rank receives per step (asserted against the engine's figure by the tests)
        self.exchange_slots = dict(hot=self.hot * bpr * (world - 1), cold=sum(ask_splits) - ask_splits[rank], cold_dense=(self.live - self.hot) * bpr * (world - 1))

    def _all_to_all(self, dist, recv, send, recv_splits, send_splits, device):
        """all_to_all_single with uneven splits on `self.pg_exchange`; gloo moves host tensors only: device tensors are staged."""
        torch = self.torch
        staged = dist.get_backend() != "nccl" and (recv.is_cuda or send.is_cuda)
        on_dev = dist.get_backend() == "nccl" and not recv.is_cuda
        if staged:
            r_host, s_host = torch.zeros(recv.numel(), dtype=recv.dtype), send.cpu()
            dist.all_to_all_single(r_host, s_host, recv_splits, send_splits, group=self.pg_exchange)
            recv.copy_(r_host)
        elif on_dev:                                                              # (setup lists are host arrays; RCCL wants device memory)
            r_dev, s_dev = torch.zeros(recv.numel(), dtype=recv.dtype, device=device), send.to(device)
            dist.all_to_all_single(r_dev, s_dev, recv_splits, send_splits, group=self.pg_exchange)
            recv.copy_(r_dev.cpu())
        else:
            dist.all_to_all_single(recv, send, recv_splits, send_splits, group=self.pg_exchange)

    def apply_bases(self, g):
        """Tells the graph where every block's slice starts inside THIS object's gather vector.  The bases are state of the graph,
        not of the buffers: the engine's own loop (pgh_dist_ppr_run) lays the same graph out its own way -- two regions -- on every
        run, so a Python-driven run re-applies its layout every time it starts (cached buffers used to read a torch
        gather vector through the native split bases after an engine run on the same graph)."""
        if self.cold_bases is not None:
            L.check(L.lib().pgh_graph_set_gather_bases_split(g._h, self.bases.ctypes.data_as(C.c_void_p), self.cold_bases.ctypes.data_as(C.c_void_p)))
        else:
            L.check(L.lib().pgh_graph_set_gather_bases(g._h, self.bases.ctypes.data_as(C.c_void_p)))

    def _pieces(self, j, lo, hi):
        """views of slots [lo, hi) of the blocks j, bpr + j, ... (one per rank) inside xg_full, and of this rank's slice"""
        L_ = self.hot if self.lists else self.live
        outs = [self.xg_full[(j * self.world + r) * L_ + lo:(j * self.world + r) * L_ + hi] for r in range(self.world)]
        return outs, self.xg_local[j * self.blk + lo:j * self.blk + hi]

    def _on_exchange_stream(self, launch):
        """Engine launches that belong to the exchange (pack, local compaction) run on the exchange queue of a three-queue run."""
        if self.engine_streams is None:
            return launch()
        lib = L.lib()
        L.check(lib.pgh_set_stream(C.c_void_p(self.engine_streams[0].cuda_stream)))
        try:
            return launch()
        finally:
            L.check(lib.pgh_set_stream(C.c_void_p(self.engine_streams[1].cuda_stream)))

    def _compact_own_copy(self, g):
        """dense exchange, compact image: the referenced cold slots of every block, out of the dense copy"""
        lib = L.lib()
        hot = self._engine_hot

        def launch():
            for b in range(self.nb):
                if self.need_counts[b]:
                    L.check(lib.pgh_dist_compact_from_dense(g._h, b, self.v_xg_full._h, int(self.bases[b]) + hot, self.v_xg_full._h,
                                                            self.cold_at + int(self.need_prefix[b])))
        self._on_exchange_stream(launch)

    def all_gather(self, dist, part="all", graph=None):
        """xg_local (this rank's slice of the next gather vector) -> every rank's xg_full, live prefixes only.
        part: "all", or "hot" / "cold" = slots [0, hot) / [hot, live) of every block.  graph: the slice's DeviceGraph (need lists)."""
        group = self.pg_exchange
        if self.lists:
            if part in ("all", "hot"):
                for j in range(self.bpr):
                    outs, mine = self._pieces(j, 0, self.hot)
                    dist.all_gather(outs, mine, group=group)
            if part in ("all", "cold"):
                self._on_exchange_stream(lambda: L.check(L.lib().pgh_dist_pack(graph._h, self.v_xg_local._h, self.v_send._h)))
                n_send, n_recv = sum(self.send_splits), sum(self.recv_splits)
                self._all_to_all(dist, self.xg_full[self.cold_at:self.cold_at + n_recv], self.send_buf[:n_send], self.recv_splits, self.send_splits,
                                 self.xg_full.device)
            return
        if part == "all" or self.hot == 0:
            if part == "hot":
                return
            per = self.live * self.world                                         # world * L floats per all-gather
            for j in range(self.bpr):
                dist.all_gather_into_tensor(self.xg_full[j * per:(j + 1) * per], self.xg_local[j * self.blk:j * self.blk + self.live],
                                            group=group)
            if self.compact:
                self._compact_own_copy(graph)
            return
        lo, hi = (0, self.hot) if part == "hot" else (self.hot, self.live)
        if hi > lo:                                                              # (else: the hot prefixes are the whole live range)
            for j in range(self.bpr):
                outs, mine = self._pieces(j, lo, hi)
                dist.all_gather(outs, mine, group=group)
        if part == "cold" and self.compact:
            self._compact_own_copy(graph)


_EXCHANGE_GROUPS = {}


def _bounded_wait(event, what):
    """Host wait for a CUDA event with a deadline (PGH_DIST_TIMEOUT_S, default 600 s).  A collective that never completes
    -- a peer that died, communicators that block each other -- would otherwise hang the job for good: the rank says what it
    was waiting for and ends the PROCESS with a non-zero code (the launcher then takes the other ranks down); nothing is
    re-executed.  PGH_DIST_SINGLE_COMM=1 / PGH_DIST_SINGLE_STREAM=1 are the conservative settings to try next."""
    limit = float(os.environ.get("PGH_DIST_TIMEOUT_S", "600"))
    start = time.monotonic()
    spins = 0
    while not event.query():
        spins += 1
        if spins > 2000:
            time.sleep(0.0005)
            if time.monotonic() - start > limit:
                sys.stderr.write(f"[pygrank_amd.distributed] rank {os.environ.get('RANK', '0')}: {what} did not complete within {limit:.0f} s "
                                 "-- a collective is stalled; exiting (retry with PGH_DIST_SINGLE_COMM=1 PGH_DIST_SINGLE_STREAM=1)\n")
                sys.stderr.flush()
                os._exit(3)


def _exchange_group(dist, world):
    """The communicator of the gather-vector exchange, created once per process and world size."""
    if world <= 1 or os.environ.get("PGH_DIST_SINGLE_COMM", "0") == "1":
        return None
    key = (world, dist.get_backend())
    if key not in _EXCHANGE_GROUPS:
        _EXCHANGE_GROUPS[key] = dist.new_group(backend=dist.get_backend())
    return _EXCHANGE_GROUPS[key]


_NATIVE_COMMS = {}


def _native_comm(dist, device):
    """The engine's own RCCL communicator pair (csrc/pgh_dist.hip), created once per process: rank 0 draws the ids, the bytes
    travel through torch.distributed, every rank joins.  None when the run is not RCCL-on-GPU (gloo / CPU tests, the host
    double) or PGH_DIST_NATIVE=0 asks for the Python-driven loop."""
    world, rank = dist.get_world_size(), dist.get_rank()
    # One rank: the engine loop (measured and tested on this pool).  MORE than one rank over RCCL has never run anywhere yet (this
    # pool has single-GPU boxes), so there the Python-driven loop over torch's communicator is the default and the engine loop is
    # opt-in: PGH_DIST_NATIVE=auto (probe first, below) or =1 (no probe); =0 never uses it.
    mode = os.environ.get("PGH_DIST_NATIVE", "auto" if world == 1 else "0")
    if device.type != "cuda" or mode == "0" or not L.runtime_name().startswith("hip:"):
        return None
    key = (world, rank)
    if key in _NATIVE_COMMS:
        return _NATIVE_COMMS[key]
    if dist.get_backend() != "nccl":
        # the engine's loop with the collectives done by the HOST through torch.distributed (pgh_comm_create_external): opt-in
        # (PGH_DIST_NATIVE=external) -- every exchange then goes through host memory; what it is for: running the engine-driven
        # choreography with several ranks where RCCL cannot (ranks sharing one GPU in tests; MPI-style hosts do the same from C)
        if mode != "external":
            return None
        if key not in _NATIVE_COMMS:
            _NATIVE_COMMS[key] = _external_comm(dist, world, rank)
        return _NATIVE_COMMS[key]
    if key not in _NATIVE_COMMS:
        import torch
        lib = L.lib()
        num_ids = 1 if os.environ.get("PGH_DIST_SINGLE_COMM", "0") == "1" else 2
        ids = (C.c_uint8 * (L.COMM_ID_BYTES * num_ids))()
        ok = 1
        if rank == 0:
            for i in range(num_ids):
                if lib.pgh_comm_unique_id(C.cast(C.byref(ids, i * L.COMM_ID_BYTES), C.c_void_p)) != 0:
                    ok = 0
        wire = torch.tensor(list(ids) + [ok], dtype=torch.uint8, device=device)      # the ids and "rank 0 could draw them"
        dist.broadcast(wire, src=0)
        got = wire.cpu().tolist()
        ids, ok = (C.c_uint8 * (L.COMM_ID_BYTES * num_ids))(*got[:-1]), (ok if rank == 0 else int(got[-1]))
        handle = C.c_void_p()
        if ok and lib.pgh_comm_create(C.cast(ids, C.c_void_p), num_ids, world, rank, C.byref(handle)) != 0:
            ok = 0
        # every rank takes the same road: one that could not set its communicators up sends all of them to the Python-driven loop
        agree = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            sys.stderr.write(f"[pygrank_amd.distributed] rank {rank}: the engine's RCCL communicators could not be created "
                             f"({lib.pgh_last_error().decode('utf-8', 'replace') if not ok else 'on another rank'}); using the Python-driven loop\n")
            if handle.value:
                lib.pgh_comm_destroy(handle)
            handle = None
        _NATIVE_COMMS[key] = handle
        # More than one rank over RCCL: the engine's loop (two communicators of its own on three streams) proves itself on a small
        # graph first -- against the Python-driven loop over torch's communicator, with short bounded waits -- and every rank takes
        # the Python-driven loop when it does not (this pool has no multi-GPU box, so the first N > 1 run of the engine
        # loop is a user's, or the driver's scaling bench).  PGH_DIST_NATIVE=1 skips the probe, =0 never uses the engine loop.
        if handle is not None and world > 1 and mode == "auto":
            PREFLIGHT[key] = _preflight(dist, device, rank, world, lib)
            if PREFLIGHT[key].startswith("stalled"):
                # A collective of the engine loop never completed: its RCCL kernels are still spinning on this GPU (RunScope leaves
                # them in flight), so nothing else can be trusted to run here -- not the staged loop over torch's communicator, not a
                # hipFree (it would wait for every queue of the device).  The process ends with a non-zero code and says how to
                # relaunch; the launcher takes the other ranks down.  Nothing is re-executed from a process that has touched the GPU.
                # (when only SOME ranks stall, the others sit in the staged loop's collectives until somebody ends them -- auto
                # mode therefore belongs under a launcher that tears the job down when a rank exits: torch.distributed.run does, and so do
                # bench.py's rank supervisors, which end every child of a rung as soon as one of them fails and start a fresh tree.)
                sys.stderr.write(f"[pygrank_amd.distributed] rank {rank}: the engine-driven RCCL loop STALLED in its probe ({PREFLIGHT[key]}); "
                                 "this process cannot continue on a GPU with collectives in flight -- relaunch with PGH_DIST_NATIVE=0 "
                                 "(the Python-driven loop), or with PGH_DIST_SINGLE_COMM=1 PGH_DIST_SINGLE_STREAM=1\n")
                sys.stderr.flush()
                os._exit(3)
            if PREFLIGHT[key] != "ok":
                sys.stderr.write(f"[pygrank_amd.distributed] rank {rank}: the engine-driven RCCL loop failed its probe ({PREFLIGHT[key]}); "
                                 "using the Python-driven loop\n")
                # a communicator with a stalled collective cannot be destroyed without waiting for it: it is abandoned
                _NATIVE_COMMS[key] = None
    return _NATIVE_COMMS[key]


PREFLIGHT = {}          # (world, rank) -> "ok" | what went wrong: the engine loop's probe of this process (bench.py reports it)


def _preflight(dist, device, rank, world, lib):
    """One small partitioned PageRank through the engine's loop and through the Python-driven loop.  "ok" when, on every rank, the
    engine loop's iterate equals the staged loop's iterate after the SAME number of steps (<= 1e-6 of the largest rank) and its
    stopping iteration lies between those of the staged loop at a tolerance 5 % looser and 5 % tighter (the two routes evaluate the
    residual differently -- in the finish kernel against a predicted quotient / by the separate kernel -- and residuals near the
    tolerance are not monotone: demanding the identical count would demote a healthy engine loop).  A verdict that starts
    with "stalled" means a bounded wait on a collective expired: the caller must end the process (see _native_comm).  Collective.
    Host waits on the engine's collectives are bounded by PGH_DIST_PREFLIGHT_S (default 30 s) while it runs."""
    import torch
    verdict = "ok"
    stalled = False
    L.check(lib.pgh_dist_set_timeout(float(os.environ.get("PGH_DIST_PREFLIGHT_S", "30"))))
    # the probe's slices get a cold image whatever their size (PGH_PB_FORCE, read at build time), so that what runs is the loop of the
    # large graphs: split regions, the residual inside the finish kernel, ONE 4-scalar all-reduce, the slice degrees' all-reduce
    saved_env = {k: os.environ.get(k) for k in ("PGH_PB", "PGH_PB_FORCE")}
    os.environ.update(PGH_PB="1", PGH_PB_FORCE="1")
    pg = None
    try:
        pg = rmat_partitioned(20, 8, rank, world, seed=0)
        rng = np.random.default_rng(1)
        p_new = np.zeros(pg.n)
        p_new[rng.choice(pg.n, 20, replace=False)] = 1.0                        # in NEW ids: the same on every rank
        p_local = DeviceVector.from_host(p_new[pg.row_begin:pg.row_begin + pg.n_local])
        tol = 1e-6
        kw = dict(alpha=0.85, error_type="l1", max_iters=200)
        # two legs of the engine loop: as this process would run a small exchange (one finish launch), and with the finish kernel in
        # two launches -- what a large exchange (configs[4]) switches on and nothing smaller would exercise over RCCL
        legs = [("", None)]
        if "PGH_DIST_FINISH_SPLIT" not in os.environ:
            legs.append(("two-launch finish: ", "2"))
        runs = []
        for label, split in legs:
            native = DistributedPageRank(tol=tol, **kw)
            native._dist, native._device = dist, device
            if split is not None:
                os.environ["PGH_DIST_FINISH_SPLIT"] = split
            try:
                got = np.asarray(native._rank_native(pg, p_local, _NATIVE_COMMS[(world, rank)], lib), dtype=np.float64)
                runs.append((label, native.iteration, got))
            except Exception as exc:                                               # EngineError of a bounded wait, ...
                stalled = "stalled" in str(exc) or "did not complete within" in str(exc)
                verdict = f"{'stalled: ' if stalled else ''}{label}engine loop: {str(exc)[:200]}"
            finally:
                if split is not None:
                    os.environ.pop("PGH_DIST_FINISH_SPLIT", None)
            if verdict != "ok":
                break                                                              # (a stalled communicator takes no further run)
        if not stalled:
            # the staged loop over torch's communicator: only on queues that are known to be drained
            def staged_run(**over):
                staged = DistributedPageRank(**dict(kw, **over))
                staged._native_formula = False
                return np.asarray(staged.rank(pg, p_local), dtype=np.float64), staged.iteration
            _, upper = staged_run(tol=tol / 1.05)                                  # a tighter tolerance stops no earlier
            _, lower = staged_run(tol=tol * 1.05)
            for label, iterations, got in runs if verdict == "ok" else []:
                want, _ = staged_run(error_type="iters", max_iters=iterations)     # the same number of steps, whatever the rule says
                top = torch.tensor([float(np.max(np.abs(want), initial=0.0))], dtype=torch.float64, device=device)
                dist.all_reduce(top, op=dist.ReduceOp.MAX)
                if not lower <= iterations <= upper:
                    verdict = f"{label}stopping iteration {iterations} outside [{lower}, {upper}] of the staged loop at tol x 1.05 / tol / 1.05"
                elif float(np.max(np.abs(got - want), initial=0.0)) > 1e-6 * float(top.item()):
                    verdict = label + f"ranks differ between the engine loop and the staged loop after {iterations - 1} steps"
                if verdict != "ok":
                    break
            pg.graph.destroy()                                                     # (hipFree: never behind a stalled collective)
    except Exception as exc:
        verdict = f"probe: {str(exc)[:200]}"
    finally:
        lib.pgh_dist_set_timeout(0.0)
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if stalled:
        if pg is not None:
            _ABANDONED.append(pg)                                                  # its device memory is never freed in this process
        return verdict                                                             # no further collective: every rank that stalled exits
    agree = torch.tensor([1 if verdict == "ok" else 0], dtype=torch.int32, device=device)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    if int(agree.item()) == 0 and verdict == "ok":
        verdict = "failed on another rank"
    return verdict


_ABANDONED = []         # graphs of a stalled probe: kept alive so that nothing frees device memory under a collective in flight


_EXTERNAL_KEEPALIVE = []


def _external_comm(dist, world, rank):
    """pgh_comm_create_external with callbacks that move every exchange through host memory and torch.distributed (gloo)."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    dtypes = {0: (torch.float32, 4), 1: (torch.float64, 8), 2: (torch.int32, 4)}

    def gather(user, send, recv, count, dtype, stream):
        try:
            tdt, size = dtypes[int(dtype)]
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            mine = torch.empty(int(count), dtype=tdt)
            if hip.hipMemcpy(mine.data_ptr(), send, int(count) * size, 2) != 0:          # device -> host
                return 1
            everyone = torch.empty(int(count) * world, dtype=tdt)
            dist.all_gather_into_tensor(everyone, mine)
            return 0 if hip.hipMemcpy(recv, everyone.data_ptr(), int(count) * size * world, 1) == 0 else 1
        except Exception as exc:                                                          # never unwind through the C frames
            sys.stderr.write(f"[pygrank_amd.distributed] all-gather callback: {exc}\n")
            return 1

    def reduce(user, buf, count, dtype, op, stream):
        try:
            tdt, size = dtypes[int(dtype)]
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            host = torch.empty(int(count), dtype=tdt)
            if hip.hipMemcpy(host.data_ptr(), buf, int(count) * size, 2) != 0:
                return 1
            dist.all_reduce(host, op=dist.ReduceOp.MAX if int(op) == 1 else dist.ReduceOp.SUM)
            return 0 if hip.hipMemcpy(buf, host.data_ptr(), int(count) * size, 1) == 0 else 1
        except Exception as exc:
            sys.stderr.write(f"[pygrank_amd.distributed] all-reduce callback: {exc}\n")
            return 1

    def exchange(user, send, scounts, soffs, recv, rcounts, roffs, stream):
        """the need lists' point-to-point stretches (4-byte elements) as one all_to_all through host memory"""
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            ssplits, rsplits = [int(scounts[r]) for r in range(world)], [int(rcounts[r]) for r in range(world)]
            out = torch.empty(max(sum(ssplits), 1), dtype=torch.int32)
            at = 0
            for r in range(world):
                if ssplits[r] and hip.hipMemcpy(out.data_ptr() + 4 * at, (send or 0) + 4 * int(soffs[r]), 4 * ssplits[r], 2) != 0:
                    return 1
                at += ssplits[r]
            got = torch.empty(max(sum(rsplits), 1), dtype=torch.int32)
            dist.all_to_all_single(got[:sum(rsplits)], out[:sum(ssplits)], rsplits, ssplits)
            at = 0
            for r in range(world):
                if rsplits[r] and hip.hipMemcpy((recv or 0) + 4 * int(roffs[r]), got.data_ptr() + 4 * at, 4 * rsplits[r], 1) != 0:
                    return 1
                at += rsplits[r]
            return 0
        except Exception as exc:
            sys.stderr.write(f"[pygrank_amd.distributed] all-to-all callback: {exc}\n")
            return 1

    cb_gather, cb_reduce, cb_exchange = L.ALLGATHER_FN(gather), L.ALLREDUCE_FN(reduce), L.ALLTOALLV_FN(exchange)
    _EXTERNAL_KEEPALIVE.extend([cb_gather, cb_reduce, cb_exchange, hip])
    handle = C.c_void_p()
    L.check(L.lib().pgh_comm_create_external(world, rank, C.cast(cb_gather, C.c_void_p), C.cast(cb_reduce, C.c_void_p), None, C.byref(handle)))
    # PGH_DIST_EXTERNAL_A2A=0: a host with all-gather and all-reduce only -- compact slices then copy their slots out of the gathered vector
    if os.environ.get("PGH_DIST_EXTERNAL_A2A", "1") != "0":
        L.check(L.lib().pgh_comm_set_alltoallv(handle, C.cast(cb_exchange, C.c_void_p)))
    return handle


def release_native_comms():
    """Destroys the engine's RCCL communicators of this process (before torch.distributed is torn down)."""
    for handle in _NATIVE_COMMS.values():
        if handle is not None:
            L.lib().pgh_comm_destroy(handle)
    _NATIVE_COMMS.clear()


class DistributedPageRank:
    """PageRank(alpha) with ConvergenceManager(tol, error_type, max_iters, end_modulo) on a PartitionedGraph.

    On GPUs over RCCL the whole run is ONE engine call (pgh_dist_ppr_run: the engine drives RCCL, streams and events itself);
    the Python-driven loop below is the same choreography call by call -- what gloo / CPU runs use, and PGH_DIST_NATIVE=0.

    The loop is device-driven: the iteration's scalars (sum(y), residual, quotient, done flag) live in an 8-double
    device tensor, the RCCL all-reduces act on its elements in place, and every engine call turns into a no-op once the
    stopping rule has fired.  The host only looks at the flag after a step that was followed by a residual check, and
    by then it has already enqueued the next step's partial sums (which do not depend on the scalars), so the device
    never waits for the host."""

    _KINDS = {"mabs": L.ERR_MABS, "l1": L.ERR_L1, "linf": L.ERR_LINF, "iters": L.ERR_ITERS}
    _native_formula = True            # pgh_dist_ppr_run implements this class's step (subclasses with another formula: the staged loop)

    def _native_operands(self, pgraph):
        return None

    def _prepare_run(self, pgraph, p, bufs, lib):
        """Per-run operands of the step formula, and the isolated rows to watch (p: the normalised personalization slice)."""
        L.check(lib.pgh_dist_watch_isolated(pgraph.graph._h, p._h, bufs.v_y[0]._h))

    def _combine(self, lib, g, p, y, xg_local, state):
        """The finish stage of one step on this rank's rows: PageRank._formula (adhoc.py:34-36)."""
        L.check(lib.pgh_dist_combine(g._h, p._h, self.alpha, y._h, xg_local._h, state))

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
        self._dist, self._device = dist, device
        comm = _native_comm(dist, device) if self._native_formula else None
        if comm is not None:
            return self._rank_native(pgraph, p_local, comm, lib)
        if self._buffers is None or self._buffers_for is not pgraph:
            self._buffers = _Buffers(pgraph, device, dist)
            self._buffers_for = pgraph
        bufs = self._buffers
        bufs.apply_bases(g)
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
        state = C.c_void_p(bufs.state.data_ptr())
        sum_view, err_view = bufs.state[2:3], bufs.state[1:2]
        err_op = dist.ReduceOp.MAX if kind == L.ERR_LINF else dist.ReduceOp.SUM
        local_kind = L.ERR_LINF if kind == L.ERR_LINF else L.ERR_L1

        def read_state():
            """(done, steps, converged, scale) once everything enqueued before this call has run."""
            bufs.state_host.copy_(bufs.state, non_blocking=True)
            if device.type == "cuda":
                landed = torch.cuda.Event()
                landed.record(self._stream)
                _bounded_wait(landed, "the loop state of the last step")
            ints = bufs.state_host.view(torch.int32)
            return int(ints[6]), int(ints[7]), int(ints[8]), float(bufs.state_host[0])

        try:
            with stream_ctx:
                return self._rank_on_stream(pgraph, p_local, bufs, dist, lib, g, device, kind, tol, state, sum_view, err_view,
                                            err_op, local_kind, read_state, torch)
        finally:
            L.check(lib.pgh_dist_release_isolated(g._h))
            if device.type == "cuda":
                # every later single-GPU call of this process goes back to the engine's own stream
                L.check(lib.pgh_sync())
                L.check(lib.pgh_set_stream(None))

    def _rank_native(self, pgraph, p_local, comm, lib):
        extra = self._native_operands(pgraph)               # None: PageRank; (deg, lam, every_row): AbsorbingWalks
        cfg = L.DistCfg(alpha=float(self.alpha), tol=0.0 if self.tol is None else max(float(self.tol), self.epsilon), n_global=int(pgraph.n_nodes),
                        err_kind=self._KINDS[self.error_type], max_iters=int(self.max_iters), end_modulo=int(self.end_modulo),
                        use_quotient=1 if self.use_quotient else 0, preserve_norm=1 if self.preserve_norm else 0,
                        every_row=int(extra[2]) if extra else 0, deg_local=extra[0]._h if extra else None,
                        lam_local=extra[1]._h if extra else None)
        res = L.DistResult()
        out = DeviceVector.empty(pgraph.n_local)
        t0 = time.perf_counter()
        L.check(lib.pgh_dist_ppr_run(pgraph.graph._h, comm, p_local._h, out._h, C.byref(cfg), C.byref(res)))
        self.elapsed = time.perf_counter() - t0
        self.iteration, self.spmv, self.converged = int(res.iterations), int(res.spmv_count), bool(res.converged)
        self.last_error, self.loop_ms = float(res.last_error), float(res.loop_ms)
        self.exchange = dict(exchange_bytes_per_iteration_per_gpu=int(res.exchange_bytes), gather_vector_slots=int(res.gather_slots),
                             column_blocks=int(res.column_blocks), split_regions=bool(res.split_regions),
                             in_kernel_residual=bool(res.flags & 2), paused_in_kernel_residual=bool(res.flags & 1),
                             finish_in_two_launches=bool(res.flags & 4),
                             exchange={0: "all-gather", 8: "need lists (point to point)", 16: "all-gather + local compaction"}[res.flags & 24],
                             driver="engine (RCCL)" if self._dist.get_backend() == "nccl" else "engine (host collectives)")
        if res.iterations == 0:
            return p_local
        if not self.converged and self.error_type != "iters" and self.iteration >= self.max_iters:
            raise Exception("Could not converge within " + str(self.max_iters) + " iterations")
        return out

    def _rank_on_stream(self, pgraph, p_local, bufs, dist, lib, g, device, kind, tol, state, sum_view, err_view, err_op,
                        local_kind, read_state, torch):
        """Three queues per rank (CUDA; the CPU/gloo path runs the same calls in order):
          C  the engine's stream: block partial sums -> phase A -> phase B + epilogue of every step;
          X  the exchange: after the epilogue of step k the hot prefixes of the gather slices are all-gathered, then the
             rest; step k + 1's block partial sums wait for the first, its phase A for the second;
          S  the scalars: all-reduce of sum(y), lazy L1 quotient, residual, its all-reduce, stopping rule -- behind the
             epilogue of step k, ahead of the epilogue of step k + 1, beside everything in between.
        The host reads the done flag of step k only after it has enqueued the first two stages of step k + 1."""
        cuda = device.type == "cuda"
        bufs.engine_streams = None               # (the prologue's exchange runs on the compute queue)
        # ---- prologue of GraphFilter.rank (abstract_filters.py:52-56): global L1 norm, x0 = p / norm
        norm = self._all_reduce(bufs, dist, p_local.abssum(), dist.ReduceOp.SUM)
        if norm == 0:
            self.iteration = 0
            return p_local
        p = p_local / norm
        cur = 0
        bufs.y[1 - cur].zero_()                  # the rows a run passes over (isolated ids) must hold zeros in both iterates
        L.check(lib.pgh_vec_copy(bufs.v_y[cur]._h, p._h))
        self._prepare_run(pgraph, p, bufs, lib)
        L.check(lib.pgh_dist_prescale(g._h, bufs.v_y[cur]._h, bufs.v_xg_local._h))
        bufs.all_gather(dist, "all", g)
        L.check(lib.pgh_dist_state_init(state))
        if cuda:
            if getattr(self, "_side", None) is None:
                # PGH_DIST_SINGLE_STREAM=1: exchange and scalars queue on the compute stream (no overlap, no cross-stream order
                # for the collectives to disagree on between ranks) -- the fallback for a first run on new hardware
                single = os.environ.get("PGH_DIST_SINGLE_STREAM", "0") == "1"
                self._side = (self._stream, self._stream) if single else (torch.cuda.Stream(), torch.cuda.Stream())
            main, (xs, ss) = self._stream, self._side
            bufs.engine_streams = (xs, main) if xs is not main else None
            ev_fin, ev_hot, ev_cold, ev_err = (torch.cuda.Event() for _ in range(4))
            for ev in (ev_hot, ev_cold, ev_err):
                ev.record(main)                   # the initial exchange and state are in place once `main` gets here

        def on(stream):
            return torch.cuda.stream(stream) if cuda else _NullCtx()

        def stages():
            if cuda:
                main.wait_event(ev_hot)
            L.check(lib.pgh_dist_partial_stage(g._h, bufs.v_xg_full._h, state, 1))
            if cuda:
                main.wait_event(ev_cold)
            L.check(lib.pgh_dist_partial_stage(g._h, bufs.v_xg_full._h, state, 2))

        t0 = time.perf_counter()
        it, spmv, converged = 1, 0, False       # `it` = ConvergenceManager.iteration of the pending has_converged call
        pending = False                         # a residual check whose outcome the host has not looked at yet
        staged = False
        while it < self.max_iters:                                             # convergence.py:86
            nxt = 1 - cur
            if not staged:
                stages()
            staged = False
            if pending:
                # the check that followed the previous step: its flag travels while the stages above run
                if cuda:
                    _bounded_wait(ev_err, "the residual all-reduce of the previous step")
                pending = False
                if int(bufs.state_host.view(torch.int32)[6]):
                    converged = True            # whatever the stages above computed is never folded into an iterate
                    break
            if cuda:
                main.wait_event(ev_err)         # quotient and done flag of the previous step; its residual has read y[nxt]
            self._combine(lib, g, p, bufs.v_y[nxt], bufs.v_xg_local, state)
            if cuda:
                ev_fin.record(main)
            with on(xs if cuda else None):                                        # ---- X: the next gather vector over xGMI
                if cuda:
                    xs.wait_event(ev_fin)
                bufs.all_gather(dist, "hot", g)
                if cuda:
                    ev_hot.record(xs)
                bufs.all_gather(dist, "cold", g)
                if cuda:
                    ev_cold.record(xs)
            cur = nxt
            spmv += 1
            it += 1
            check = it < self.max_iters and kind != L.ERR_ITERS and it % self.end_modulo == 0
            with on(ss if cuda else None):                                        # ---- S: the scalars of the step
                if cuda:
                    ss.wait_event(ev_fin)
                    L.check(lib.pgh_set_stream(C.c_void_p(ss.cuda_stream)))
                try:
                    dist.all_reduce(sum_view)
                    L.check(lib.pgh_dist_close_sum(state, 1 if self.use_quotient else 0))
                    if check:
                        L.check(lib.pgh_dist_residual(local_kind, bufs.v_y[cur]._h, bufs.v_y[1 - cur]._h, state))
                        dist.all_reduce(err_view, op=err_op)
                        L.check(lib.pgh_dist_close_err(state, kind, tol, pgraph.n_nodes))
                        bufs.state_host.copy_(bufs.state, non_blocking=True)
                finally:
                    if cuda:
                        L.check(lib.pgh_set_stream(C.c_void_p(main.cuda_stream)))
                if cuda:
                    ev_err.record(ss)
            if it >= self.max_iters:
                break
            if check:
                pending = True
                if cuda:
                    stages()                    # speculate: the next step's first two stages need the exchange only
                    staged = True
        if cuda:
            main.wait_event(ev_err)
            main.wait_event(ev_cold)
        if pending and not converged:
            if cuda:
                _bounded_wait(ev_err, "the residual all-reduce of the last step")
            converged = bool(int(bufs.state_host.view(torch.int32)[6]))
        done, steps, conv, scale = read_state()
        self.elapsed = time.perf_counter() - t0
        assert steps == spmv, (steps, spmv)
        self.iteration, self.spmv, self.converged = it, spmv, converged
        self.last_error = float(bufs.state_host[6])
        bufs.engine_streams = None
        self.exchange = dict(exchange_bytes_per_iteration_per_gpu=bufs.exchange_bytes, gather_vector_slots=bufs.nb * bufs.live,
                             column_blocks=bufs.nb, split_regions=False, driver="python (torch.distributed)",
                             exchange="need lists (all_to_all)" if bufs.lists else "all-gather", compact_cold_image=bufs.compact,
                             **({"exchange_slots": bufs.exchange_slots} if bufs.lists else {}))
        if not converged and self.error_type != "iters" and it >= self.max_iters:
            raise Exception("Could not converge within " + str(self.max_iters) + " iterations")
        factor = scale * (norm if self.preserve_norm else 1.0)                 # abstract_filters.py:63-64
        out = bufs.v_y[cur] * factor
        L.check(lib.pgh_sync())
        return out


class DistributedAbsorbingWalks(DistributedPageRank):
    """AbsorbingWalks(alpha) (adhoc.py:125-174) on a PartitionedGraph: the loop, the exchange and the stopping rule of
    DistributedPageRank with the step ``(conv(ranks, M) * deg + p * lam) / (lam + deg)``, deg = degrees(M) and lam = absorption *
    (1 - alpha) / alpha on this rank's rows (pgh_dist_combine_absorb).  absorption: this rank's slice (new ids) or None = 1."""
    def __init__(self, alpha=1 - 1.e-6, absorption=None, **kwargs):
        super().__init__(alpha=alpha, **kwargs)
        self.absorption = absorption

    def _operands(self, pgraph):
        """(deg, lam, every_row) on this rank's rows.  degrees(M) = row sums of M: a rank holds its COLUMNS of M (its rows of M^T), so
        its row sums are partial -- summed over the ranks (once per graph)."""
        import torch
        lo, m = pgraph.row_begin, pgraph.n_local
        if getattr(self, "_deg_for", None) is not pgraph:
            deg_all = torch.from_numpy(np.asarray(pgraph.graph.degrees(), dtype=np.float64)).to(self._device)
            self._dist.all_reduce(deg_all)
            self._deg, self._deg_for = DeviceVector.from_host(deg_all[lo:lo + m].cpu().numpy()), pgraph
        lam = (np.ones(m) if self.absorption is None else np.asarray(self.absorption, dtype=np.float64)) * ((1 - self.alpha) / self.alpha)
        self._lam = DeviceVector.from_host(lam)
        # an isolated row is p * lam / (lam + 0) = p: zero while p is zero there -- unless lam is zero too (0 / 0 in the reference):
        # such a run processes every row
        return self._deg, self._lam, not float(lam.min(initial=1.0)) > 0

    def _native_operands(self, pgraph):
        return self._operands(pgraph)

    def _prepare_run(self, pgraph, p, bufs, lib):
        _, _, every_row = self._operands(pgraph)
        watch = DeviceVector.from_host(np.ones(pgraph.n_local)) if every_row else bufs.v_y[0]
        L.check(lib.pgh_dist_watch_isolated(pgraph.graph._h, p._h, watch._h))

    def _combine(self, lib, g, p, y, xg_local, state):
        L.check(lib.pgh_dist_combine_absorb(g._h, p._h, self._deg._h, self._lam._h, y._h, xg_local._h, state))


class DistributedClosedFormFilter:
    """ClosedFormGraphFilter (abstract_filters.py:152-270, taylor form) on a PartitionedGraph: result = sum_k c_k (M^T)^(k-1) p with the
    stopping rule of ConvergenceManager on the change of the result (convergence.py:77-101).  `coefficient(previous, iteration)` is
    the filter's _coefficient (see DistributedHeatKernel / DistributedPageRankClosed).  One exchange per term: the all-gather of the
    term's gather slice, and one 8-byte all-reduce of the change.  The staged loop without run-ahead (the host reads the flag once per
    term): functional coverage of the filters next to PageRank, not a tuned path."""

    _KINDS = DistributedPageRank._KINDS
    _native_formula = True            # False (on an instance): always the staged Python loop

    def __init__(self, coefficient, tol=1e-6, error_type="mabs", max_iters=100, end_modulo=1, preserve_norm=True,
                 epsilon=float(np.finfo(np.float32).eps)):
        self.coefficient, self.tol, self.error_type = coefficient, tol, error_type
        self.max_iters, self.end_modulo, self.preserve_norm, self.epsilon = max_iters, end_modulo, preserve_norm, epsilon
        self.iteration, self.spmv, self._buffers = 0, 0, None

    def rank(self, pgraph, p_local):
        import torch
        import torch.distributed as dist
        lib, g = L.lib(), pgraph.graph
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        comm = _native_comm(dist, device) if self._native_formula else None
        if comm is not None:
            return self._rank_native(pgraph, p_local, comm, lib, dist)
        if self._buffers is None or self._buffers_for is not pgraph:
            self._buffers, self._buffers_for = _Buffers(pgraph, device, dist), pgraph
        bufs = self._buffers
        bufs.apply_bases(g)
        # one stream for the engine's launches and torch's collectives / tensor ops (as DistributedPageRank.rank)
        if device.type == "cuda":
            if getattr(self, "_stream", None) is None:
                self._stream = torch.cuda.Stream()
            L.check(lib.pgh_sync())
            L.check(lib.pgh_set_stream(C.c_void_p(self._stream.cuda_stream)))
            ctx = torch.cuda.stream(self._stream)
        else:
            ctx = _NullCtx()
        try:
            with ctx:
                return self._rank_on_stream(pgraph, p_local, bufs, dist, lib, g, device, torch)
        finally:
            if device.type == "cuda":
                L.check(lib.pgh_sync())
                L.check(lib.pgh_set_stream(None))

    def _rank_native(self, pgraph, p_local, comm, lib, dist):
        """The whole run as ONE engine call (pgh_dist_poly_run): the coefficient schedule is evaluated up front."""
        coeffs, c = [], None
        for it in range(1, max(int(self.max_iters), 2)):
            c = float(self.coefficient(c, it))
            coeffs.append(c)
        arr = (C.c_double * len(coeffs))(*coeffs)
        cfg = L.DistCfg(alpha=0.0, tol=0.0 if self.tol is None else max(float(self.tol), self.epsilon), n_global=int(pgraph.n_nodes),
                        err_kind=self._KINDS[self.error_type], max_iters=int(self.max_iters), end_modulo=int(self.end_modulo),
                        use_quotient=0, preserve_norm=1 if self.preserve_norm else 0, every_row=1, deg_local=None, lam_local=None)
        res = L.DistResult()
        out = DeviceVector.empty(pgraph.n_local)
        t0 = time.perf_counter()
        L.check(lib.pgh_dist_poly_run(pgraph.graph._h, comm, p_local._h, C.cast(arr, C.c_void_p), len(coeffs), out._h, C.byref(cfg),
                                      C.byref(res)))
        self.elapsed = time.perf_counter() - t0
        self.iteration, self.spmv, self.converged = int(res.iterations), int(res.spmv_count), bool(res.converged)
        self.last_error, self.loop_ms = float(res.last_error), float(res.loop_ms)
        self.exchange = dict(exchange_bytes_per_iteration_per_gpu=int(res.exchange_bytes), gather_vector_slots=int(res.gather_slots),
                             column_blocks=int(res.column_blocks), split_regions=bool(res.split_regions),
                             finish_in_two_launches=bool(res.flags & 4),
                             exchange={0: "all-gather", 8: "need lists (point to point)", 16: "all-gather + local compaction"}[res.flags & 24],
                             driver="engine (RCCL)" if dist.get_backend() == "nccl" else "engine (host collectives)")
        if res.iterations == 0:
            return p_local
        if not self.converged and self.error_type != "iters" and self.iteration >= self.max_iters:
            raise Exception("Could not converge within " + str(self.max_iters) + " iterations")
        return out

    def _rank_on_stream(self, pgraph, p_local, bufs, dist, lib, g, device, torch):
        self.exchange = dict(driver="python (torch.distributed)", exchange="need lists (all_to_all)" if bufs.lists else "all-gather")
        bufs.engine_streams = None
        kind = self._KINDS[self.error_type]
        tol = 0.0 if self.tol is None else max(self.tol, self.epsilon)
        linf = 1 if kind == L.ERR_LINF else 0
        scalar = lambda value, op: DistributedPageRank._all_reduce(self, bufs, dist, value, op)      # noqa: E731
        norm = scalar(p_local.abssum(), dist.ReduceOp.SUM)
        if norm == 0:
            self.iteration = 0
            return p_local
        state = C.c_void_p(bufs.state.data_ptr())
        err_view = bufs.state[1:2]
        n_local = pgraph.n_local
        res_t = torch.zeros(n_local, dtype=torch.float32, device=device)
        v_res = DeviceVector.wrap(res_t.data_ptr(), n_local, keepalive=res_t)
        p = p_local / norm
        c = self.coefficient(None, 1)                                            # iteration 1: result_1 = c_1 p
        L.check(lib.pgh_vec_copy(bufs.v_y[0]._h, p._h))
        first = p * float(c)                                                     # (kept alive until the copy has been enqueued)
        L.check(lib.pgh_vec_copy(v_res._h, first._h))
        L.check(lib.pgh_dist_prescale(g._h, bufs.v_y[0]._h, bufs.v_xg_local._h))
        bufs.all_gather(dist, "all", g)
        L.check(lib.pgh_dist_state_init(state))
        delta = abs(float(c)) * (1.0 if not linf else scalar(float(backend_max_abs(p)), dist.ReduceOp.MAX))
        if kind == L.ERR_MABS:
            delta /= pgraph.n_nodes
        cur, it, spmv, converged = 0, 2, 0, False
        t0 = time.perf_counter()
        while True:                                                              # `it` = the iteration has_converged is asked about
            if it >= self.max_iters:
                break
            if kind != L.ERR_ITERS and it % self.end_modulo == 0 and delta <= tol:
                converged = True
                break
            c = self.coefficient(c, it)
            nxt = 1 - cur
            L.check(lib.pgh_dist_partial_stage(g._h, bufs.v_xg_full._h, state, 0))
            L.check(lib.pgh_dist_combine_poly(g._h, bufs.v_y[cur]._h, bufs.v_y[nxt]._h, 1.0, 0.0, v_res._h, float(c), linf,
                                              bufs.v_xg_local._h, state))
            bufs.all_gather(dist, "all", g)
            dist.all_reduce(err_view, op=dist.ReduceOp.MAX if linf else dist.ReduceOp.SUM)
            bufs.state_host.copy_(bufs.state)                                    # the host decides (one read per term)
            delta = float(bufs.state_host[1]) / (pgraph.n_nodes if kind == L.ERR_MABS else 1)
            cur, spmv, it = nxt, spmv + 1, it + 1
        self.elapsed = time.perf_counter() - t0
        self.iteration, self.spmv, self.converged, self.last_error = it, spmv, converged, delta
        if not converged and self.error_type != "iters" and it >= self.max_iters:
            raise Exception("Could not converge within " + str(self.max_iters) + " iterations")
        out = v_res * (norm if self.preserve_norm else 1.0)
        L.check(lib.pgh_sync())
        return out


def backend_max_abs(vec):
    from pygrank_amd import backend
    return backend.max(backend.abs(vec)) if len(vec) else 0.0


class DistributedHeatKernel(DistributedClosedFormFilter):
    """HeatKernel(t) (adhoc.py:93-122) on a PartitionedGraph."""

    def __init__(self, t=3, **kwargs):
        super().__init__(lambda prev, it: 1.0 if prev is None else prev * t / (it + 1), **kwargs)     # adhoc.py:113-116
        self.t = t


class DistributedPageRankClosed(DistributedClosedFormFilter):
    """PageRankClosed(alpha) (adhoc.py:63-90) on a PartitionedGraph."""

    def __init__(self, alpha=0.85, **kwargs):
        super().__init__(lambda prev, it: 1.0 if prev is None else prev * alpha, **kwargs)           # adhoc.py:83-84
        self.alpha = alpha


def replica_columns(width, rank, world):
    """Rank `rank`'s contiguous share [lo, hi) of `width` feature columns; the shares differ by at most one column."""
    return width * rank // world, width * (rank + 1) // world


class _DeviceMemoryView:
    """Engine-owned device memory seen through the CUDA array interface (what torch.as_tensor wraps without a copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = dict(shape=tuple(int(s) for s in shape), typestr="<f4", data=(int(ptr), False), version=2)


class ReplicatedPropagation:
    """NodeRanking.propagate (pygrank/core/signals.py:225-226) across the GPUs of a node as a REPLICA SPLIT (last
    sentence: "cfg3-style batches can alternatively be split across GPUs with zero communication").  Every rank holds the whole
    graph (the scale-23 bench graph is 1 GB of images in 288 GB of HBM) and the same feature matrix; rank r runs the columns
    [r * B / P, (r + 1) * B / P) through `ranker.propagate` -- the single-GPU multi-seed loop (pgh_ppr_run_batch: every column keeps
    its own quotient, residual and stopping iteration) -- and nothing is exchanged until the result slabs are, optionally,
    all-gathered at the end.  `ranker` is any filter of pygrank_amd with a `propagate` (PageRank takes the batched route)."""

    def __init__(self, ranker):
        self.ranker = ranker
        self.columns, self.last_batches, self.elapsed = (0, 0), [], None

    def propagate(self, graph, features, *args, gather=True, **kwargs):
        """features: [n, B] (DeviceMatrix, array or list of columns), the same on every rank.  Returns the [n, B] ranks on every
        rank (gather=True: one all-gather of the result slabs) or this rank's [n, hi - lo] share (gather=False; `self.columns`)."""
        import torch
        import torch.distributed as dist
        from pygrank_amd import backend
        from pygrank_amd.device import DeviceMatrix
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
        F = features if isinstance(features, DeviceMatrix) else backend.to_primitive(features)
        if not isinstance(F, DeviceMatrix):
            F = DeviceMatrix.from_columns(list(F) if isinstance(F, (list, tuple)) else [F])
        lo, hi = replica_columns(F.b, rank, world)
        self.columns = (lo, hi)
        t0 = time.perf_counter()
        mine = None
        if hi > lo:
            share = F if (lo, hi) == (0, F.b) else F.get_cols(lo, hi - lo)
            mine = self.ranker.propagate(graph, share, *args, **kwargs)
            if not isinstance(mine, DeviceMatrix):          # the per-column fallback of NodeRanking.propagate returns signals
                mine = DeviceMatrix.from_columns([backend.to_primitive(getattr(col, "np", col)) for col in mine])
            self.last_batches = getattr(self.ranker, "last_batches", [])
        L.check(L.lib().pgh_sync())
        self.elapsed = time.perf_counter() - t0
        if not gather or (world == 1 and os.environ.get("PGH_REPLICA_GATHER_ALONE", "0") != "1"):     # (=1: tests run the all-gather with one rank)
            return mine
        return self._all_gather(F.n, F.b, mine, rank, world, dist, torch)

    def _all_gather(self, n, width, mine, rank, world, dist, torch):
        """Every rank's share -> the [n, width] slab on every rank.  Shares are padded to the widest one (equal counts per rank,
        ncclAllGather's contract); over RCCL the slabs travel device to device, with gloo (CPU tests, ranks sharing a GPU) through
        host memory."""
        from pygrank_amd.device import DeviceMatrix
        shares = [replica_columns(width, r, world) for r in range(world)]
        wmax = max(hi - lo for lo, hi in shares)
        out = DeviceMatrix.empty(n, width)
        lib = L.lib()
        on_device = dist.get_backend() == "nccl" and torch.cuda.is_available() and L.runtime_name().startswith("hip:")
        if on_device:
            device = torch.device("cuda", torch.cuda.current_device())
            recv = torch.zeros((world, n, wmax), dtype=torch.float32, device=device)
            try:
                if mine is not None:
                    view = torch.as_tensor(_DeviceMemoryView(lib.pgh_mat_ptr(mine._h), (n, mine.b)), device=device)
                    recv[rank, :, :mine.b].copy_(view)
                dist.all_gather_into_tensor(recv.view(-1), recv[rank].reshape(-1).clone())
                whole = torch.as_tensor(_DeviceMemoryView(lib.pgh_mat_ptr(out._h), (n, width)), device=device)
                for r, (lo, hi) in enumerate(shares):
                    if hi > lo:
                        whole[:, lo:hi].copy_(recv[r, :, :hi - lo])
                torch.cuda.synchronize(device)
                return out
            except (TypeError, RuntimeError, AttributeError) as exc:      # no zero-copy view on this torch build: through the host
                sys.stderr.write(f"[pygrank_amd.distributed] replica all-gather on the device failed ({str(exc)[:120]}); staging through the host\n")
        send = torch.zeros((n, wmax), dtype=torch.float32)
        if mine is not None:
            send[:, :mine.b] = torch.from_numpy(mine.numpy().astype(np.float32))
        recv = torch.zeros((world, n, wmax), dtype=torch.float32)
        if on_device:
            dev = torch.device("cuda", torch.cuda.current_device())
            recv_d = recv.to(dev)
            dist.all_gather_into_tensor(recv_d.view(-1), send.to(dev).view(-1))
            recv = recv_d.cpu()
        else:
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
        full = np.empty((n, width), dtype=np.float64)
        for r, (lo, hi) in enumerate(shares):
            full[:, lo:hi] = recv[r, :, :hi - lo].numpy()
        return DeviceMatrix.from_host(full)


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
