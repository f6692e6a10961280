// Row-partitioned PageRank with the whole loop behind ONE C-ABI call per run: the engine drives RCCL itself.
//
// What it replaces: the per-iteration choreography of pygrank_amd/distributed.py (three queues, four events, four
// torch.distributed collectives and a dozen engine calls per iteration from Python -- ~130 us of host time per iteration on
// top of 267 us of kernels at RMAT scale 23 with one rank, profiles/r03/partitioned_one_rank.log).  The loop semantics are
// those of GraphFilter.rank + RecursiveGraphFilter._step + ConvergenceManager (pygrank/algorithms/filters/
// abstract_filters.py:44-65,126-136; pygrank/algorithms/convergence.py:77-101); the reference has no distributed counterpart.
//
// Three queues per rank, as in the Python driver (which stays: gloo / CPU runs, and the fallback of this file):
//   C  compute stream: block partial sums -> cold image phase A -> phase B + epilogue of every step;
//   X  exchange stream + communicator: after the epilogue of step k, ncclAllGather of the HOT prefixes of the gather slices
//      (what the next step's block partial sums read), then of the cold parts, which travel while those sums run.  The two
//      parts of a block's slice live in two REGIONS of the gather vector ([j][rank][hot] | [j][rank][live - hot]), so every
//      exchange is one in-place-shaped all-gather on contiguous memory (the list form of torch.distributed staged the 141 MB
//      of configs[4] through a temporary);
//   S  scalar stream + communicator of its own: ncclAllReduce of sum(y) -> lazy L1 quotient -> residual -> ncclAllReduce ->
//      stopping rule on the device -> 64 bytes of state to pinned host memory.
// The host never waits inside an iteration: it reads the done flag of step k after it has enqueued the first two stages of
// step k + 1; every kernel of the loop is a no-op once the flag is set.  Host waits are bounded (PGH_DIST_TIMEOUT_S): a
// collective that never completes ends in an error return, not in a hang.
//
// RCCL is loaded at run time (dlopen of librccl.so.1 -- the copy torch has already mapped when there is one), so the engine
// library itself has no link-time dependency on it.
#include "pgh_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <thread>

using namespace pgh;

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.handle != nullptr) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* name : names) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h != nullptr) break;
    }
    PGH_CHECK(h != nullptr, std::string("pgh_comm: cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : ""));
#define PGH_RCCL_SYM(FIELD, NAME)                                                            \
    g_rccl.FIELD = reinterpret_cast<decltype(g_rccl.FIELD)>(dlsym(h, NAME));                \
    PGH_CHECK(g_rccl.FIELD != nullptr, std::string("pgh_comm: RCCL has no symbol ") + NAME);
    PGH_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    PGH_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    PGH_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    PGH_RCCL_SYM(AllGather, "ncclAllGather")
    PGH_RCCL_SYM(AllReduce, "ncclAllReduce")
    PGH_RCCL_SYM(Send, "ncclSend")
    PGH_RCCL_SYM(Recv, "ncclRecv")
    PGH_RCCL_SYM(GroupStart, "ncclGroupStart")
    PGH_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    PGH_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PGH_RCCL_SYM
    g_rccl.handle = h;
    return 0;
}

#define PGH_RCCL(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess)                                                                                 \
            return ::pgh::fail(std::string(#expr) + ": " + g_rccl.GetErrorString(_r) + " (" + __FILE__ + ":" + \
                               std::to_string(__LINE__) + ")");                                                \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PGH_COMM_ID_BYTES, "pgh_comm_unique_id hands out ncclUniqueId bytes");

}  // namespace

struct pgh_comm_s {
    ncclComm_t  x = nullptr;        // gather-vector exchange
    ncclComm_t  s = nullptr;        // scalar reductions (== x with a single communicator)
    int         world = 1, rank = 0;
    hipStream_t main = nullptr, xs = nullptr, ss = nullptr;      // xs == ss == main: PGH_DIST_SINGLE_STREAM
    bool        own_streams = false;
    hipEvent_t  ev_fin = nullptr, ev_fin2 = nullptr, ev_hot = nullptr, ev_cold = nullptr, ev_err = nullptr, ev_host = nullptr;
    // buffers of the graph the communicator last ran on
    pgh_graph_t graph = nullptr;
    int         nb = 0, bpr = 0, live = 0, hot = 0;
    int64_t     blk = 0, n_xg = 0, n_local = 0, buf_local = 0;
    float*      xg_full = nullptr;
    float*      xg_local = nullptr;   // this rank's slice inside xg_full (prepare_graph)
    int64_t     local_off = 0, cold_rel = 0;
    float*      y[2] = {nullptr, nullptr};
    float*      p_norm = nullptr;
    double*     state = nullptr;         // [8] device (pgh_dist_* layout)
    double*     state_host = nullptr;    // [8] pinned
    LoopAux*    aux = nullptr;           // the in-kernel residual's predictions (ResParams, pgh_kernels.h)
    double*     red = nullptr;           // [4] {S, T, D, R'}: this rank's sums, all-reduced in place once per iteration
    int32_t*    agree = nullptr;         // [4] device words of the layout negotiation
    // host-visible word of the loop: {steps closed << 32 | done flag}, written by the closes of a step (publish_progress) into pinned
    // mapped memory -- the host paces itself by it instead of copying the 64-byte state back after every check
    volatile unsigned long long* progress_host = nullptr;
    unsigned long long*          progress_dev = nullptr;
    bool        one_gather = false;      // single queue: the whole packed slice travels as ONE all-gather (nothing to overlap with)
    // collectives supplied by the host instead of RCCL (pgh_comm_create_external): MPI, gloo, a test harness ...
    pgh_allgather_fn ext_gather = nullptr;
    pgh_allreduce_fn ext_reduce = nullptr;
    pgh_alltoallv_fn ext_alltoallv = nullptr;
    void*            ext_user = nullptr;
    // need lists (pgh_dist_need_counts): the cold parts travel point to point, every rank receives the slots its slice references
    bool        lists = false;           // this run exchanges the cold parts by need lists
    bool        compact_copy = false;    // dense exchange, compact image: the slice copies its slots out of the gathered vector
    int64_t     need_total = 0, compact_at = 0, dense_cold_at = 0;
    int64_t     send_counts[8] = {0}, send_offs[8] = {0}, recv_counts[8] = {0}, recv_offs[8] = {0};
    float*      send_buf = nullptr;
    int64_t     send_cap = 0;
    unsigned long long lists_stamp = 0;
    int64_t     dense_cold_base[8] = {0}, compact_base[8] = {0};
    int64_t     exchange_bytes = 0;     // received per rank and iteration with the layout of the last prepare_graph
};

namespace {

// one all-gather / all-reduce of the run: RCCL on `st`, or the host's callback (which completes the exchange in stream order on `st`
// before it returns: dtype 0 = f32, 1 = f64, 2 = i32; op 0 = sum, 1 = max)
int comm_all_gather(pgh_comm_s* c, const void* send, void* recv, size_t count, ncclComm_t comm, hipStream_t st) {
    // (a rank alone: its slice already lies where the gathered vector wants it; PGH_DIST_GATHER_ALONE=1 makes the call all the same,
    // so that one GPU can exercise RCCL's all-gather: tests)
    if (c->world == 1 && c->ext_gather == nullptr && !(getenv("PGH_DIST_GATHER_ALONE") != nullptr && atoi(getenv("PGH_DIST_GATHER_ALONE")) != 0))
        return 0;
    if (c->ext_gather != nullptr) {
        PGH_CHECK(c->ext_gather(c->ext_user, send, recv, (int64_t)count, 0, (void*)st) == 0, "pgh_dist_ppr_run: the host's all-gather callback failed");
        return 0;
    }
    PGH_RCCL(g_rccl.AllGather(send, recv, count, ncclFloat32, comm, st));
    return 0;
}
int comm_all_reduce(pgh_comm_s* c, void* buf, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    // (a rank alone reduces with nobody: in place, the result is already there -- one launch fewer in the dependent chain of a step)
    if (c->world == 1 && c->ext_reduce == nullptr && !(getenv("PGH_DIST_REDUCE_ALONE") != nullptr && atoi(getenv("PGH_DIST_REDUCE_ALONE")) != 0)) return 0;
    if (c->ext_reduce != nullptr) {
        const int32_t dtype = dt == ncclFloat32 ? 0 : (dt == ncclFloat64 ? 1 : 2);
        PGH_CHECK(c->ext_reduce(c->ext_user, buf, (int64_t)count, dtype, op == ncclMax ? 1 : 0, (void*)st) == 0,
                  "pgh_dist_ppr_run: the host's all-reduce callback failed");
        return 0;
    }
    PGH_RCCL(g_rccl.AllReduce(buf, buf, count, dt, op, comm, st));
    return 0;
}

inline bool p2p_alone() { return getenv("PGH_DIST_P2P_ALONE") != nullptr && atoi(getenv("PGH_DIST_P2P_ALONE")) != 0; }

// every rank's stretch for every other rank, point to point (4-byte elements; the stretch a rank keeps for itself is a device copy)
int comm_all_to_all_v(pgh_comm_s* c, const void* send, const int64_t* scounts, const int64_t* soffs, void* recv, const int64_t* rcounts,
                      const int64_t* roffs, hipStream_t st) {
    if (c->ext_alltoallv != nullptr) {
        PGH_CHECK(c->ext_alltoallv(c->ext_user, send, scounts, soffs, recv, rcounts, roffs, (void*)st) == 0,
                  "pgh_dist: the host's all-to-all callback failed");
        return 0;
    }
    const char* sb = static_cast<const char*>(send);
    char* rb = static_cast<char*>(recv);
    // PGH_DIST_P2P_ALONE=1 (tests; a box with one GPU): the stretch a rank keeps for itself travels through the grouped ncclSend / ncclRecv
    // pair like every other one -- RCCL pairs a send to the own rank with the receive of the same group -- so that one GPU executes the
    // point-to-point path the N-rank exchange consists of
    const bool self_p2p = p2p_alone() && c->ext_gather == nullptr;
    if (scounts[c->rank] > 0 && !self_p2p)
        PGH_HIP(hipMemcpyAsync(rb + 4 * roffs[c->rank], sb + 4 * soffs[c->rank], 4 * (size_t)scounts[c->rank], hipMemcpyDeviceToDevice, st));
    if (c->world == 1 && !self_p2p) return 0;
    PGH_CHECK(c->ext_gather == nullptr, "pgh_dist: a host-collective communicator without an all-to-all callback cannot exchange need lists");
    PGH_RCCL(g_rccl.GroupStart());
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank && !self_p2p) continue;
        if (scounts[r] > 0) PGH_RCCL(g_rccl.Send(sb + 4 * soffs[r], (size_t)scounts[r], ncclFloat32, r, c->x, st));
        if (rcounts[r] > 0) PGH_RCCL(g_rccl.Recv(rb + 4 * roffs[r], (size_t)rcounts[r], ncclFloat32, r, c->x, st));
    }
    PGH_RCCL(g_rccl.GroupEnd());
    return 0;
}

void free_buffers(pgh_comm_s* c) {
    (void)hipFree(c->send_buf);
    c->send_buf = nullptr;
    c->send_cap = 0;
    c->lists_stamp = 0;
    (void)hipFree(c->xg_full);         // (xg_local points into it)
    (void)hipFree(c->y[0]);
    (void)hipFree(c->y[1]);
    (void)hipFree(c->p_norm);
    c->xg_full = c->xg_local = c->y[0] = c->y[1] = c->p_norm = nullptr;
    c->graph = nullptr;
}

// host wait with a deadline: 0 = the event completed
double default_wait_limit_s() { return getenv("PGH_DIST_TIMEOUT_S") != nullptr ? atof(getenv("PGH_DIST_TIMEOUT_S")) : 600.0; }
double& wait_limit_s() {
    static double limit = default_wait_limit_s();
    return limit;
}
int bounded_wait(hipEvent_t ev, const char* what) {
    const double limit = wait_limit_s();
    const auto start = std::chrono::steady_clock::now();
    long spins = 0;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail(std::string("pgh_dist_ppr_run: waiting for ") + what + ": " + hipGetErrorString(e));
        if (++spins > 4000) {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
            if (waited > limit)
                return fail(std::string("pgh_dist_ppr_run: ") + what + " did not complete within PGH_DIST_TIMEOUT_S -- a collective is "
                            "stalled (a peer gone, or communicators blocking each other: retry with a single communicator / stream)");
        }
    }
}

__global__ void k_scale_into(const float* __restrict__ in, float* __restrict__ out, int64_t n, double factor) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)((double)in[i] * factor);
}
__global__ void k_div_into(const float* __restrict__ in, float* __restrict__ out, int64_t n, float norm) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i] / norm;
}

inline int grid_of(int64_t n) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)rt().num_cus * 16;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

struct StreamSwap {          // the engine launches on rt().stream: point it at one of the communicator's queues for a scope
    hipStream_t saved;
    explicit StreamSwap(hipStream_t s) : saved(rt().stream) { rt().stream = s; }
    ~StreamSwap() { rt().stream = saved; }
};

// gather-vector layout and buffers for graph g (collective: every rank calls it with its slice of the same partition).  The
// layout is negotiated on every run (one 8-byte all-reduce: a handle is no proof that the graph behind it is the one of the
// last run); the buffers are kept while their sizes fit.
unsigned long long g_lists_stamp = 0;

// once per graph and communicator: every rank tells the owners of the blocks which of their cold slots its slice references; what the
// peers asked of this rank becomes the graph's send lists (one pack launch per step), the counts become the step's point-to-point sizes
int setup_need_lists(pgh_comm_s* c, pgh_graph_t g, const int64_t* need8) {
    BsfFormat& f = g->bsf;
    const int nb = c->nb, bpr = c->bpr, world = c->world;
    int64_t need_prefix[9] = {0};
    for (int b = 0; b < 8; ++b) need_prefix[b + 1] = need_prefix[b] + (b < nb ? need8[b] : 0);
    int64_t* d_counts = nullptr;
    PGH_HIP(hipMalloc(&d_counts, sizeof(int64_t) * (size_t)(world * nb)));
    std::vector<int64_t> counts_all((size_t)world * nb, 0);
    uint32_t* d_asked = nullptr;
    int rc = 0;
    // (after a bounded wait has given up on a stalled collective nothing may be freed -- hipFree waits for every queue of the
    // device, which turns the bounded exit into a hang: the two buffers are then left to the process, like RunScope and _preflight do)
    bool stalled = false;
    auto bail = [&](int code) {
        if (!stalled) {
            (void)hipFree(d_counts);
            (void)hipFree(d_asked);
        }
        return code;
    };
    auto waited = [&](int code) {
        stalled = code != 0;
        return code;
    };
    if (hipMemcpyAsync(d_counts + (int64_t)c->rank * nb, need8, sizeof(int64_t) * (size_t)nb, hipMemcpyHostToDevice, c->main) != hipSuccess)
        return bail(fail("setup_need_lists: copy failed"));
    if ((rc = comm_all_gather(c, d_counts + (int64_t)c->rank * nb, d_counts, (size_t)nb * 2, c->x, c->main)) != 0) return bail(rc);
    if (hipMemcpyAsync(counts_all.data(), d_counts, sizeof(int64_t) * counts_all.size(), hipMemcpyDeviceToHost, c->main) != hipSuccess ||
        hipEventRecord(c->ev_host, c->main) != hipSuccess)
        return bail(fail("setup_need_lists: copy failed"));
    if ((rc = waited(bounded_wait(c->ev_host, "the exchange of the need-list sizes"))) != 0) return bail(rc);
    int64_t scounts[8] = {0}, soffs[8] = {0}, rcounts[8] = {0}, roffs[8] = {0}, asked_total = 0;
    for (int s = 0; s < world; ++s) {
        soffs[s] = need_prefix[s * bpr];
        scounts[s] = need_prefix[(s + 1) * bpr] - need_prefix[s * bpr];
        roffs[s] = asked_total;
        for (int j = 0; j < bpr; ++j) rcounts[s] += counts_all[(size_t)s * nb + c->rank * bpr + j];
        asked_total += rcounts[s];
    }
    if (asked_total >= (1LL << 31)) return bail(fail("setup_need_lists: the send lists are too long"));
    if (hipMalloc(&d_asked, sizeof(uint32_t) * (size_t)(asked_total > 0 ? asked_total : 1)) != hipSuccess) return bail(fail("setup_need_lists: out of device memory"));
    if ((rc = comm_all_to_all_v(c, f.need_idx, scounts, soffs, d_asked, rcounts, roffs, c->main)) != 0) return bail(rc);
    if (hipEventRecord(c->ev_host, c->main) != hipSuccess) return bail(fail("setup_need_lists: event"));
    if ((rc = waited(bounded_wait(c->ev_host, "the exchange of the need lists"))) != 0) return bail(rc);
    std::vector<int64_t> seg_off((size_t)world * bpr + 1, 0);
    std::vector<int32_t> seg_block((size_t)world * bpr, 0);
    for (int rk = 0; rk < world; ++rk)
        for (int j = 0; j < bpr; ++j) {
            const size_t k = (size_t)rk * bpr + j;
            seg_block[k] = j;
            seg_off[k + 1] = seg_off[k] + counts_all[(size_t)rk * nb + c->rank * bpr + j];
        }
    if ((rc = dist_set_send_lists_device(g, d_asked, seg_block.data(), seg_off.data(), world * bpr)) != 0) return bail(rc);
    for (int s = 0; s < 8; ++s) {
        c->send_counts[s] = s < world ? rcounts[s] : 0;      // what rank s asked of this rank
        c->send_offs[s] = s < world ? roffs[s] : 0;
        c->recv_counts[s] = s < world ? scounts[s] : 0;      // what this rank asked of rank s: its blocks' stretch of the compact region
        c->recv_offs[s] = s < world ? soffs[s] : 0;
    }
    if (c->send_cap < asked_total) {
        (void)hipFree(c->send_buf);
        c->send_buf = nullptr;
        c->send_cap = 0;
        if (hipMalloc(&c->send_buf, sizeof(float) * (size_t)(asked_total > 0 ? asked_total : 1)) != hipSuccess) return bail(fail("setup_need_lists: out of device memory"));
        c->send_cap = asked_total > 0 ? asked_total : 1;
    }
    f.send_stamp = ++g_lists_stamp;
    c->lists_stamp = f.send_stamp;
    return bail(0);
}

int prepare_graph(pgh_comm_s* c, pgh_graph_t g, bool* fused) {
    Runtime& r = rt();
    int32_t nb = 0, live8[8] = {0}, hot_slots = 0;
    int64_t blk = 0, need8[8] = {0};
    PGH_TRY(pgh_graph_gather_layout(g, &nb, &blk, live8));
    PGH_TRY(pgh_graph_hot_prefix(g, &hot_slots));
    PGH_TRY(pgh_dist_need_counts(g, need8));
    PGH_CHECK(nb % c->world == 0, "pgh_dist_ppr_run: the column blocks of the slice do not divide among the ranks");
    int32_t top = 0;
    for (int b = 0; b < nb; ++b) top = live8[b] > top ? live8[b] : top;
    const bool compact = g->bsf.need_idx != nullptr;        // the slice numbers its cold sources compactly (need lists)
    // every rank must lay the gather vector out the same way: live = max over ranks, hot prefix = min over ranks; and every rank
    // must close the steps the same way: the in-kernel residual only when every slice can (min), the slice's degrees recomputed
    // when any rank lacks them (max); need lists only when every slice is compact
    // (the degree word counts only where this rank could fuse at all: a run that cannot -- closed-form filters, AbsorbingWalks, the
    // max rule -- neither needs the slice's degrees nor gives anybody's up)
    const bool may_fuse = fused != nullptr && *fused;
    int32_t h_agree[5] = {top, -hot_slots, may_fuse ? 0 : 1, (may_fuse && g->bsf.deg_int == nullptr) ? 1 : 0, compact ? 0 : 1};
    PGH_HIP(hipMemcpyAsync(c->agree, h_agree, sizeof(h_agree), hipMemcpyHostToDevice, c->main));
    PGH_TRY(comm_all_reduce(c, c->agree, 5, ncclInt32, ncclMax, c->s, c->main));
    PGH_HIP(hipMemcpyAsync(h_agree, c->agree, sizeof(h_agree), hipMemcpyDeviceToHost, c->main));
    PGH_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_TRY(bounded_wait(c->ev_host, "the layout negotiation"));
    if (fused != nullptr) *fused = h_agree[2] == 0;
    // some rank rebuilds its degrees and the run fuses on every rank: the all-reduce of ensure_slice_degrees needs everybody
    if (h_agree[2] == 0 && h_agree[3] != 0 && g->bsf.deg_int != nullptr) {
        (void)pooled_free(g->bsf.deg_int);
        g->bsf.deg_int = nullptr;
        g->bsf.device_bytes -= (int64_t)g->n_cols * 4;
    }
    const int64_t live = std::min<int64_t>(blk, ((int64_t)h_agree[0] + 63) / 64 * 64);
    const int64_t hot_all = -h_agree[1];
    // split regions only when the exchange's hot prefix is exactly what the block partial sums read
    const bool split = hot_all > 0 && hot_all < live && hot_all % 64 == 0 && hot_all == hot_slots &&
                       !(getenv("PGH_DIST_SPLIT") != nullptr && atoi(getenv("PGH_DIST_SPLIT")) == 0);
    // need lists: every slice compact, the hot prefixes a region of their own, a communicator that can move stretches point to point
    const bool can_p2p = c->ext_gather == nullptr || c->ext_alltoallv != nullptr;
    const char* xenv = getenv("PGH_DIST_EXCHANGE");
    // a rank ALONE references every live slot of its own blocks (the relabelling sorts by reference count: nothing unreferenced lies below
    // `live`), so its compact numbering is the dense one: the slice is written in place as ever and nothing is packed or copied
    // (PGH_DIST_P2P_ALONE=1: the lone rank goes through compact numbering -> pack launch -> point-to-point transfers to itself all the same)
    bool identity = compact && c->world == 1 && split && !(p2p_alone() && c->ext_gather == nullptr);
    for (int b = 0; b < nb && identity; ++b) identity = need8[b] == (live8[b] > hot_slots ? live8[b] - hot_slots : 0);
    const bool lists = compact && !identity && h_agree[4] == 0 && split && can_p2p && !(xenv != nullptr && std::string(xenv) == "allgather");
    c->nb = nb;
    c->blk = blk;
    c->bpr = nb / c->world;
    c->live = (int)live;
    c->hot = split ? (int)hot_all : 0;
    c->n_local = g->n_cols;
    c->lists = lists;
    c->compact_copy = compact && !lists && !identity;
    PGH_CHECK(blk * nb == g->n_rows && c->n_local == (int64_t)c->bpr * blk, "pgh_dist_ppr_run: the slice does not match the block layout");
    int64_t need_prefix[9] = {0};
    for (int b = 0; b < 8; ++b) need_prefix[b + 1] = need_prefix[b] + (b < nb ? need8[b] : 0);
    c->need_total = compact ? need_prefix[nb] : 0;
    // A rank keeps its slice of the next gather vector PACKED for the exchange -- the hot prefixes of its blocks one after the
    // other, then their cold parts (dist_set_local_layout: the epilogue writes it that way) -- so every exchange is ONE all-gather
    // per region however many blocks a rank owns, and block b = rank * bpr + j of the gathered vector starts at b * (region width).
    // (single queue: the whole packed slice is one all-gather, [rank][hot prefixes | cold parts]; the bases say where a block's two
    // parts landed)
    int64_t hot_bases[8] = {0}, cold_bases[8] = {0};
    int64_t n_xg = 0;
    if (lists) {
        // [block][hot] | [block][the cold slots THIS slice references] | this rank's own cold parts [j][live - hot]: the hot prefixes are
        // all-gathered in place, the cold parts are packed per destination (pgh_dist_pack) and travel point to point
        for (int b = 0; b < nb; ++b) {
            hot_bases[b] = (int64_t)b * c->hot;
            cold_bases[b] = (int64_t)nb * c->hot + need_prefix[b];
        }
        c->compact_at = (int64_t)nb * c->hot;
        c->dense_cold_at = (c->compact_at + c->need_total + 63) / 64 * 64;
        c->local_off = (int64_t)c->rank * c->bpr * c->hot;
        c->cold_rel = c->dense_cold_at - c->local_off;
        n_xg = c->dense_cold_at + (int64_t)c->bpr * (live - c->hot) + 32768;
        PGH_TRY(pgh_graph_set_gather_bases_split(g, hot_bases, cold_bases));
    } else {
        for (int b = 0; b < nb; ++b) {
            if (c->one_gather && split) {
                const int rk = b / c->bpr, j = b % c->bpr;
                hot_bases[b] = (int64_t)rk * c->bpr * live + (int64_t)j * c->hot;
                cold_bases[b] = (int64_t)rk * c->bpr * live + (int64_t)c->bpr * c->hot + (int64_t)j * (live - c->hot);
            } else {
                hot_bases[b] = (int64_t)b * (split ? c->hot : live);
                cold_bases[b] = (int64_t)nb * c->hot + (int64_t)b * (live - c->hot);
            }
        }
        n_xg = (int64_t)nb * live;
        if (c->compact_copy) {
            // dense exchange, compact image (some peer's slice is dense, or the host's collectives cannot move stretches): the gathered
            // vector is laid out as ever and this slice copies the slots it references out of it, block by block, behind every exchange
            c->compact_at = (n_xg + 63) / 64 * 64;
            int64_t compact_bases[8] = {0};
            for (int b = 0; b < nb; ++b) {
                c->dense_cold_base[b] = split ? cold_bases[b] : hot_bases[b] + hot_slots;
                compact_bases[b] = c->compact_at + need_prefix[b];
                c->compact_base[b] = compact_bases[b];
            }
            n_xg = c->compact_at + c->need_total;
            PGH_TRY(pgh_graph_set_gather_bases_split(g, hot_bases, compact_bases));
        } else if (split) {
            PGH_TRY(pgh_graph_set_gather_bases_split(g, hot_bases, cold_bases));
        } else {
            PGH_TRY(pgh_graph_set_gather_bases(g, hot_bases));
        }
        // The slice is written IN PLACE: the epilogue of a step stores this rank's slots straight into their places of the gathered
        // vector (ncclAllGather's in-place form: send = recv + rank * count), so no collective copies a rank's own slots and a rank alone
        // has nothing to exchange at all.  One region: the packed slice [hot prefixes | cold parts] is rank r's stretch of the vector;
        // two regions: its hot prefixes are rank r's stretch of the hot region, its cold parts rank r's stretch of the cold region.
        // (Safe: a step's finish kernel is the last reader-free point of the vector on this rank -- the block partial sums and phase A of
        // the step have run, the next all-gather is enqueued behind it -- and peers write only their own stretches.)
        const bool two_regions = split && !c->one_gather;
        c->local_off = two_regions ? (int64_t)c->rank * c->bpr * c->hot : (int64_t)c->rank * c->bpr * live;
        c->cold_rel = two_regions ? (int64_t)nb * c->hot + (int64_t)c->rank * c->bpr * (live - c->hot) - c->local_off : (int64_t)c->bpr * c->hot;
        n_xg += 32768;                                     // + the hot cache's read-ahead past a short block
    }
    PGH_CHECK(c->cold_rel >= 0 && c->cold_rel < (1LL << 31), "pgh_dist_ppr_run: the gather vector is too long for the slice layout");
    PGH_TRY(dist_set_local_layout(g, (int)live, c->hot, (int)c->cold_rel));      // (after the bases: setting them resets the layout)
    c->exchange_bytes = lists ? 4LL * ((int64_t)c->hot * c->bpr * (c->world - 1)) : 4LL * c->live * c->bpr * (c->world - 1);
    if (!(c->graph != nullptr && c->n_xg == n_xg && c->buf_local == c->n_local)) {
        free_buffers(c);
        c->n_xg = n_xg;
        c->buf_local = c->n_local;
        PGH_HIP(hipMalloc(&c->xg_full, sizeof(float) * (size_t)c->n_xg));
        PGH_HIP(hipMemsetAsync(c->xg_full, 0, sizeof(float) * (size_t)c->n_xg, r.stream));
        PGH_HIP(hipMalloc(&c->y[0], sizeof(float) * (size_t)c->n_local));
        PGH_HIP(hipMalloc(&c->y[1], sizeof(float) * (size_t)c->n_local));
        PGH_HIP(hipMalloc(&c->p_norm, sizeof(float) * (size_t)c->n_local));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    c->xg_local = c->xg_full + c->local_off;
    c->graph = g;
    if (lists) {
        // the send lists are state of the GRAPH, the point-to-point sizes state of the communicator: both are this pair's when the
        // stamp the last set-up left on the graph is the communicator's (a Python-driven run in between re-registers its own lists)
        if (g->bsf.send_stamp == 0 || g->bsf.send_stamp != c->lists_stamp) PGH_TRY(setup_need_lists(c, g, need8));
        int64_t received = 0;
        for (int s = 0; s < c->world; ++s) received += s == c->rank ? 0 : c->recv_counts[s];
        c->exchange_bytes += 4LL * received;
    }
    return 0;
}

// all-gather of slots [lo, hi) of every block of this rank's slice into `region` ([rank][j][hi - lo]) on stream `st`: ONE collective
// (the slice is stored packed, lo == 0: from its start, else from the cold part on)
int gather_part(pgh_comm_s* c, int64_t region, int lo, int hi, hipStream_t st) {
    if (hi <= lo) return 0;
    const int64_t len = (int64_t)(hi - lo) * c->bpr;
    const int64_t from = lo == 0 ? 0 : c->cold_rel;
    return comm_all_gather(c, c->xg_local + from, c->xg_full + region, (size_t)len, c->x, st);
}

// the exchange of a step: this rank's packed slice of the next gather vector -> every rank's xg_full.  Three queues: the hot
// prefixes first (ev_hot: all that the next step's block partial sums read), then the cold bulk (ev_cold), so that the bulk travels
// while those sums run; one queue: ONE all-gather of the whole slice.  Need lists: the cold bulk is ONE pack launch (every
// destination's stretch) and one group of point-to-point transfers into the compact cold region.
int exchange_slices(pgh_comm_s* c, hipStream_t st, bool events) {
    pgh_vec_s v_full{c->xg_full, c->n_xg, false};
    if (c->lists) {
        PGH_TRY(gather_part(c, 0, 0, c->hot, st));
        if (events) PGH_HIP(hipEventRecord(c->ev_hot, st));
        {
            StreamSwap on_exchange(st);
            pgh_vec_s v_local{c->xg_local, c->n_xg - c->local_off, false}, v_send{c->send_buf, c->send_cap, false};
            PGH_TRY(pgh_dist_pack(c->graph, &v_local, &v_send));
        }
        PGH_TRY(comm_all_to_all_v(c, c->send_buf, c->send_counts, c->send_offs, c->xg_full + c->compact_at, c->recv_counts, c->recv_offs, st));
        if (events) PGH_HIP(hipEventRecord(c->ev_cold, st));
        return 0;
    }
    if (c->one_gather || c->hot == 0) {
        PGH_TRY(comm_all_gather(c, c->xg_local, c->xg_full, (size_t)((int64_t)c->bpr * c->live), c->x, st));
        if (events && !c->compact_copy) PGH_HIP(hipEventRecord(c->ev_hot, st));
    } else {
        PGH_TRY(gather_part(c, 0, 0, c->hot, st));
        if (events) PGH_HIP(hipEventRecord(c->ev_hot, st));
        PGH_TRY(gather_part(c, (int64_t)c->nb * c->hot, c->hot, c->live, st));
    }
    if (c->compact_copy) {
        StreamSwap on_exchange(st);
        for (int b = 0; b < c->nb; ++b)
            PGH_TRY(pgh_dist_compact_from_dense(c->graph, b, &v_full, c->dense_cold_base[b], &v_full, c->compact_base[b]));
        if (events && (c->one_gather || c->hot == 0)) PGH_HIP(hipEventRecord(c->ev_hot, st));
    }
    if (events) PGH_HIP(hipEventRecord(c->ev_cold, st));
    return 0;
}

// The finish kernel of a step in TWO launches when the exchange is what a step waits for (three queues, a cold image, 32 MB or more
// received per rank and iteration: 141 MB at configs[4]): first the items that hold rows whose slots of the next gather vector are
// exchanged (the referenced prefix of every block: the hottest rows), then the rest -- the all-gathers start behind the first launch and
// travel while the second runs.  Two launches cost: measured with one rank at scale 23, 86 -> 133 us for the two
// (each ends on its slowest workgroup, a workgroup holds 1-2 items of a phase instead of 3 of the step), so a small exchange keeps
// one.  PGH_DIST_FINISH_SPLIT=0 keeps one launch always, =2 forces two (tests).
bool finish_in_two(const pgh_comm_s* c, const pgh_graph_s* g) {
    const char* env = getenv("PGH_DIST_FINISH_SPLIT");
    const int mode = env != nullptr ? atoi(env) : 1;
    if (mode == 0 || !g->bsf.enabled || !g->bsf.pb.enabled) return false;
    // (both launches keep a partial sum per workgroup and per tail item: they must fit the partial buffers)
    if (2 * (g->bsf.pb.sched_groups + g->bsf.pb.tail_count) > kMaxPartials) return false;
    const int64_t received = c->exchange_bytes;
    return mode == 2 || (c->world > 1 && !c->one_gather && received >= (32LL << 20));
}


}  // namespace

extern "C" int pgh_comm_unique_id(uint8_t* id /* [PGH_COMM_ID_BYTES] */) {
    PGH_CHECK(id != nullptr, "pgh_comm_unique_id: null argument");
    PGH_TRY(ensure_init());
    PGH_TRY(load_rccl());
    ncclUniqueId uid;
    PGH_RCCL(g_rccl.GetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    return 0;
}

namespace {
// streams, events and scalars of a communicator; on failure the caller destroys the half-built object (pgh_comm_destroy frees
// whatever exists)
int comm_resources(pgh_comm_s* c, bool two_comms) {
    // one queue when asked for (PGH_DIST_SINGLE_STREAM=1: the conservative setting of a first run on new hardware) and when there is
    // nobody to exchange with: three queues exist to hide the exchange behind the step, and every cross-queue event costs
    // microseconds (one rank over RCCL at scale 23: 467 GTEPS on three queues, 496 on one; PGH_DIST_SINGLE_STREAM=0 forces three)
    const char* ss_env = getenv("PGH_DIST_SINGLE_STREAM");
    const bool single_stream = ss_env != nullptr ? atoi(ss_env) != 0 : c->world == 1;
    PGH_HIP(hipStreamCreateWithFlags(&c->main, hipStreamNonBlocking));
    c->own_streams = true;
    if (single_stream) {
        c->xs = c->ss = c->main;
    } else {
        PGH_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
        // one communicator: RCCL wants its operations in ONE order, so exchange and scalars share the side stream
        if (two_comms) PGH_HIP(hipStreamCreateWithFlags(&c->ss, hipStreamNonBlocking));
        else c->ss = c->xs;
    }
    for (hipEvent_t* ev : {&c->ev_fin, &c->ev_fin2, &c->ev_hot, &c->ev_cold, &c->ev_err, &c->ev_host})
        PGH_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    PGH_HIP(hipMalloc(&c->state, sizeof(double) * 8));
    PGH_HIP(hipHostMalloc(&c->state_host, sizeof(double) * 8, hipHostMallocDefault));
    PGH_HIP(hipMalloc(&c->aux, sizeof(LoopAux)));
    PGH_HIP(hipMalloc(&c->red, sizeof(double) * 4));
    PGH_HIP(hipMalloc(&c->agree, sizeof(int32_t) * 8));
    void* hp = nullptr;
    PGH_HIP(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
    c->progress_host = reinterpret_cast<volatile unsigned long long*>(hp);
    c->progress_host[0] = 0ULL;
    void* dp = nullptr;
    PGH_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    c->progress_dev = reinterpret_cast<unsigned long long*>(dp);
    c->one_gather = c->xs == c->main;
    return 0;
}

// host wait for the close of step `step` (or any raised flag): 0 = seen, *flag = the done flag that came with it
int wait_step(pgh_comm_s* c, int step, int* flag, const char* what) {
    const double limit = wait_limit_s();
    const auto start = std::chrono::steady_clock::now();
    long spins = 0;
    for (;;) {
        const unsigned long long w = __atomic_load_n(const_cast<const unsigned long long*>(c->progress_host), __ATOMIC_ACQUIRE);
        if ((int)(w >> 32) >= step || (unsigned int)w != 0u) {
            *flag = (int)(unsigned int)w;
            return 0;
        }
        if ((++spins & 4095) != 0) continue;
        // (a close normally lands within a step's time, a few hundred microseconds: the host spins through that -- a sleep here is
        // a late enqueue of the next finish kernel -- and only a wait that has lasted 20 ms yields the core between looks)
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        if (waited > 0.02) {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            const hipError_t e = hipStreamQuery(c->ss);                        // a device fault ends the wait at once
            if (e != hipSuccess && e != hipErrorNotReady) return fail(std::string("pgh_dist: waiting for ") + what + ": " + hipGetErrorString(e));
        }
        if (waited > limit)
            return fail(std::string("pgh_dist: ") + what + " did not complete within PGH_DIST_TIMEOUT_S -- a collective is stalled (a peer "
                        "gone, or communicators blocking each other: retry with a single communicator / stream)");
    }
}
}  // namespace

extern "C" int pgh_comm_create(const uint8_t* ids, int32_t num_ids, int32_t world, int32_t rank, pgh_comm_t* out) {
    PGH_CHECK(ids != nullptr && out != nullptr && (num_ids == 1 || num_ids == 2) && world >= 1 && rank >= 0 && rank < world,
              "pgh_comm_create: bad arguments");
    PGH_TRY(ensure_init());
    PGH_TRY(load_rccl());
    pgh_comm_s* c = new pgh_comm_s();
    c->world = world;
    c->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, ids, sizeof(uid));
    ncclResult_t rc = g_rccl.CommInitRank(&c->x, world, uid, rank);
    if (rc == ncclSuccess && num_ids == 2) {
        memcpy(&uid, ids + PGH_COMM_ID_BYTES, sizeof(uid));
        rc = g_rccl.CommInitRank(&c->s, world, uid, rank);
    } else {
        c->s = c->x;
    }
    if (rc != ncclSuccess) {
        const std::string msg = std::string("pgh_comm_create: ncclCommInitRank: ") + g_rccl.GetErrorString(rc);
        (void)pgh_comm_destroy(c);             // the first communicator, when only the second failed
        return fail(msg);
    }
    if (comm_resources(c, num_ids == 2) != 0) {
        const std::string msg = pgh_last_error();
        (void)pgh_comm_destroy(c);
        return fail(msg);
    }
    *out = c;
    return 0;
}

// A communicator whose collectives the HOST performs (MPI, gloo, ...): the engine still drives the loop, its streams and its
// events, and calls back for every exchange.  A callback receives device pointers and the HIP stream the exchange is ordered on; it
// must have completed the exchange, in stream order on that stream, when it returns (the simplest form: synchronise the stream,
// exchange through host memory, copy back).  Every rank issues the same callbacks in the same order.
extern "C" int pgh_comm_create_external(int32_t world, int32_t rank, pgh_allgather_fn all_gather, pgh_allreduce_fn all_reduce, void* user,
                                        pgh_comm_t* out) {
    PGH_CHECK(out != nullptr && all_gather != nullptr && all_reduce != nullptr && world >= 1 && rank >= 0 && rank < world,
              "pgh_comm_create_external: bad arguments");
    PGH_TRY(ensure_init());
    pgh_comm_s* c = new pgh_comm_s();
    c->world = world;
    c->rank = rank;
    c->ext_gather = all_gather;
    c->ext_reduce = all_reduce;
    c->ext_user = user;
    if (comm_resources(c, true) != 0) {
        const std::string msg = pgh_last_error();
        (void)pgh_comm_destroy(c);
        return fail(msg);
    }
    *out = c;
    return 0;
}

extern "C" int pgh_comm_set_alltoallv(pgh_comm_t c, pgh_alltoallv_fn all_to_all_v) {
    PGH_CHECK(c != nullptr && c->ext_gather != nullptr, "pgh_comm_set_alltoallv: a communicator of pgh_comm_create_external, please");
    c->ext_alltoallv = all_to_all_v;
    return 0;
}

extern "C" int pgh_comm_destroy(pgh_comm_t c) {
    if (c == nullptr) return 0;
    if (rt().initialised) (void)hipDeviceSynchronize();
    free_buffers(c);
    (void)hipFree(c->state);
    (void)hipFree(c->aux);
    (void)hipFree(c->red);
    (void)hipFree(c->agree);
    (void)hipHostFree(c->state_host);
    if (c->progress_host != nullptr) (void)hipHostFree(const_cast<unsigned long long*>(c->progress_host));
    for (hipEvent_t ev : {c->ev_fin, c->ev_fin2, c->ev_hot, c->ev_cold, c->ev_err, c->ev_host})
        if (ev) (void)hipEventDestroy(ev);
    if (g_rccl.handle != nullptr) {
        if (c->s != nullptr && c->s != c->x) (void)g_rccl.CommDestroy(c->s);
        if (c->x != nullptr) (void)g_rccl.CommDestroy(c->x);
    }
    if (c->ss != nullptr && c->ss != c->main && c->ss != c->xs) (void)hipStreamDestroy(c->ss);
    if (c->xs != nullptr && c->xs != c->main) (void)hipStreamDestroy(c->xs);
    if (c->main != nullptr) (void)hipStreamDestroy(c->main);
    delete c;
    return 0;
}

namespace {

// Whatever way a run ends: its timing events are destroyed, the isolated-row flag goes back to "process every row" IN ORDER with
// the engine's own stream (the release is enqueued on the communicator's compute queue, which is then drained), and after a
// failure nothing of the run is left in flight on the three queues -- unless a collective has stalled (bounded_wait gave up):
// draining would then wait for ever, so the queues are left alone and the caller is expected to end the process.
struct RunScope {
    pgh_comm_s* c;
    pgh_graph_t g;
    hipEvent_t  t_begin = nullptr, t_end = nullptr;
    bool        ok = false, stalled = false;
    int drain(hipStream_t st, const char* what) {
        if (hipEventRecord(c->ev_host, st) != hipSuccess) return 1;
        return bounded_wait(c->ev_host, what);
    }
    ~RunScope() {
        if (t_begin) (void)hipEventDestroy(t_begin);
        if (t_end) (void)hipEventDestroy(t_end);
        pb_set_residual(nullptr);
        pb_set_finish_phase(0, 0);
        if (stalled) return;
        if (!ok) {
            if (c->xs != c->main) (void)drain(c->xs, "the exchange queue after a failed run");
            if (c->ss != c->main && c->ss != c->xs) (void)drain(c->ss, "the scalar queue after a failed run");
        }
        (void)pgh_dist_release_isolated(g);          // on rt().stream == c->main (StreamSwap outlives this scope)
        (void)drain(c->main, "the compute queue at the end of the run");
    }
};

// row sums of M on this rank's rows (BsfFormat::deg_int of a slice): a rank holds its COLUMNS of M, so pgh_graph_s::degrees are
// partial sums over all ranks' ids -- summed over the ranks once per graph (collective), the slice kept with the graph
int ensure_slice_degrees(pgh_comm_s* c, pgh_graph_t g) {
    BsfFormat& f = g->bsf;
    if (f.deg_int != nullptr) return 0;
    const int64_t n_all = g->n_rows;
    float* all = nullptr;
    PGH_HIP(hipMalloc(&all, sizeof(float) * (size_t)(n_all > 0 ? n_all : 1)));
    int rc = 0;
    if (hipMemcpyAsync(all, g->degrees, sizeof(float) * (size_t)n_all, hipMemcpyDeviceToDevice, c->main) != hipSuccess) rc = fail("ensure_slice_degrees: copy failed");
    if (rc == 0 && c->world > 1) rc = comm_all_reduce(c, all, (size_t)n_all, ncclFloat32, ncclSum, c->s, c->main);
    float* mine = nullptr;
    if (rc == 0 && hipMalloc(&mine, sizeof(float) * (size_t)(g->n_cols > 0 ? g->n_cols : 1)) != hipSuccess) rc = fail("ensure_slice_degrees: out of device memory");
    if (rc == 0 && hipMemcpyAsync(mine, all + g->row_begin, sizeof(float) * (size_t)g->n_cols, hipMemcpyDeviceToDevice, c->main) != hipSuccess)
        rc = fail("ensure_slice_degrees: copy failed");
    if (rc == 0 && hipEventRecord(c->ev_host, c->main) != hipSuccess) rc = fail("ensure_slice_degrees: event");
    if (rc == 0) rc = bounded_wait(c->ev_host, "the all-reduce of the slice's degrees");
    if (rc == 0) {
        (void)hipFree(all);
        f.deg_int = mine;
        f.device_bytes += (int64_t)g->n_cols * 4;
        return 0;
    }
    // (a stalled all-reduce may still be writing `all`: it is leaked rather than freed under it)
    (void)hipFree(mine);
    return rc;
}

}  // namespace

extern "C" int pgh_dist_ppr_run(pgh_graph_t g, pgh_comm_t c, pgh_vec_t p_local, pgh_vec_t ranks_local, const pgh_dist_cfg* cfg,
                                pgh_dist_result* res) {
    PGH_CHECK(g && c && p_local && ranks_local && cfg && res, "pgh_dist_ppr_run: null argument");
    PGH_CHECK(p_local->n == g->n_cols && ranks_local->n == g->n_cols, "pgh_dist_ppr_run: vectors must have the slice's length");
    PGH_CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    PGH_TRY(ensure_init());
    Runtime& r = rt();
    memset(res, 0, sizeof(*res));
    PGH_HIP(hipStreamSynchronize(r.stream));               // the caller's operands are in place; from here on: the communicator's queues
    const int kind = cfg->err_kind;
    const bool absorbing = cfg->deg_local != nullptr && cfg->lam_local != nullptr;
    // the in-kernel residual (ResParams, pgh_kernels.h): PageRank with the L1 / Mabs rule on slices with a cold image -- on EVERY
    // rank (prepare_graph lets the ranks agree)
    bool fused = !absorbing && (kind == PGH_ERR_L1 || kind == PGH_ERR_MABS) && dist_can_fuse(g);
    PGH_TRY(prepare_graph(c, g, &fused));
    const int64_t n_local = c->n_local;
    PGH_CHECK(!absorbing || (cfg->deg_local->n == n_local && cfg->lam_local->n == n_local), "pgh_dist_ppr_run: deg / lam must have the slice's length");
    const int local_kind = kind == PGH_ERR_LINF ? PGH_ERR_LINF : PGH_ERR_L1;
    const ncclRedOp_t err_op = kind == PGH_ERR_LINF ? ncclMax : ncclSum;
    StreamSwap on_main(c->main);
    RunScope scope{c, g};
    pgh_vec_s v_p{c->p_norm, n_local, false}, v_xg_full{c->xg_full, c->n_xg, false}, v_xg_local{c->xg_local, n_local, false};
    pgh_vec_s v_y[2] = {{c->y[0], n_local, false}, {c->y[1], n_local, false}};
    int rc = 0;
#define PGH_RUN(expr)                                    \
    do {                                                 \
        if ((rc = (expr)) != 0) return rc;               \
    } while (0)
#define PGH_RUN_HIP(expr)                                                                                       \
    do {                                                                                                        \
        const hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (pgh_dist_ppr_run)"); \
    } while (0)
    auto wait_host = [&](hipEvent_t ev, const char* what) -> int {
        const int w = bounded_wait(ev, what);
        if (w != 0) scope.stalled = true;
        return w;
    };
    if (fused) PGH_RUN(ensure_slice_degrees(c, g));

    // ---- prologue of GraphFilter.rank (abstract_filters.py:52-56): global L1 norm, x0 = p / norm
    double local_abs = 0.0;
    PGH_RUN(pgh_reduce(PGH_ABSSUM, p_local, &local_abs));
    PGH_RUN_HIP(hipMemcpyAsync(c->state, &local_abs, sizeof(double), hipMemcpyHostToDevice, c->main));
    PGH_RUN(comm_all_reduce(c, c->state, 1, ncclFloat64, ncclSum, c->s, c->main));
    PGH_RUN_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(double), hipMemcpyDeviceToHost, c->main));
    PGH_RUN_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_RUN(wait_host(c->ev_host, "the all-reduce of the personalization's norm"));
    const double norm = c->state_host[0];
    if (norm == 0.0) {
        PGH_RUN(pgh_vec_copy(ranks_local, p_local));
        res->iterations = 0;
        scope.ok = true;
        return 0;
    }
    k_div_into<<<grid_of(n_local), 256, 0, c->main>>>(p_local->data, c->p_norm, n_local, (float)norm);     // the backend's f32 `p / norm`
    int cur = 0;
    PGH_RUN_HIP(hipMemsetAsync(c->y[1], 0, sizeof(float) * (size_t)n_local, c->main));      // rows a run passes over hold zeros in both iterates
    PGH_RUN(pgh_vec_copy(&v_y[0], &v_p));
    if (!cfg->every_row) PGH_RUN(pgh_dist_watch_isolated(g, &v_p, &v_y[0]));
    PGH_RUN(pgh_dist_prescale(g, &v_y[0], &v_xg_local));
    PGH_RUN(exchange_slices(c, c->main, false));
    c->progress_host[0] = 0ULL;                // (the previous run has drained: nobody writes the word any more)
    PGH_RUN(pgh_dist_state_init(c->state));
    if (fused) PGH_RUN(dist_aux_init(c->aux));
    for (hipEvent_t ev : {c->ev_hot, c->ev_cold, c->ev_err}) PGH_RUN_HIP(hipEventRecord(ev, c->main));
    PGH_RUN_HIP(hipEventCreate(&scope.t_begin));
    PGH_RUN_HIP(hipEventCreate(&scope.t_end));
    PGH_RUN_HIP(hipEventRecord(scope.t_begin, c->main));

    auto stages = [&]() -> int {
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_hot, 0));
        PGH_TRY(pgh_dist_partial_stage(g, &v_xg_full, c->state, 1));
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_cold, 0));
        PGH_TRY(pgh_dist_partial_stage(g, &v_xg_full, c->state, 2));
        return 0;
    };
    // the scalars of a step the plain way, on the scalar queue (the caller has swapped rt().stream to it): all-reduce of sum(y)
    // -> lazy quotient -> residual -> all-reduce -> stopping rule -> 64 bytes of state to the host
    auto scalars_plain = [&](bool check, bool with_sum) -> int {
        if (with_sum) {
            PGH_TRY(comm_all_reduce(c, c->state + 2, 1, ncclFloat64, ncclSum, c->s, c->ss));
            PGH_TRY(pgh_dist_close_sum(c->state, cfg->use_quotient));
        }
        if (check) {
            PGH_TRY(pgh_dist_residual(local_kind, &v_y[cur], &v_y[1 - cur], c->state));
            PGH_TRY(comm_all_reduce(c, c->state + 1, 1, ncclFloat64, err_op, c->s, c->ss));
            PGH_TRY(dist_close_err(c->state, kind, cfg->tol, cfg->n_global, c->progress_dev));
        }
        return 0;
    };
    auto wait_close = [&](int step, int* flag, const char* what) -> int {
        const int w = wait_step(c, step, flag, what);
        if (w != 0) scope.stalled = true;
        return w;
    };
    const int max_iters = cfg->max_iters;
    int it = 1, spmv = 0;                      // `it` = ConvergenceManager.iteration of the pending has_converged call
    int fused_partials = 0;
    const bool two_launches = finish_in_two(c, g);
    res->flags |= two_launches ? 4 : 0;
    res->flags |= c->lists ? 8 : (c->compact_copy ? 16 : 0);       // how the cold parts of the gather vector travel
    bool pending = false, staged = false, converged = false;
    while (it < max_iters) {                   // convergence.py:86
        const int nxt = 1 - cur;
        if (!staged) PGH_RUN(stages());
        staged = false;
        if (pending) {
            // the check that followed the previous step: its flag has travelled while the stages above were enqueued
            int flag = 0;
            PGH_RUN(wait_close(spmv, &flag, "the scalar all-reduce of the previous step"));
            pending = false;
            if (flag == 2) {
                // the in-kernel residual could not vouch for its verdict (the quotient's prediction missed by more than the
                // distance to the tolerance, or a value was negative / not finite): the step is complete but for its residual --
                // the separate kernel evaluates it, and the run goes on without the fusion.  Every rank sees the same flag.
                fused = false;
                res->flags |= 1;
                c->progress_host[0] = 0ULL;    // (the queues are idle behind the pause: every later kernel saw the flag and left)
                {
                    StreamSwap on_scalars(c->ss);
                    PGH_RUN(dist_resume(c->state, nullptr));
                    PGH_RUN(pgh_dist_close_sum(c->state, cfg->use_quotient));
                    PGH_RUN(scalars_plain(true, false));
                }
                PGH_RUN_HIP(hipEventRecord(c->ev_err, c->ss));
                PGH_RUN(wait_close(spmv, &flag, "the re-evaluated residual of a paused step"));
                if (flag == 0) {
                    PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));
                    PGH_RUN(stages());         // the speculated stages may have seen the pause and done nothing
                }
            }
            if (flag != 0) {
                converged = true;              // whatever the stages above computed is never folded into an iterate
                break;
            }
        }
        PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));      // quotient and done flag of the previous step; its residual has read y[nxt]
        auto finish = [&]() -> int {
            if (fused) return dist_combine_fused(g, c->p_norm, cfg->alpha, c->y[nxt], c->xg_local, c->y[cur], g->bsf.deg_int, c->state, c->aux, spmv + 1, &fused_partials);
            return absorbing ? pgh_dist_combine_absorb(g, &v_p, cfg->deg_local, cfg->lam_local, &v_y[nxt], &v_xg_local, c->state)
                             : pgh_dist_combine(g, &v_p, cfg->alpha, &v_y[nxt], &v_xg_local, c->state);
        };
        if (two_launches) {
            pb_set_finish_phase(1, c->live);
            PGH_RUN(finish());
            PGH_RUN_HIP(hipEventRecord(c->ev_fin, c->main));          // every exchanged slot is written: the exchange may start
            pb_set_finish_phase(2, c->live);
            PGH_RUN(finish());
        } else {
            PGH_RUN(finish());
            PGH_RUN_HIP(hipEventRecord(c->ev_fin, c->main));
        }
        PGH_RUN_HIP(hipEventRecord(c->ev_fin2, c->main));             // ... and the step's sums are complete: the scalars may start
        // ---- X: the next gather vector over xGMI
        PGH_RUN_HIP(hipStreamWaitEvent(c->xs, c->ev_fin, 0));
        PGH_RUN(exchange_slices(c, c->xs, true));
        cur = nxt;
        ++spmv;
        ++it;
        const bool check = it < max_iters && kind != PGH_ERR_ITERS && it % cfg->end_modulo == 0;
        // ---- S: the scalars of the step
        PGH_RUN_HIP(hipStreamWaitEvent(c->ss, c->ev_fin2, 0));
        {
            StreamSwap on_scalars(c->ss);
            if (fused) {
                // ONE all-reduce: {S, T, D, R'}; the first step of a run has no prediction and takes the separate residual
                PGH_RUN(dist_fold_fused(c->state, c->red, fused_partials));
                PGH_RUN(comm_all_reduce(c, c->red, 4, ncclFloat64, ncclSum, c->s, c->ss));
                PGH_RUN(dist_close_fused(c->state, c->aux, c->red, spmv, check ? 1 : 0, kind, cfg->tol, cfg->n_global, cfg->use_quotient,
                                         cfg->alpha, 1.0 - cfg->alpha, c->progress_dev));
                if (spmv == 1) PGH_RUN(scalars_plain(check, false));
            } else {
                PGH_RUN(scalars_plain(check, true));
            }
        }
        PGH_RUN_HIP(hipEventRecord(c->ev_err, c->ss));
        if (it >= max_iters) break;
        if (check) {
            pending = true;
            PGH_RUN(stages());                 // speculate: the next step's first two stages need the exchange only
            staged = true;
        }
    }
    PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));
    PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_cold, 0));
    if (pending && !converged) {
        int flag = 0;
        PGH_RUN(wait_close(spmv, &flag, "the scalar all-reduce of the last step"));
        if (flag == 2) {                       // (a pause at the very last check: same re-evaluation as inside the loop)
            res->flags |= 1;
            c->progress_host[0] = 0ULL;
            {
                StreamSwap on_scalars(c->ss);
                PGH_RUN(dist_resume(c->state, nullptr));
                PGH_RUN(pgh_dist_close_sum(c->state, cfg->use_quotient));
                PGH_RUN(scalars_plain(true, false));
            }
            PGH_RUN_HIP(hipEventRecord(c->ev_err, c->ss));
            PGH_RUN(wait_close(spmv, &flag, "the re-evaluated residual of a paused step"));
            PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));
        }
        converged = flag != 0;
    }
    PGH_RUN_HIP(hipEventRecord(scope.t_end, c->main));
    PGH_RUN_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(double) * 8, hipMemcpyDeviceToHost, c->main));
    PGH_RUN_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_RUN(wait_host(c->ev_host, "the loop state of the last step"));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, scope.t_begin, scope.t_end);
    const int steps = reinterpret_cast<const int*>(c->state_host)[7];
    PGH_CHECK(steps == spmv, "pgh_dist_ppr_run: the device ran a different number of steps than the host enqueued");
    const double scale = c->state_host[0];
    const double factor = scale * (cfg->preserve_norm ? norm : 1.0);           // abstract_filters.py:63-64
    k_scale_into<<<grid_of(n_local), 256, 0, c->main>>>(c->y[cur], ranks_local->data, n_local, factor);
    PGH_RUN_HIP(hipGetLastError());
    res->iterations = it;
    res->spmv_count = spmv;
    res->converged = converged ? 1 : 0;
    res->last_error = c->state_host[6];
    res->loop_ms = (double)ms;
    res->exchange_bytes = c->exchange_bytes;
    res->gather_slots = (int64_t)c->nb * c->live;
    res->column_blocks = c->nb;
    res->split_regions = c->hot > 0 ? 1 : 0;
    if (fused || (res->flags & 1)) res->flags |= 2;        // bit 1: the run used the in-kernel residual (bit 0: and paused it once)
    scope.ok = true;
    return 0;
#undef PGH_RUN
#undef PGH_RUN_HIP
}

// ClosedFormGraphFilter (abstract_filters.py:152-270, taylor form) on a partition, the whole run behind one call like
// pgh_dist_ppr_run: result = sum_k coeffs[k - 1] (M^T)^(k-1) p with ConvergenceManager's rule on the change of the result
// (convergence.py:77-101).  Same three queues: C = block partial sums -> phase A -> finish with the accumulate epilogue
// (pgh_dist_combine_poly), X = the split all-gather of the term's gather slice, S = ONE all-reduce of the change per term and the
// stopping rule on the device; the host reads the flag of term k after it has enqueued the first two stages of term k + 1.
// HeatKernel / PageRankClosed pass their coefficient schedules (adhoc.py:83-84,113-116); coefficients past num_coeffs are 0.
extern "C" int pgh_dist_poly_run(pgh_graph_t g, pgh_comm_t c, pgh_vec_t p_local, const double* coeffs, int32_t num_coeffs,
                                 pgh_vec_t result_local, const pgh_dist_cfg* cfg, pgh_dist_result* res) {
    PGH_CHECK(g && c && p_local && result_local && cfg && res && (coeffs != nullptr || num_coeffs == 0), "pgh_dist_poly_run: null argument");
    PGH_CHECK(p_local->n == g->n_cols && result_local->n == g->n_cols && p_local->data != result_local->data,
              "pgh_dist_poly_run: vectors must have the slice's length");
    PGH_CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    PGH_TRY(ensure_init());
    Runtime& r = rt();
    memset(res, 0, sizeof(*res));
    PGH_HIP(hipStreamSynchronize(r.stream));
    PGH_TRY(prepare_graph(c, g, nullptr));
    const int64_t n_local = c->n_local;
    const int kind = cfg->err_kind;
    const int linf = kind == PGH_ERR_LINF ? 1 : 0;
    const ncclRedOp_t err_op = linf ? ncclMax : ncclSum;
    auto coeff = [&](int it) -> double { return (it >= 1 && it <= num_coeffs) ? coeffs[it - 1] : 0.0; };
    StreamSwap on_main(c->main);
    RunScope scope{c, g};
    pgh_vec_s v_xg_full{c->xg_full, c->n_xg, false}, v_xg_local{c->xg_local, n_local, false};
    pgh_vec_s v_y[2] = {{c->y[0], n_local, false}, {c->y[1], n_local, false}};
    int rc = 0;
#define PGH_RUN(expr)                                    \
    do {                                                 \
        if ((rc = (expr)) != 0) return rc;               \
    } while (0)
#define PGH_RUN_HIP(expr)                                                                                       \
    do {                                                                                                        \
        const hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (pgh_dist_poly_run)"); \
    } while (0)
    auto wait_host = [&](hipEvent_t ev, const char* what) -> int {
        const int w = bounded_wait(ev, what);
        if (w != 0) scope.stalled = true;
        return w;
    };
    // ---- prologue: global L1 norm (and, for the max rule, the global max |p|) in one round trip each
    double local[2] = {0.0, 0.0};
    PGH_RUN(pgh_reduce(PGH_ABSSUM, p_local, &local[0]));
    if (linf && n_local > 0) {
        double hi = 0.0, lo = 0.0;
        PGH_RUN(pgh_reduce(PGH_MAX, p_local, &hi));
        PGH_RUN(pgh_reduce(PGH_MIN, p_local, &lo));
        local[1] = fmax(fabs(hi), fabs(lo));
    }
    PGH_RUN_HIP(hipMemcpyAsync(c->state, local, sizeof(local), hipMemcpyHostToDevice, c->main));
    PGH_RUN(comm_all_reduce(c, c->state, 1, ncclFloat64, ncclSum, c->s, c->main));
    if (linf) PGH_RUN(comm_all_reduce(c, c->state + 1, 1, ncclFloat64, ncclMax, c->s, c->main));
    PGH_RUN_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(local), hipMemcpyDeviceToHost, c->main));
    PGH_RUN_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_RUN(wait_host(c->ev_host, "the all-reduce of the personalization's norm"));
    const double norm = c->state_host[0];
    if (norm == 0.0) {
        PGH_RUN(pgh_vec_copy(result_local, p_local));
        res->iterations = 0;
        scope.ok = true;
        return 0;
    }
    const float p_absmax = (float)c->state_host[1] / (float)norm;
    const double c1 = coeff(1);
    k_div_into<<<grid_of(n_local), 256, 0, c->main>>>(p_local->data, c->y[0], n_local, (float)norm);        // term_1 = p / norm
    PGH_RUN(pgh_ewise_vs(PGH_MUL, &v_y[0], c1, 0, result_local));                                           // result_1 = c_1 term_1
    PGH_RUN(pgh_dist_prescale(g, &v_y[0], &v_xg_local));
    PGH_RUN(exchange_slices(c, c->main, false));
    c->progress_host[0] = 0ULL;
    PGH_RUN(pgh_dist_state_init(c->state));
    for (hipEvent_t ev : {c->ev_hot, c->ev_cold, c->ev_err}) PGH_RUN_HIP(hipEventRecord(ev, c->main));
    PGH_RUN_HIP(hipEventCreate(&scope.t_begin));
    PGH_RUN_HIP(hipEventCreate(&scope.t_end));
    PGH_RUN_HIP(hipEventRecord(scope.t_begin, c->main));
    // the change of the first term decides only whether the loop stops at iteration 2: |result_1 - 0| = |c_1| sum |p / norm| = |c_1|
    double delta = fabs(c1) * (linf ? (double)p_absmax : 1.0);
    if (kind == PGH_ERR_MABS) delta /= (double)cfg->n_global;
    auto stages = [&]() -> int {
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_hot, 0));
        PGH_TRY(pgh_dist_partial_stage(g, &v_xg_full, c->state, 1));
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_cold, 0));
        PGH_TRY(pgh_dist_partial_stage(g, &v_xg_full, c->state, 2));
        return 0;
    };
    const int max_iters = cfg->max_iters;
    const bool two_launches = finish_in_two(c, g);
    res->flags |= two_launches ? 4 : 0;
    res->flags |= c->lists ? 8 : (c->compact_copy ? 16 : 0);       // how the cold parts of the gather vector travel
    int it = 2, spmv = 0, cur = 0;             // `it` = the iteration has_converged is asked about
    bool converged = false, pending = false, staged = false;
    if (it < max_iters && kind != PGH_ERR_ITERS && it % cfg->end_modulo == 0 && delta <= cfg->tol) converged = true;
    while (!converged && it < max_iters) {
        const int nxt = 1 - cur;
        if (!staged) PGH_RUN(stages());
        staged = false;
        if (pending) {
            int flag = 0;
            const int w = wait_step(c, spmv, &flag, "the all-reduce of the previous term's change");
            if (w != 0) {
                scope.stalled = true;
                return w;
            }
            pending = false;
            if (flag != 0) {
                converged = true;
                break;
            }
        }
        PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));      // the done flag of the previous term; its all-reduce has read state[1]
        if (two_launches) {
            pb_set_finish_phase(1, c->live);
            PGH_RUN(pgh_dist_combine_poly(g, &v_y[cur], &v_y[nxt], 1.0, 0.0, result_local, coeff(it), linf, &v_xg_local, c->state));
            PGH_RUN_HIP(hipEventRecord(c->ev_fin, c->main));
            pb_set_finish_phase(2, c->live);
        }
        PGH_RUN(pgh_dist_combine_poly(g, &v_y[cur], &v_y[nxt], 1.0, 0.0, result_local, coeff(it), linf, &v_xg_local, c->state));
        if (!two_launches) PGH_RUN_HIP(hipEventRecord(c->ev_fin, c->main));
        PGH_RUN_HIP(hipEventRecord(c->ev_fin2, c->main));
        PGH_RUN_HIP(hipStreamWaitEvent(c->xs, c->ev_fin, 0));
        PGH_RUN(exchange_slices(c, c->xs, true));
        cur = nxt;
        ++spmv;
        ++it;
        const bool check = it < max_iters && kind != PGH_ERR_ITERS && it % cfg->end_modulo == 0;
        PGH_RUN_HIP(hipStreamWaitEvent(c->ss, c->ev_fin2, 0));
        {
            StreamSwap on_scalars(c->ss);
            PGH_RUN(pgh_dist_close_sum(c->state, 0));                // counts the step
            if (check) {
                PGH_RUN(comm_all_reduce(c, c->state + 1, 1, ncclFloat64, err_op, c->s, c->ss));
                PGH_RUN(dist_close_err(c->state, kind, cfg->tol, cfg->n_global, c->progress_dev));
            }
        }
        PGH_RUN_HIP(hipEventRecord(c->ev_err, c->ss));
        if (it >= max_iters) break;
        if (check) {
            pending = true;
            PGH_RUN(stages());
            staged = true;
        }
    }
    PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));
    PGH_RUN_HIP(hipStreamWaitEvent(c->main, c->ev_cold, 0));
    if (pending && !converged) {
        int flag = 0;
        const int w = wait_step(c, spmv, &flag, "the all-reduce of the last term's change");
        if (w != 0) {
            scope.stalled = true;
            return w;
        }
        converged = flag != 0;
    }
    PGH_RUN_HIP(hipEventRecord(scope.t_end, c->main));
    PGH_RUN_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(double) * 8, hipMemcpyDeviceToHost, c->main));
    PGH_RUN_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_RUN(wait_host(c->ev_host, "the loop state of the last term"));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, scope.t_begin, scope.t_end);
    const int steps = reinterpret_cast<const int*>(c->state_host)[7];
    PGH_CHECK(steps == spmv, "pgh_dist_poly_run: the device ran a different number of steps than the host enqueued");
    if (cfg->preserve_norm && norm != 1.0) {
        pgh_vec_s v_res{result_local->data, n_local, false};
        PGH_RUN(pgh_ewise_vs(PGH_MUL, &v_res, norm, 0, &v_res));
    }
    PGH_RUN_HIP(hipGetLastError());
    res->iterations = it;
    res->spmv_count = spmv;
    res->converged = converged ? 1 : 0;
    res->last_error = spmv > 0 ? c->state_host[6] : delta;
    res->loop_ms = (double)ms;
    res->exchange_bytes = c->exchange_bytes;
    res->gather_slots = (int64_t)c->nb * c->live;
    res->column_blocks = c->nb;
    res->split_regions = c->hot > 0 ? 1 : 0;
    scope.ok = true;
    return 0;
#undef PGH_RUN
#undef PGH_RUN_HIP
}

// upper bound of every host wait on a collective from now on (seconds; <= 0: back to PGH_DIST_TIMEOUT_S / 600 s)
extern "C" int pgh_dist_set_timeout(double seconds) {
    wait_limit_s() = seconds > 0.0 ? seconds : default_wait_limit_s();
    return 0;
}

PGH_WARM_KERNEL(k_scale_into)
