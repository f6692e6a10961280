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
    hipEvent_t  ev_fin = nullptr, ev_hot = nullptr, ev_cold = nullptr, ev_err = nullptr, ev_host = nullptr;
    // buffers of the graph the communicator last ran on
    pgh_graph_t graph = nullptr;
    int         nb = 0, bpr = 0, live = 0, hot = 0;
    int64_t     blk = 0, n_xg = 0, n_local = 0, buf_local = 0;
    float*      xg_full = nullptr;
    float*      xg_local = nullptr;
    float*      y[2] = {nullptr, nullptr};
    float*      p_norm = nullptr;
    double*     state = nullptr;         // [8] device (pgh_dist_* layout)
    double*     state_host = nullptr;    // [8] pinned
    int32_t*    agree = nullptr;         // [2] device words of the layout negotiation
    // collectives supplied by the host instead of RCCL (pgh_comm_create_external): MPI, gloo, a test harness ...
    pgh_allgather_fn ext_gather = nullptr;
    pgh_allreduce_fn ext_reduce = nullptr;
    void*            ext_user = nullptr;
};

namespace {

// one all-gather / all-reduce of the run: RCCL on `st`, or the host's callback (which completes the exchange in stream order on `st`
// before it returns: dtype 0 = f32, 1 = f64, 2 = i32; op 0 = sum, 1 = max)
int comm_all_gather(pgh_comm_s* c, const void* send, void* recv, size_t count, ncclComm_t comm, hipStream_t st) {
    if (c->ext_gather != nullptr) {
        PGH_CHECK(c->ext_gather(c->ext_user, send, recv, (int64_t)count, 0, (void*)st) == 0, "pgh_dist_ppr_run: the host's all-gather callback failed");
        return 0;
    }
    PGH_RCCL(g_rccl.AllGather(send, recv, count, ncclFloat32, comm, st));
    return 0;
}
int comm_all_reduce(pgh_comm_s* c, void* buf, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (c->ext_reduce != nullptr) {
        const int32_t dtype = dt == ncclFloat32 ? 0 : (dt == ncclFloat64 ? 1 : 2);
        PGH_CHECK(c->ext_reduce(c->ext_user, buf, (int64_t)count, dtype, op == ncclMax ? 1 : 0, (void*)st) == 0,
                  "pgh_dist_ppr_run: the host's all-reduce callback failed");
        return 0;
    }
    PGH_RCCL(g_rccl.AllReduce(buf, buf, count, dt, op, comm, st));
    return 0;
}

void free_buffers(pgh_comm_s* c) {
    (void)hipFree(c->xg_full);
    (void)hipFree(c->xg_local);
    (void)hipFree(c->y[0]);
    (void)hipFree(c->y[1]);
    (void)hipFree(c->p_norm);
    c->xg_full = c->xg_local = c->y[0] = c->y[1] = c->p_norm = nullptr;
    c->graph = nullptr;
}

// host wait with a deadline: 0 = the event completed
int bounded_wait(hipEvent_t ev, const char* what) {
    static const double limit = getenv("PGH_DIST_TIMEOUT_S") != nullptr ? atof(getenv("PGH_DIST_TIMEOUT_S")) : 600.0;
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

// gather-vector layout and buffers for graph g (collective: every rank calls it with its slice of the same partition).  The
// layout is negotiated on every run (one 8-byte all-reduce: a handle is no proof that the graph behind it is the one of the
// last run); the buffers are kept while their sizes fit.
int prepare_graph(pgh_comm_s* c, pgh_graph_t g) {
    Runtime& r = rt();
    int32_t nb = 0, live8[8] = {0}, hot_slots = 0;
    int64_t blk = 0;
    PGH_TRY(pgh_graph_gather_layout(g, &nb, &blk, live8));
    PGH_TRY(pgh_graph_hot_prefix(g, &hot_slots));
    PGH_CHECK(nb % c->world == 0, "pgh_dist_ppr_run: the column blocks of the slice do not divide among the ranks");
    int32_t top = 0;
    for (int b = 0; b < nb; ++b) top = live8[b] > top ? live8[b] : top;
    // every rank must lay the gather vector out the same way: live = max over ranks, hot prefix = min over ranks
    int32_t h_agree[2] = {top, -hot_slots};
    PGH_HIP(hipMemcpyAsync(c->agree, h_agree, sizeof(h_agree), hipMemcpyHostToDevice, c->main));
    PGH_TRY(comm_all_reduce(c, c->agree, 2, ncclInt32, ncclMax, c->s, c->main));
    PGH_HIP(hipMemcpyAsync(h_agree, c->agree, sizeof(h_agree), hipMemcpyDeviceToHost, c->main));
    PGH_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_TRY(bounded_wait(c->ev_host, "the layout negotiation"));
    const int64_t live = std::min<int64_t>(blk, ((int64_t)h_agree[0] + 63) / 64 * 64);
    const int64_t hot_all = -h_agree[1];
    // split regions only when the exchange's hot prefix is exactly what the block partial sums read
    const bool split = hot_all > 0 && hot_all < live && hot_all % 64 == 0 && hot_all == hot_slots &&
                       !(getenv("PGH_DIST_SPLIT") != nullptr && atoi(getenv("PGH_DIST_SPLIT")) == 0);
    c->nb = nb;
    c->blk = blk;
    c->bpr = nb / c->world;
    c->live = (int)live;
    c->hot = split ? (int)hot_all : 0;
    c->n_local = g->n_cols;
    PGH_CHECK(blk * nb == g->n_rows && c->n_local == (int64_t)c->bpr * blk, "pgh_dist_ppr_run: the slice does not match the block layout");
    int64_t hot_bases[8] = {0}, cold_bases[8] = {0};
    for (int b = 0; b < nb; ++b) {
        const int rk = b / c->bpr, j = b % c->bpr;
        if (split) {
            hot_bases[b] = ((int64_t)j * c->world + rk) * c->hot;
            cold_bases[b] = (int64_t)nb * c->hot + ((int64_t)j * c->world + rk) * (live - c->hot);
        } else {
            hot_bases[b] = ((int64_t)j * c->world + rk) * live;
        }
    }
    if (split) PGH_TRY(pgh_graph_set_gather_bases_split(g, hot_bases, cold_bases));
    else PGH_TRY(pgh_graph_set_gather_bases(g, hot_bases));
    const int64_t n_xg = (int64_t)nb * live + 32768;          // + the hot cache's read-ahead past a short block
    if (c->graph != nullptr && c->n_xg == n_xg && c->buf_local == c->n_local) {
        c->graph = g;
        return 0;
    }
    free_buffers(c);
    c->n_xg = n_xg;
    c->buf_local = c->n_local;
    PGH_HIP(hipMalloc(&c->xg_full, sizeof(float) * (size_t)c->n_xg));
    PGH_HIP(hipMemsetAsync(c->xg_full, 0, sizeof(float) * (size_t)c->n_xg, r.stream));
    PGH_HIP(hipMalloc(&c->xg_local, sizeof(float) * (size_t)c->n_local));
    PGH_HIP(hipMalloc(&c->y[0], sizeof(float) * (size_t)c->n_local));
    PGH_HIP(hipMalloc(&c->y[1], sizeof(float) * (size_t)c->n_local));
    PGH_HIP(hipMalloc(&c->p_norm, sizeof(float) * (size_t)c->n_local));
    PGH_HIP(hipStreamSynchronize(r.stream));
    c->graph = g;
    return 0;
}

// all-gather of slots [lo, hi) of every block of this rank's slice into `region` ([j][rank][hi - lo]) on stream `st`
int gather_part(pgh_comm_s* c, int64_t region, int lo, int hi, hipStream_t st) {
    if (hi <= lo) return 0;
    const int64_t len = hi - lo;
    for (int j = 0; j < c->bpr; ++j)
        PGH_TRY(comm_all_gather(c, c->xg_local + (int64_t)j * c->blk + lo, c->xg_full + region + (int64_t)j * c->world * len, (size_t)len, c->x, st));
    return 0;
}

struct StreamSwap {          // the engine launches on rt().stream: point it at one of the communicator's queues for a scope
    hipStream_t saved;
    explicit StreamSwap(hipStream_t s) : saved(rt().stream) { rt().stream = s; }
    ~StreamSwap() { rt().stream = saved; }
};

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
        delete c;
        return fail(msg);
    }
    const bool single_stream = getenv("PGH_DIST_SINGLE_STREAM") != nullptr && atoi(getenv("PGH_DIST_SINGLE_STREAM")) != 0;
    PGH_HIP(hipStreamCreateWithFlags(&c->main, hipStreamNonBlocking));
    if (single_stream) {
        c->xs = c->ss = c->main;
    } else {
        PGH_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
        // one communicator: RCCL wants its operations in ONE order, so exchange and scalars share the side stream
        if (num_ids == 2) PGH_HIP(hipStreamCreateWithFlags(&c->ss, hipStreamNonBlocking));
        else c->ss = c->xs;
    }
    c->own_streams = true;
    for (hipEvent_t* ev : {&c->ev_fin, &c->ev_hot, &c->ev_cold, &c->ev_err, &c->ev_host})
        PGH_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    PGH_HIP(hipMalloc(&c->state, sizeof(double) * 8));
    PGH_HIP(hipHostMalloc(&c->state_host, sizeof(double) * 8, hipHostMallocDefault));
    PGH_HIP(hipMalloc(&c->agree, sizeof(int32_t) * 2));
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
    const bool single_stream = getenv("PGH_DIST_SINGLE_STREAM") != nullptr && atoi(getenv("PGH_DIST_SINGLE_STREAM")) != 0;
    PGH_HIP(hipStreamCreateWithFlags(&c->main, hipStreamNonBlocking));
    if (single_stream) {
        c->xs = c->ss = c->main;
    } else {
        PGH_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
        PGH_HIP(hipStreamCreateWithFlags(&c->ss, hipStreamNonBlocking));
    }
    c->own_streams = true;
    for (hipEvent_t* ev : {&c->ev_fin, &c->ev_hot, &c->ev_cold, &c->ev_err, &c->ev_host})
        PGH_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    PGH_HIP(hipMalloc(&c->state, sizeof(double) * 8));
    PGH_HIP(hipHostMalloc(&c->state_host, sizeof(double) * 8, hipHostMallocDefault));
    PGH_HIP(hipMalloc(&c->agree, sizeof(int32_t) * 2));
    *out = c;
    return 0;
}

extern "C" int pgh_comm_destroy(pgh_comm_t c) {
    if (c == nullptr) return 0;
    if (rt().initialised) (void)hipDeviceSynchronize();
    free_buffers(c);
    (void)hipFree(c->state);
    (void)hipFree(c->agree);
    (void)hipHostFree(c->state_host);
    for (hipEvent_t ev : {c->ev_fin, c->ev_hot, c->ev_cold, c->ev_err, c->ev_host})
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

extern "C" int pgh_dist_ppr_run(pgh_graph_t g, pgh_comm_t c, pgh_vec_t p_local, pgh_vec_t ranks_local, const pgh_dist_cfg* cfg,
                                pgh_dist_result* res) {
    PGH_CHECK(g && c && p_local && ranks_local && cfg && res, "pgh_dist_ppr_run: null argument");
    PGH_CHECK(p_local->n == g->n_cols && ranks_local->n == g->n_cols, "pgh_dist_ppr_run: vectors must have the slice's length");
    PGH_CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    PGH_TRY(ensure_init());
    Runtime& r = rt();
    memset(res, 0, sizeof(*res));
    PGH_HIP(hipStreamSynchronize(r.stream));               // the caller's operands are in place; from here on: the communicator's queues
    PGH_TRY(prepare_graph(c, g));
    const int64_t n_local = c->n_local;
    const int kind = cfg->err_kind;
    const int local_kind = kind == PGH_ERR_LINF ? PGH_ERR_LINF : PGH_ERR_L1;
    const ncclRedOp_t err_op = kind == PGH_ERR_LINF ? ncclMax : ncclSum;
    StreamSwap on_main(c->main);
    struct IsoGuard {
        pgh_graph_t g;
        ~IsoGuard() { (void)pgh_dist_release_isolated(g); }
    } iso_guard{g};
    pgh_vec_s v_p{c->p_norm, n_local, false}, v_xg_full{c->xg_full, c->n_xg, false}, v_xg_local{c->xg_local, n_local, false};
    pgh_vec_s v_y[2] = {{c->y[0], n_local, false}, {c->y[1], n_local, false}};

    // ---- prologue of GraphFilter.rank (abstract_filters.py:52-56): global L1 norm, x0 = p / norm
    double local_abs = 0.0;
    PGH_TRY(pgh_reduce(PGH_ABSSUM, p_local, &local_abs));
    PGH_HIP(hipMemcpyAsync(c->state, &local_abs, sizeof(double), hipMemcpyHostToDevice, c->main));
    PGH_TRY(comm_all_reduce(c, c->state, 1, ncclFloat64, ncclSum, c->s, c->main));
    PGH_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(double), hipMemcpyDeviceToHost, c->main));
    PGH_HIP(hipEventRecord(c->ev_host, c->main));
    PGH_TRY(bounded_wait(c->ev_host, "the all-reduce of the personalization's norm"));
    const double norm = c->state_host[0];
    if (norm == 0.0) {
        PGH_TRY(pgh_vec_copy(ranks_local, p_local));
        PGH_HIP(hipStreamSynchronize(c->main));
        res->iterations = 0;
        return 0;
    }
    k_div_into<<<grid_of(n_local), 256, 0, c->main>>>(p_local->data, c->p_norm, n_local, (float)norm);     // the backend's f32 `p / norm`
    int cur = 0;
    PGH_HIP(hipMemsetAsync(c->y[1], 0, sizeof(float) * (size_t)n_local, c->main));      // rows a run passes over hold zeros in both iterates
    PGH_TRY(pgh_vec_copy(&v_y[0], &v_p));
    const bool absorbing = cfg->deg_local != nullptr && cfg->lam_local != nullptr;
    PGH_CHECK(!absorbing || (cfg->deg_local->n == n_local && cfg->lam_local->n == n_local), "pgh_dist_ppr_run: deg / lam must have the slice's length");
    if (!cfg->every_row) PGH_TRY(pgh_dist_watch_isolated(g, &v_p, &v_y[0]));
    PGH_TRY(pgh_dist_prescale(g, &v_y[0], &v_xg_local));
    const int64_t cold_region = (int64_t)c->nb * c->hot;
    if (c->hot > 0) {
        PGH_TRY(gather_part(c, 0, 0, c->hot, c->main));
        PGH_TRY(gather_part(c, cold_region, c->hot, c->live, c->main));
    } else {
        PGH_TRY(gather_part(c, 0, 0, c->live, c->main));
    }
    PGH_TRY(pgh_dist_state_init(c->state));
    for (hipEvent_t ev : {c->ev_hot, c->ev_cold, c->ev_err}) PGH_HIP(hipEventRecord(ev, c->main));
    hipEvent_t t_begin = nullptr, t_end = nullptr;
    PGH_HIP(hipEventCreate(&t_begin));
    PGH_HIP(hipEventCreate(&t_end));
    PGH_HIP(hipEventRecord(t_begin, c->main));

    auto stages = [&]() -> int {
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_hot, 0));
        PGH_TRY(pgh_dist_partial_stage(g, &v_xg_full, c->state, 1));
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_cold, 0));
        PGH_TRY(pgh_dist_partial_stage(g, &v_xg_full, c->state, 2));
        return 0;
    };
    const int max_iters = cfg->max_iters;
    int it = 1, spmv = 0;                      // `it` = ConvergenceManager.iteration of the pending has_converged call
    bool pending = false, staged = false, converged = false;
    int rc = 0;
    while (it < max_iters) {                   // convergence.py:86
        const int nxt = 1 - cur;
        if (!staged && (rc = stages()) != 0) break;
        staged = false;
        if (pending) {
            // the check that followed the previous step: its flag has travelled while the stages above were enqueued
            if ((rc = bounded_wait(c->ev_err, "the residual all-reduce of the previous step")) != 0) break;
            pending = false;
            if (reinterpret_cast<const int*>(c->state_host)[6] != 0) {
                converged = true;              // whatever the stages above computed is never folded into an iterate
                break;
            }
        }
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));      // quotient and done flag of the previous step; its residual has read y[nxt]
        rc = absorbing ? pgh_dist_combine_absorb(g, &v_p, cfg->deg_local, cfg->lam_local, &v_y[nxt], &v_xg_local, c->state)
                       : pgh_dist_combine(g, &v_p, cfg->alpha, &v_y[nxt], &v_xg_local, c->state);
        if (rc != 0) break;
        PGH_HIP(hipEventRecord(c->ev_fin, c->main));
        // ---- X: the next gather vector over xGMI
        PGH_HIP(hipStreamWaitEvent(c->xs, c->ev_fin, 0));
        if (c->hot > 0) {
            if ((rc = gather_part(c, 0, 0, c->hot, c->xs)) != 0) break;
            PGH_HIP(hipEventRecord(c->ev_hot, c->xs));
            if ((rc = gather_part(c, cold_region, c->hot, c->live, c->xs)) != 0) break;
        } else {
            if ((rc = gather_part(c, 0, 0, c->live, c->xs)) != 0) break;
            PGH_HIP(hipEventRecord(c->ev_hot, c->xs));
        }
        PGH_HIP(hipEventRecord(c->ev_cold, c->xs));
        cur = nxt;
        ++spmv;
        ++it;
        const bool check = it < max_iters && kind != PGH_ERR_ITERS && it % cfg->end_modulo == 0;
        // ---- S: the scalars of the step
        PGH_HIP(hipStreamWaitEvent(c->ss, c->ev_fin, 0));
        {
            StreamSwap on_scalars(c->ss);
            if ((rc = comm_all_reduce(c, c->state + 2, 1, ncclFloat64, ncclSum, c->s, c->ss)) != 0) break;
            if ((rc = pgh_dist_close_sum(c->state, cfg->use_quotient)) != 0) break;
            if (check) {
                if ((rc = pgh_dist_residual(local_kind, &v_y[cur], &v_y[1 - cur], c->state)) != 0) break;
                if ((rc = comm_all_reduce(c, c->state + 1, 1, ncclFloat64, err_op, c->s, c->ss)) != 0) break;
                if ((rc = pgh_dist_close_err(c->state, kind, cfg->tol, cfg->n_global)) != 0) break;
                PGH_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(double) * 8, hipMemcpyDeviceToHost, c->ss));
            }
        }
        PGH_HIP(hipEventRecord(c->ev_err, c->ss));
        if (it >= max_iters) break;
        if (check) {
            pending = true;
            if ((rc = stages()) != 0) break;   // speculate: the next step's first two stages need the exchange only
            staged = true;
        }
    }
    if (rc == 0) {
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_err, 0));
        PGH_HIP(hipStreamWaitEvent(c->main, c->ev_cold, 0));
        if (pending && !converged) {
            rc = bounded_wait(c->ev_err, "the residual all-reduce of the last step");
            if (rc == 0) converged = reinterpret_cast<const int*>(c->state_host)[6] != 0;
        }
    }
    if (rc == 0) {
        PGH_HIP(hipEventRecord(t_end, c->main));
        PGH_HIP(hipMemcpyAsync(c->state_host, c->state, sizeof(double) * 8, hipMemcpyDeviceToHost, c->main));
        PGH_HIP(hipEventRecord(c->ev_host, c->main));
        rc = bounded_wait(c->ev_host, "the loop state of the last step");
    }
    if (rc != 0) {
        (void)hipEventDestroy(t_begin);
        (void)hipEventDestroy(t_end);
        return rc;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t_begin, t_end);
    (void)hipEventDestroy(t_begin);
    (void)hipEventDestroy(t_end);
    const int steps = reinterpret_cast<const int*>(c->state_host)[7];
    PGH_CHECK(steps == spmv, "pgh_dist_ppr_run: the device ran a different number of steps than the host enqueued");
    const double scale = c->state_host[0];
    const double factor = scale * (cfg->preserve_norm ? norm : 1.0);           // abstract_filters.py:63-64
    k_scale_into<<<grid_of(n_local), 256, 0, c->main>>>(c->y[cur], ranks_local->data, n_local, factor);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(c->main));
    res->iterations = it;
    res->spmv_count = spmv;
    res->converged = converged ? 1 : 0;
    res->last_error = c->state_host[6];
    res->loop_ms = (double)ms;
    res->exchange_bytes = 4LL * c->live * c->bpr * (c->world - 1);
    res->gather_slots = (int64_t)c->nb * c->live;
    res->column_blocks = c->nb;
    res->split_regions = c->hot > 0 ? 1 : 0;
    return 0;
}
