// Shared device-side definitions of the propagation kernels (internal; not part of the C-ABI).
#pragma once
#include "pgh_common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace pgh {

constexpr int WG = 256;

enum EpiMode { EPI_PLAIN = 0, EPI_AXPBY = 1, EPI_ABSORB = 2, EPI_POLY = 3 };

struct EpiParams {
    double       a;        // multiplies the row sum (alpha, or 1 / 2 for the polynomial recurrences)
    double       b;        // multiplies v[row]  ((1 - alpha) for PageRank, 0 / -1 for polynomial terms)
    const float* v;        // personalization p (PPR / ABSORB) or the current term (POLY with b != 0)
    const float* deg;      // ABSORB: degrees(M)
    const float* lam;      // ABSORB: absorption * (1 - alpha) / alpha
    float*       y;        // output vector
    float*       r;        // POLY: result accumulator (in place)
    double       c;        // POLY: coefficient of the new term
    int          err_linf; // POLY: delta is a max instead of a sum
    float*       xg_out;   // blocked format with a source scale: next gather source y * src_scale (else null)
    const float* src_scale;
    int          xg_blk;   // trimmed gather vector (BsfFormat::xg_live): slots per block in the id space ...
    int          xg_live;  // ... and slots per block actually stored (0 = stored in full, slot = row)
    int          xg_hot;   // > 0 (partitioned runs): the first xg_hot slots of every block are stored block after block at the front,
    int          xg_cold;  // ... the slots [xg_hot, xg_live) block after block from position xg_cold on (two contiguous exchange regions)
};

// position of internal id `row` inside a gather vector that keeps only the first `live` slots of every block of `blk`
// ids (never-referenced sources sort last inside a block and are not stored); -1 = not stored
// (hot > 0: the split form of a partitioned run -- [block][hot] first, then [block][live - hot] from `cold` on)
__device__ __forceinline__ int xg_slot(int row, int blk, int live, int hot = 0, int cold = 0) {
    if (live == 0) return row;
    int b = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) b += (row >= k * blk) ? 1 : 0;
    const int loc = row - b * blk;
    if (loc >= live) return -1;
    return loc < hot ? b * hot + loc : cold + b * (live - hot) + (loc - hot);
}

// Device-resident loop state (ConvergenceManager on the device, convergence.py:77-101).
struct LoopState {
    double scale;       // lazily applied L1 quotient of the current iterate (abstract_filters.py:133-134)
    double err;         // last residual
    double sum;         // sum(y) of the last step
    int    done;        // convergence flag: once set every later kernel of the loop is a no-op
    int    steps;       // propagation steps executed so far
    int    converged;   // 1 when the tolerance was met
    int    pad;
};

// Residual of a PageRank step INSIDE the finish kernel (k_pb_finish<..., RES>; L1 / Mabs rules).  The residual
// sum_i |y_i * inv - x_i * scale| needs inv = 1 / sum(y) of the step that is being written, which is only known when the
// kernel ends -- but it can be PREDICTED: sum(y) = a * scale * sum_j deg_j * x_j + b * sum(p) with deg = the row sums of M
// (column sums of M^T), up to the f32 roundings of the step (a relative 1e-8: the stored degrees and the products the
// kernels form round differently), and that bias barely moves from one step to the next, so the prediction is multiplied
// by the ratio measured / predicted of the previous step: it then misses by 1e-10 .. 1e-12.  The kernel evaluates
//     R' = sum_i |y_i * inv' - x_i * scale|   and   D = sum_i sign(y_i * inv' - x_i * scale) * y_i
// against the predicted inv', and the close takes  R = R' + (inv - inv') * D : exact but for rows whose term changes sign
// between inv' and inv, each wrong by less than 2 |inv - inv'| |y_i| -- in total less than 2 |inv - inv'| sum(y) while no y_i is
// negative (a workgroup that meets a negative y_i -- a signed personalization -- reports R' = NaN, which pauses).  The close
// therefore knows the residual to within that bound; when the tolerance lies inside it (or anything is not finite) the
// step is PAUSED: the host re-evaluates it with the separate residual kernel and goes on without the fusion.  The stopping
// decision is the one the separate kernel would take, always.  The kernel also accumulates T = sum_j deg_j * y_j for the
// next prediction; the first step of a run has no prediction: it accumulates sum(p) in D's place and uses the separate kernel.
struct LoopAux {
    double pred_inv[2];   // predicted 1 / sum(y) of step k at [k & 1]
    double pred_raw[2];   // the uncorrected prediction of sum(y) of step k at [k & 1] (0: none)
    double sum_p;         // sum of the (normalised) personalization
    double worst_miss;    // largest |inv - inv'| / |inv| seen by a checking close of this run (diagnostic)
    double in_norm;       // L1 norm of the caller's personalization when the run computed it itself (pgh_loop_cfg::in_norm < 0)
    // what the way INTO the id space sums for the FIRST step's prediction (k_permute_in_pair: block sums added as 2^-40 fixed point,
    // so that the order of the workgroups does not show): sum_j deg_j * x0_j and sum_j p_j of the normalised operands
    long long t0_fix, sp_fix;
    unsigned long long t_begin;   // the device's constant-rate clock (s_memrealtime) when the run's first kernel started: loop_ms without events
};
constexpr double kPredFix = 1099511627776.0;      // 2^40
struct ResParams {
    const float*   x_prev;   // previous iterate (internal ids, un-normalised; its quotient is LoopState::scale)
    const float*   deg;      // row sums of M in internal ids
    const LoopAux* aux;
    double*        part_r;   // per-workgroup / per-tail-item partials, like partial_sum
    double*        part_d;
    double*        part_t;
    int            step;     // k
    int            first;    // 1: no prediction yet (D accumulates sum(p), R is not used)
};

// The close of a step (k_step_close: fold the partials, update the loop state, evaluate the stopping rule) can be
// DEFERRED into the first kernel of the next step: every workgroup of that kernel folds the same partials in the same
// order (so they all reach the same verdict, and the same bits as k_step_close), workgroup 0 writes the state.  One
// launch and one dependent boundary fewer per iteration (k_step_close: 5.4 us + the gap around it).
struct PendingClose {
    LoopState*    state;
    const double* partial_sum;
    const double* res_partials;   // residual partials (recursive filters)
    int*          progress;       // host-visible {steps, done} or null
    double        tol;
    long long     n;
    int           num_sum, num_res;
    int           use_quotient, check, err_kind;
    int           active;
    // fused residual (ResParams): 0 = off, 1 = the finish kernel evaluated the residual against the predicted quotient,
    // 2 = first step of such a run (residual from the separate kernel; sum(p) arrives in part_d)
    int           res_mode;
    int           step;
    LoopAux*      aux;
    const double* part_r;
    const double* part_d;
    const double* part_t;
    double        a, b;           // the step is y = a * scale * (M^T x) + b * p
    unsigned long long tag;       // this run's number: part of the checksum of the state a close publishes (a stale block never fits)
    int           first_pred;     // 1: no close to run, but the prediction of step 1 is made from LoopAux::t0_fix / sp_fix (first_prediction)
};

// The isolated tail of every column block of a square relabelled graph (BsfFormat::iso_begin): rows without entries that
// nobody references.  flag: device word, 0 = this run's operands are zero on all of them (they stay zero: skipped).
struct IsoTail {
    int* flag;
    int  blk;
    int  shift;          // log2(blk) when the block size is a power of two (it usually is), else -1
    int  num_blocks;
    int  begin[8];
    __device__ __forceinline__ bool holds(int64_t row) const {
        if (blk <= 0) return false;
        const unsigned int r32 = (unsigned int)row;            // ids are below 2^31
        const int b = shift >= 0 ? (int)(r32 >> shift) : (int)(r32 / (unsigned int)blk);
        // (a chain of selects over the 8 table entries: indexing a kernel-argument array with a per-lane value sends the table through
        // scratch memory -- the way out of the id space took 54 us with it, 40 without, tools/permute_probe.hip.  Round 5: and a select
        // between two LOADS of kernel arguments is folded into one load through a selected address -- a vector load from the argument
        // segment with a full wait behind it, once per id; readfirstlane makes the entries scalar VALUES before the selects)
        int first = __builtin_amdgcn_readfirstlane(begin[0]);
#pragma unroll
        for (int k = 1; k < 8; ++k) first = b == k ? __builtin_amdgcn_readfirstlane(begin[k]) : first;
        return b < num_blocks && (int)(r32 - (unsigned int)b * (unsigned int)blk) >= first;
    }
};

struct GraphView {
    const int32_t* rowptr;
    const int32_t* col;
    const float*   val;
    const int2*    tile_coord;
    const int32_t* chain_first;
    double*        tail_carry;
    double*        head_partial;
    int            n;          // rows of M^T (= length of y)
    int            num_tiles;
};

// The epilogue of a step in two halves, so that a kernel can issue the operand loads of several rows before the first
// dependent use (one row after the other leaves a thread with a chain of exposed memory latencies).
struct EpiOps {
    float v, deg, lam, src, r_old;
    int   slot;           // position of the row in the (trimmed) next gather vector, -1 = not stored
};
// element `byte_off / sizeof(T)` of an array: a uniform base plus a 32-bit unsigned byte offset is one scalar register
// pair and ONE address register, shared by every array that is indexed by the same row (a 64-bit address per load costs
// two registers each and was what pushed the fused kernels into scratch)
template <typename T>
__device__ __forceinline__ T ld_off(const T* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_off(T* base, uint32_t byte_off, T value) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = value;
}
template <int MODE>
__device__ __forceinline__ EpiOps epi_load(const EpiParams& ep, int row) {
    EpiOps o;
    o.v = 0.f, o.deg = 0.f, o.lam = 0.f, o.src = 0.f, o.r_old = 0.f, o.slot = -1;
    const uint32_t at = (uint32_t)row << 2;
    if (MODE == EPI_AXPBY || MODE == EPI_POLY) {
        if (ep.v != nullptr) o.v = ld_off(ep.v, at);
    } else if (MODE == EPI_ABSORB) {
        o.v = ld_off(ep.v, at);
        o.deg = ld_off(ep.deg, at);
        o.lam = ld_off(ep.lam, at);
    }
    if (ep.xg_out != nullptr) {
        o.slot = xg_slot(row, ep.xg_blk, ep.xg_live, ep.xg_hot, ep.xg_cold);
        o.src = ld_off(ep.src_scale, at);     // unconditional: a load under a divergent branch serialises the loads around it
    }
    if (MODE == EPI_POLY) o.r_old = ld_off(const_cast<const float*>(ep.r), at);
    return o;
}
template <int MODE>
__device__ __forceinline__ float epi_apply(const EpiParams& ep, const EpiOps& o, float a_eff, int row, float sum,
                                           double& sum_y, double& delta) {
    float y;
    if (MODE == EPI_PLAIN) {
        y = a_eff * sum;
    } else if (MODE == EPI_AXPBY || MODE == EPI_POLY) {
        y = a_eff * sum;
        if (ep.v != nullptr) y += (float)ep.b * o.v;
    } else {   // EPI_ABSORB: ((M^T x) * deg + p * lam) / (lam + deg), adhoc.py:167-168
        y = (a_eff * sum * o.deg + o.v * o.lam) / (o.lam + o.deg);
    }
    const uint32_t at = (uint32_t)row << 2;
    st_off(ep.y, at, y);
    if (o.slot >= 0) st_off(ep.xg_out, (uint32_t)o.slot << 2, y * o.src);
    sum_y += (double)y;
    if (MODE == EPI_POLY) {
        const float r_new = o.r_old + (float)ep.c * y;
        st_off(ep.r, at, r_new);
        const double d = fabs((double)r_new - (double)o.r_old);
        delta = ep.err_linf ? fmax(delta, d) : delta + d;
    }
    return y;
}
// The same two halves WITHOUT run-time branches around the loads (round 5): which operands a run has is decided per launch, but a load
// under a branch -- even a wavefront-uniform one -- gets a basic block and a drained wait of its own, and a row's operands then arrive
// one round trip after the other.  A missing operand reads `zero` (>= 4 zero bytes: the zero slot of the partial sums) and its use is a
// select.  `slot`: the caller's (xg_slot, or -1); `live`: false = the lane repeats a row of another lane -- nothing is stored or summed.
template <int MODE>
__device__ __forceinline__ EpiOps epi_load_z(const EpiParams& ep, int row, const char* zero) {
    EpiOps o;
    o.v = 0.f, o.deg = 0.f, o.lam = 0.f, o.src = 0.f, o.r_old = 0.f, o.slot = -1;
    const uint32_t at = (uint32_t)row << 2;
    if (MODE == EPI_AXPBY || MODE == EPI_POLY) {
        const bool has_v = ep.v != nullptr;
        o.v = *reinterpret_cast<const float*>((has_v ? reinterpret_cast<const char*>(ep.v) : zero) + (has_v ? at : 0u));
    } else if (MODE == EPI_ABSORB) {
        o.v = ld_off(ep.v, at);
        o.deg = ld_off(ep.deg, at);
        o.lam = ld_off(ep.lam, at);
    }
    const bool has_xg = ep.xg_out != nullptr;
    o.src = *reinterpret_cast<const float*>((has_xg ? reinterpret_cast<const char*>(ep.src_scale) : zero) + (has_xg ? at : 0u));
    if (MODE == EPI_POLY) o.r_old = ld_off(const_cast<const float*>(ep.r), at);
    return o;
}
template <int MODE>
__device__ __forceinline__ float epi_apply_z(const EpiParams& ep, const EpiOps& o, float a_eff, int row, float sum, bool live,
                                             double& sum_y, double& delta) {
    float y;
    if (MODE == EPI_PLAIN) {
        y = a_eff * sum;
    } else if (MODE == EPI_AXPBY || MODE == EPI_POLY) {
        y = a_eff * sum;
        y = ep.v != nullptr ? y + (float)ep.b * o.v : y;
    } else {   // EPI_ABSORB: ((M^T x) * deg + p * lam) / (lam + deg), adhoc.py:167-168
        y = (a_eff * sum * o.deg + o.v * o.lam) / (o.lam + o.deg);
    }
    const uint32_t at = (uint32_t)row << 2;
    float r_new = 0.f;
    if (MODE == EPI_POLY) r_new = o.r_old + (float)ep.c * y;
    if (live) {
        st_off(ep.y, at, y);
        if (o.slot >= 0) st_off(ep.xg_out, (uint32_t)o.slot << 2, y * o.src);
        if (MODE == EPI_POLY) st_off(ep.r, at, r_new);
    }
    y = live ? y : 0.f;
    sum_y += (double)y;
    if (MODE == EPI_POLY) {
        const double d = live ? fabs((double)r_new - (double)o.r_old) : 0.0;
        delta = ep.err_linf ? fmax(delta, d) : delta + d;
    }
    return y;
}
template <int MODE>
__device__ __forceinline__ float apply_epilogue(const EpiParams& ep, float a_eff, int row, float sum,
                                                double& sum_y, double& delta) {
    return epi_apply<MODE>(ep, epi_load<MODE>(ep, row), a_eff, row, sum, sum_y, delta);
}

// graph_dropout(M, rate) (pytorch.py:34-38: torch.nn.functional.dropout on the edge values): an entry survives with
// probability 1 - rate and is then scaled by 1 / (1 - rate); the mask is a pure function of (seed, entry index in
// CSR(M^T) order), so a dropped graph is a (graph, rate, seed) triple and costs no memory.  Host twin: tests/kernel_checks.py.
__device__ __forceinline__ float dropout_factor(uint64_t seed, uint64_t entry, uint32_t threshold, float keep_scale) {
    uint64_t z = (seed ^ (entry * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32) >= threshold ? keep_scale : 0.f;
}

// graph_dropout on the blocked layouts (single vectors): every entry of the stream and of the cold image carries its index in
// CSR(M^T) order in a word of its own (BsfFormat::drop_edge / PbFormat::drop_edge, built on the first dropout launch:
// bsf_ensure_edge_ids), so the mask is the one of the row-major kernel, of the multi-seed kernel and of degrees(dropped graph) --
// repeated entries of a multigraph share one bit.  edge == nullptr: no dropout.
struct DropView {
    const int32_t* edge;
    uint64_t       seed;
    uint32_t       threshold;    // floor(rate * 2^32)
    float          keep_scale;   // 1 / (1 - rate)
};

// Row sum of the blocked SpMV layout: the row's segment of every column block, found through the block's SegMeta
// word (BsfFormat::psum / meta), added in block order in f64.
struct RowSums {
    const SegMeta* meta;
    const float*   psum;
    int64_t        words;       // SegMeta words per block
    int            num_blocks;
    unsigned int   zero_at;     // a slot of psum that always holds 0 (branch-free lookups of rows without a segment)
};
template <int B>
__device__ __forceinline__ double block_row_sum(const RowSums& rs, int64_t row) {
    const int64_t w = row >> 6;
    const unsigned long long bit = 1ULL << (row & 63);
    // all map words first, then all segment sums: two exposed latencies per row instead of two per block -- and no run-time branch
    // around a load (a block past num_blocks re-reads block 0's word, a row without a segment the zero slot: a load under `if` gets
    // a basic block and a drained wait of its own, and the eight of a row then take eight round trips one after the other)
    SegMeta m[B];
#pragma unroll
    for (int b = 0; b < B; ++b) m[b] = rs.meta[(b < rs.num_blocks ? (int64_t)b * rs.words : 0) + w];
    float v[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const bool has = b < rs.num_blocks && (m[b].mask & bit) != 0ULL;
        v[b] = rs.psum[has ? (unsigned int)m[b].base + (unsigned int)__popcll(m[b].mask & (bit - 1ULL)) : rs.zero_at];
    }
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < B; ++b)
        if (b < rs.num_blocks) s += (double)v[b];
    return s;
}
// the same lookup in two steps for kernels that keep several rows in flight
template <int B>
struct RowLookup {
    SegMeta m[B];
};
template <int B>
__device__ __forceinline__ void row_lookup_meta(const RowSums& rs, int64_t row, RowLookup<B>& q) {
    const int64_t w = row >> 6;
#pragma unroll
    for (int b = 0; b < B; ++b)
        if (b < rs.num_blocks) q.m[b] = rs.meta[(int64_t)b * rs.words + w];
}
template <int B>
__device__ __forceinline__ void row_lookup_vals(const RowSums& rs, int64_t row, const RowLookup<B>& q, float (&v)[B]) {
    const unsigned long long bit = 1ULL << (row & 63);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        v[b] = 0.f;
        if (b < rs.num_blocks && (q.m[b].mask & bit)) v[b] = rs.psum[q.m[b].base + __popcll(q.m[b].mask & (bit - 1ULL))];
    }
}

// Cross-tile fix-up of the blocked SpMV layout: a row segment that spans tiles is closed by adding the carries of the
// tiles it crosses in a fixed order (deterministic, atomic-free).  One thread per closing tile; a hub row spans
// hundreds of tiles, so chains of 32+ tiles are summed by the whole wavefront (strided lanes + fixed reduction tree)
// instead of leaving one thread with a serial chain of dependent loads.  Runs as its own launch (k_bsf_fixup) or at
// the start of the cold image's phase A (k_pb_gather), which needs nothing from it.
struct FixView {
    const int32_t* fix_seg;    // [num_tiles] segment (index into psum) that receives tile t's fix-up, or -1
    const int4*    tile;       // .w = first tile of the carry chain that ends in tile t
    const double*  tail_carry;
    const double*  head_partial;
    float*         psum;
    int            num_tiles;
};
// `first`, `stride`: this workgroup's threads take tiles first + threadIdx.x, then += stride (whole wavefronts: the
// cooperative part uses shuffles, so the trip count is wavefront-uniform)
__device__ __forceinline__ void bsf_fixup_tiles(const FixView& f, int first_tile, int stride) {
    const int lane = threadIdx.x & 63;
    for (int t0 = first_tile; t0 < f.num_tiles; t0 += stride) {
        const int t = t0 + (int)threadIdx.x;
        const int dst = t < f.num_tiles ? f.fix_seg[t] : -1;
        const int first = dst >= 0 ? f.tile[t].w : 0;
        const int len = dst >= 0 ? t - first : 0;
        const bool is_long = len >= 32;
        if (dst >= 0 && !is_long) {
            double total = 0.0;
            for (int s = first; s < t; ++s) total += f.tail_carry[s];
            total += f.head_partial[t];
            f.psum[dst] = (float)total;
        }
        unsigned long long todo = __ballot(is_long);
        while (todo != 0ULL) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ULL;
            const int c_first = __shfl(first, src, 64), c_t = __shfl(t, src, 64);
            double part_sum = 0.0;
            for (int s = c_first + lane; s < c_t; s += 64) part_sum += f.tail_carry[s];
            part_sum = wave_reduce_sum(part_sum);
            const double total = __shfl(part_sum, 0, 64) + f.head_partial[c_t];
            if (lane == src) f.psum[dst] = (float)total;
        }
    }
}

// diagnostic builds only (-DPGH_PROBE_TIMES=1): every workgroup stamps its start and end (100 MHz wall clock) so that the host
// can print how evenly a launch's workgroups finish (tools/probe_variants.py with PGH_DUMP_TIMES=1)
#ifndef PGH_PROBE_TIMES
#define PGH_PROBE_TIMES 0
#endif
#if PGH_PROBE_TIMES
#define PGH_STAMP_DECL(NAME) __device__ unsigned long long NAME[2 * 4096];
#define PGH_STAMP_BEGIN(NAME) \
    if (threadIdx.x == 0 && blockIdx.x < 4096) NAME[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#define PGH_STAMP_END(NAME)                                                                              \
    __syncthreads();                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 4096) NAME[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#define PGH_STAMP_DUMP(NAME, GRID, LABEL)                                                                               \
    if (getenv("PGH_DUMP_TIMES") != nullptr) {                                                                          \
        static int dumped = 0;                                                                                          \
        if (dumped++ == 8) {                                                                                            \
            std::vector<unsigned long long> h(2 * 4096);                                                                \
            (void)hipStreamSynchronize(rt().stream);                                                                    \
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(NAME), sizeof(unsigned long long) * 2 * 4096);               \
            const int n_ = (GRID) < 4096 ? (GRID) : 4096;                                                               \
            unsigned long long t0 = ~0ULL;                                                                              \
            for (int i = 0; i < n_; ++i) t0 = h[2 * i] < t0 ? h[2 * i] : t0;                                            \
            std::vector<double> st(n_), en(n_);                                                                         \
            for (int i = 0; i < n_; ++i) st[i] = (h[2 * i] - t0) * 0.01, en[i] = (h[2 * i + 1] - t0) * 0.01;            \
            std::sort(st.begin(), st.end());                                                                            \
            std::sort(en.begin(), en.end());                                                                            \
            double mean = 0;                                                                                            \
            for (double v : en) mean += v / n_;                                                                         \
            fprintf(stderr, "[pgh times] %s: %d workgroups, start max %.1f us; end min %.1f p10 %.1f median %.1f mean %.1f p90 %.1f max %.1f us\n", \
                    LABEL, n_, st[n_ - 1], en[0], en[n_ / 10], en[n_ / 2], mean, en[n_ * 9 / 10], en[n_ - 1]);          \
            if (getenv("PGH_DUMP_TIMES_RAW") != nullptr) {                                                              \
                FILE* raw_ = fopen(getenv("PGH_DUMP_TIMES_RAW"), "a");                                                  \
                if (raw_ != nullptr) {                                                                                  \
                    for (int i = 0; i < n_; ++i)                                                                        \
                        fprintf(raw_, "%s,%d,%.2f,%.2f\n", LABEL, i, (h[2 * i] - t0) * 0.01, (h[2 * i + 1] - t0) * 0.01); \
                    fclose(raw_);                                                                                       \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    }
#else
#define PGH_STAMP_DECL(NAME)
#define PGH_STAMP_BEGIN(NAME)
#define PGH_STAMP_END(NAME)
#define PGH_STAMP_DUMP(NAME, GRID, LABEL)
#endif

// fold of <= kMaxPartials partials by the first 256 threads of a workgroup of any size >= 256, in the order k_step_close /
constexpr int kFoldBatch = 6;     // strides of 256 partials in flight per thread (1024 workgroups + the tail's items of the finish kernel: 5-6)
// fold_partials use (thread t adds partials t, t + 256, ...; wavefront shuffles; wavefronts 0..3 in order): every thread
// returns the result.  s4: LDS scratch of 4 doubles.
__device__ __forceinline__ double fold_partials_wide(const double* __restrict__ partials, int count, int linf, double* s4) {
    double acc = 0.0;
    if (threadIdx.x < 256) {
        // the loads of kFoldBatch strides are issued together, the additions keep their order (round 5: one load, one wait, one add per
        // stride was five to six memory round trips in the prologue of every block-partial-sums launch -- ~1300 partials, 256 threads)
        for (int i0 = threadIdx.x; i0 < count; i0 += 256 * kFoldBatch) {
            double v[kFoldBatch];
#pragma unroll
            for (int u = 0; u < kFoldBatch; ++u) v[u] = partials[min(i0 + 256 * u, count - 1)];
#pragma unroll
            for (int u = 0; u < kFoldBatch; ++u)
                if (i0 + 256 * u < count) acc = linf ? fmax(acc, v[u]) : acc + v[u];
        }
        acc = linf ? wave_reduce_max(acc) : wave_reduce_sum(acc);
    }
    __syncthreads();
    if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = acc;
    __syncthreads();
    double r = s4[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = linf ? fmax(r, s4[w]) : r + s4[w];
    return r;
}

// K arrays of `count` partials folded at once (sums), same order as fold_partials_wide: one barrier pair for all of them.
// s: LDS scratch of 4 * K doubles.
template <int K>
__device__ __forceinline__ void fold_partials_multi(const double* const (&arr)[K], int count, double (&out)[K], double* s) {
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    if (threadIdx.x < 256) {
        for (int i0 = threadIdx.x; i0 < count; i0 += 256 * kFoldBatch) {
            double v[K][kFoldBatch];
#pragma unroll
            for (int u = 0; u < kFoldBatch; ++u) {
                const int i = min(i0 + 256 * u, count - 1);
#pragma unroll
                for (int k = 0; k < K; ++k) v[k][u] = arr[k][i];
            }
#pragma unroll
            for (int u = 0; u < kFoldBatch; ++u)
                if (i0 + 256 * u < count) {
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[k] += v[k][u];
                }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = wave_reduce_sum(acc[k]);
    }
    __syncthreads();
    if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) s[k * 4 + (threadIdx.x >> 6)] = acc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = ((s[k * 4] + s[k * 4 + 1]) + s[k * 4 + 2]) + s[k * 4 + 3];
}

// What a close decides, from the folded partials (shared by k_step_close and the deferred close so that both produce the
// same bits).  verdict: 0 = go on, 1 = converged, 2 = paused (fused residual: the quotient's prediction missed, kPredGuard).
struct CloseOutcome {
    double S, scale, err, next_pred, next_raw, sum_p, miss;
    int    verdict;
};
__device__ __forceinline__ CloseOutcome close_outcome(const PendingClose& pc, double S, double err_plain, double R, double D, double T) {
    CloseOutcome o;
    o.S = S;
    o.scale = pc.use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;
    o.err = err_plain;
    o.miss = 0.0;
    o.verdict = 0;
    o.sum_p = 0.0;
    o.next_pred = 1.0;
    o.next_raw = 0.0;
    double slack = 0.0;                  // the residual is known to within this (fused residual)
    if (pc.res_mode != 0) {
        o.sum_p = pc.res_mode == 2 ? D : pc.aux->sum_p;
        if (pc.use_quotient) {
            // with the factors as the epilogue forms them: a_eff = (float)(a * scale), (float)b (epi_apply)
            o.next_raw = (double)(float)(pc.a * o.scale) * T + (double)(float)pc.b * o.sum_p;
            const double raw_now = pc.aux->pred_raw[pc.step & 1];
            const double ratio = (pc.res_mode == 1 && raw_now != 0.0) ? S / raw_now : 1.0;
            const double S_next = o.next_raw * (ratio == ratio && fabs(ratio - 1.0) < 1e-4 ? ratio : 1.0);
            o.next_pred = S_next != 0.0 ? 1.0 / S_next : 0.0;
        }
        if (pc.res_mode == 1 && pc.check) {
            const double ip = pc.aux->pred_inv[pc.step & 1];
            o.err = R + (o.scale - ip) * D;
            o.miss = fabs(o.scale - ip) / fmax(fabs(o.scale), 1e-300);
            slack = 2.0 * fabs(o.scale - ip) * fabs(S);
        }
    }
    if (pc.check) {
        if (pc.err_kind == PGH_ERR_MABS) {
            o.err /= (double)pc.n;
            slack /= (double)pc.n;
        }
        if (pc.res_mode == 1 && !(fabs(o.err - pc.tol) > 2.0 * slack)) o.verdict = 2;      // too close to call (or not finite)
        else if (o.err <= pc.tol) o.verdict = 1;
    }
    return o;
}
// What the host reads at the END of a run without copying the state back (pgh_spmv.hip::published_state): the closes of a loop leave
// {scale, err, in_norm, steps | done | converged, device ticks since the run's first kernel} and a checksum of the five with the run's
// number, beside the progress words (bytes 16-63 of the same pinned, mapped 64 bytes).  Relaxed stores: the host accepts the words only
// when the checksum fits them and this run, and the progress words agree.
constexpr unsigned long long kPublishMagic = 0x9e3779b97f4a7c15ULL;
__host__ __device__ __forceinline__ unsigned long long publish_checksum(unsigned long long a, unsigned long long b, unsigned long long c,
                                                                        unsigned long long d, unsigned long long e, unsigned long long tag) {
    return a ^ ((b << 1) | (b >> 63)) ^ ((c << 2) | (c >> 62)) ^ ((d << 3) | (d >> 61)) ^ ((e << 4) | (e >> 60)) ^ (tag * kPublishMagic);
}
__device__ __forceinline__ void publish_state(int* progress, const LoopState* st, const LoopAux* aux, unsigned long long tag) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(progress) + 2;
    const unsigned long long a = (unsigned long long)__double_as_longlong(st->scale), b = (unsigned long long)__double_as_longlong(st->err),
                             c = (unsigned long long)__double_as_longlong(aux != nullptr ? aux->in_norm : 0.0),
                             d = ((unsigned long long)(unsigned)st->steps << 32) | ((unsigned long long)(unsigned)(st->done & 0xff) << 8) |
                                 (unsigned long long)(unsigned)(st->converged & 0xff),
                             e = aux != nullptr ? __builtin_amdgcn_s_memrealtime() - aux->t_begin : 0ULL;
    __hip_atomic_store(w + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 3, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 4, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 5, publish_checksum(a, b, c, d, e, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the state update of a close (one thread)
__device__ __forceinline__ void close_commit(const PendingClose& pc, const CloseOutcome& o) {
    LoopState* state = pc.state;
    if (o.verdict == 2) {                // paused: the step stays open, the host re-evaluates it
        state->done = 2;
        if (o.miss > pc.aux->worst_miss) pc.aux->worst_miss = o.miss;
        if (pc.progress != nullptr) __hip_atomic_store(pc.progress + 1, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    state->sum = o.S;
    state->scale = o.scale;
    const int steps = state->steps + 1;
    state->steps = steps;
    if (pc.res_mode != 0) {
        pc.aux->pred_inv[(pc.step + 1) & 1] = o.next_pred;
        pc.aux->pred_raw[(pc.step + 1) & 1] = o.next_raw;
        if (pc.res_mode == 2) pc.aux->sum_p = o.sum_p;
        if (o.miss > pc.aux->worst_miss) pc.aux->worst_miss = o.miss;
    }
    if (pc.check) {
        state->err = o.err;
        if (o.verdict == 1) {
            state->done = 1;
            state->converged = 1;
        }
    }
    // (relaxed stores: the host only PACES itself by these words and reads every result after a stream synchronisation; a
    // system-scope release here is a write-back of the XCD's whole L2 in the middle of the launch that carries the close)
    if (pc.progress != nullptr) {        // host-visible progress word (pinned, mapped): lets the host run ahead without syncs
        if (pc.tag != 0ULL) publish_state(pc.progress, state, pc.aux, pc.tag);
        __hip_atomic_store(pc.progress + 1, o.verdict == 1 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(pc.progress, steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The prediction of the FIRST step (its residual inside the finish kernel too): sum(y_1) = a * scale_0 * sum_j deg_j x0_j
// + b * sum(p), both sums made by the pass that brought the operands into the id space.  One thread of the first kernel of step 1
// (a sum that does not fit the fixed point, or none: the prediction is garbage, the close's bound sees it and pauses).
__device__ __forceinline__ void first_prediction(const PendingClose& pc) {
    LoopAux* aux = pc.aux;
    const double t0 = (double)aux->t0_fix / kPredFix, sp = (double)aux->sp_fix / kPredFix;
    aux->sum_p = sp;
    if (pc.use_quotient) {
        const double raw = (double)(float)(pc.a * pc.state->scale) * t0 + (double)(float)pc.b * sp;
        aux->pred_raw[1] = raw;
        aux->pred_inv[1] = raw != 0.0 ? 1.0 / raw : 0.0;
    }
}

// the deferred close (see PendingClose); returns true when the loop has ended or paused, i.e. the calling kernel must do
// nothing.  s: LDS scratch of 16 doubles.
__device__ __forceinline__ bool run_pending_close(const PendingClose& pc, double* s) {
    double S, err = 0.0, R = 0.0, D = 0.0, T = 0.0;
    if (pc.res_mode == 0) {
        S = fold_partials_wide(pc.partial_sum, pc.num_sum, 0, s);
        if (pc.check) {
            __syncthreads();
            err = fold_partials_wide(pc.res_partials, pc.num_res, pc.err_kind == PGH_ERR_LINF, s);
        }
    } else {
        const double* const arr[4] = {pc.partial_sum, pc.part_t, pc.part_d, pc.part_r};
        double out[4];
        fold_partials_multi<4>(arr, pc.num_sum, out, s);
        S = out[0], T = out[1], D = out[2], R = out[3];
        if (pc.res_mode == 2 && pc.check) {
            __syncthreads();
            err = fold_partials_wide(pc.res_partials, pc.num_res, 0, s);
        }
    }
    const CloseOutcome o = close_outcome(pc, S, err, R, D, T);
    if (blockIdx.x == 0 && threadIdx.x == 0) close_commit(pc, o);
    __syncthreads();
    return o.verdict != 0;
}

// Where the 8 values of B-order group G of the cold image live in PbFormat::tmp: inside every block of 64 groups the 64
// low quads come first, then the 64 high quads -- so that both 16-byte stores of phase A (lane = group) and both 16-byte
// loads of phase B cover CONTIGUOUS memory when the lanes of a wavefront hold consecutive groups (they mostly do: a
// (chunk, bin) run is ~80 groups).  With the quads of a group side by side every such instruction touched every second
// 16 bytes: half-filled requests on both sides.
// Images whose (chunk, bin) runs are short -- the slices of a partitioned graph gather from 2-16x more chunks: ~1-5 groups per
// run -- keep the two quads of a group side by side instead (planes == 0): an isolated group then touches ONE 32-byte range
// instead of two 16-byte ranges in different lines (measured on the slices of the N-GPU bench, profiles/r03/partition_slices.log).
__device__ __forceinline__ uint32_t pb_tmp_quad(uint32_t group, int high, int planes) {
    return planes ? ((group >> 6) << 9) + ((uint32_t)high << 8) + ((group & 63u) << 2) : (group << 3) + ((uint32_t)high << 2);
}
// device view of the propagation-blocking image (PbFormat, pgh_pb.hip)
struct PbView {
    const uint16_t* sloc;
    const float*    val;
    const uint32_t* dstg;
    const int4*     task;
    const int*      task_range;
    const int4*     first_task;        // [num_tasks] the first piece of every phase A workgroup
    float*          tmp;
    int             tmp_planes;  // layout of tmp (pb_tmp_quad)
    int             short_piece; // phase A pieces with fewer entries run rounds of one group per lane
    const int4*     item_a;            // work list of k_pb_finish (PbFormat::item_a / item_b)
    const int4*     item_b;
    int             num_items;
    const int*      sched;             // item order (PbFormat::sched); static deal: slices sched_begin[w] .. sched_begin[w + 1]
    const int*      sched_begin;
    const int4*     first_a;           // PbFormat::first_a / first_b / first_item
    const int4*     first_b;
    const int*      first_item;
    uint32_t*       work_counter;      // hand-out of the schedule's tail: next position (null without a tail)
    int             tail_begin, tail_count;   // sched[tail_begin .. tail_begin + tail_count): handed out dynamically
    uint32_t*       hub_ticket;
    const uint16_t* drow;
    double*         hub_part;          // [num_bins] sums of the pieces of split hub rows
    uint32_t*       amax;              // [0] bit pattern of max |value| written by phase A, [1] phase B's exit tickets
    const int*      iso_flag;          // BsfFormat::iso_flag or null: 0 = items marked -2 (isolated rows) are passed over
    // the finish kernel in two launches (partitioned runs with more than one rank): phase 1 = the items that hold a row whose slot of
    // the next gather vector is EXCHANGED (slot inside its block < phase_live), phase 2 = the others; 0 = everything in one launch
    int             phase, phase_blk, phase_live;
    int64_t         cold_prefix[9];
    int64_t         xg_base[8];
    int             num_blocks, hot, chunk, num_chunks, num_bins;
    int64_t         num_cold;          // referenced cold sources in total
};

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// slot the loop driver fills (pgh_spmv.hip) and the next blocked-format launch consumes (pgh_bsf.hip)
PendingClose& pending_close_slot();

// blocked-format entry points (pgh_bsf.hip)
// pgh_pb.hip: propagation-blocking image of the cold entries
struct PbPlan {
    int32_t* row_bin = nullptr;    // device: bin of every output row, -1 = too heavy for a bin
    int      num_bins = 0, num_chunks = 0, bin_rows = 0, heavy_row = 0;
    int64_t  entries = 0;          // cold entries that go into the image
    int      slices = 1;           // the bins are cut into `slices` consecutive groups of about equal entry counts
    int      slice_first[kPbMaxSlices + 1] = {0};   // first bin of every slice
    int64_t  slice_entries[kPbMaxSlices] = {0};
    int      num_split = 0;
    int4*    host_split = nullptr; // {row, first bin, pieces, -} of the hub rows with several bins (new[]), owned by the plan
    int4*    host_bins = nullptr;  // {first row, rows, -, entries} of every bin (new[]), owned by the plan
    bool     heavy_rows = false;   // some rows keep their cold entries in the blocked stream
    // need lists (BsfFormat::want_compact): dense cold id (block-major slot - hot) -> compact id of the referenced ones, or null
    uint32_t* cold_rank = nullptr;
    int64_t  dense_prefix[9] = {0}, compact_prefix[9] = {0};
};
int pb_plan(BsfFormat& f, const uint64_t* keys, int64_t E, const int* live, int hot, unsigned char* is_hot, PbPlan* plan, bool* use);
int pb_build(BsfFormat& f, PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, const int* live, int hot);
// compacts the entries of one slice (stream keys whose row belongs to bins [slice_first[s], slice_first[s + 1])) out of the cold keys
int pb_select_slice(const PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, uint64_t* keys_out,
                    float* vals_out, int64_t* selected);
void pb_plan_release(PbPlan* plan);
int pb_launch_gather(pgh_graph_s* g, const float* xg, const LoopState* state, const FixView& fix);
template <int MODE>
int pb_launch_finish(pgh_graph_s* g, const RowSums& rs, const EpiParams& ep, const LoopState* state, int* num_partials);
void pb_destroy(PbFormat& p);
int bsf_launch_partial(pgh_graph_s* g, const float* xg, const LoopState* state, int stage = 0);

void pb_set_residual(const ResParams* rp);   // the next pb_launch_finish<EPI_AXPBY> evaluates the residual in the kernel (ResParams)
// the next pb_launch_finish runs only its phase-1 / phase-2 items (PbView::phase; live: exchanged slots per block); phase 2 puts its
// partial sums behind phase 1's, and *num_partials of the phase-2 launch counts both
void pb_set_finish_phase(int phase, int live);
int  pb_pending_finish_phase();      // the phase the NEXT finish launch will run (0: the whole step in one launch)
// graph_dropout: the launches of the NEXT step's block partial sums and phase A multiply every entry by its mask factor (rate in
// [0, 1), seed: pgh_spmv_dropout's); cleared by bsf_clear_dropout.  bsf_ensure_edge_ids builds the entry -> CSR index words once.
bool bsf_dropout_usable(const pgh_graph_s* g);
int bsf_ensure_edge_ids(pgh_graph_s* g);
void bsf_set_dropout(double rate, uint64_t seed);
void bsf_clear_dropout();
DropView bsf_dropout_view(const int32_t* edge);
// the partitioned loop's fused scalars (pgh_spmv.hip; driven by pgh_dist.hip)
bool dist_can_fuse(const pgh_graph_s* g);
// where a partitioned run keeps this rank's slice of the next gather vector: packed for the exchange (BsfFormat::lg_*), 0 = by row
int dist_set_local_layout(pgh_graph_s* g, int live, int hot, int cold = -1);
int dist_prescale_packed(pgh_graph_s* g, const float* x_local, float* xg_local_out);
// pgh_dist_set_send_lists with the requested slots already on the device (the engine's loop receives them there)
int dist_set_send_lists_device(pgh_graph_s* g, const uint32_t* slots_dev, const int32_t* local_block, const int64_t* seg_offsets, int32_t segments);
int dist_aux_init(LoopAux* aux);
int dist_combine_fused(pgh_graph_s* g, const float* p_local, double alpha, float* y_local, float* xg_local_out, const float* x_prev,
                       const float* deg_local, double* state, LoopAux* aux, int step, int* num_partials);
int dist_fold_fused(double* state, double* red, int num_partials);
int dist_close_fused(double* state, LoopAux* aux, const double* red, int step, int check, int err_kind, double tol, int64_t n_global,
                     int use_quotient, double a, double b, unsigned long long* progress);
int dist_resume(double* state, unsigned long long* progress);
int dist_close_err(double* state, int kind, double tol, int64_t n_global, unsigned long long* progress);
int bsf_ensure_degrees(pgh_graph_s* g);       // BsfFormat::deg_int

template <int MODE>
int bsf_launch_combine(pgh_graph_s* g, const EpiParams& ep, const LoopState* state, int* num_partials);
// small graphs (pgh_bsf.hip, k_small_tail): block partial sums + ONE one-workgroup launch for fix-ups, epilogue, residual and close
bool bsf_small_tail_usable(const pgh_graph_s* g);
template <int MODE>
int bsf_launch_small(pgh_graph_s* g, const EpiParams& ep, const float* xg, const float* x_prev, const LoopState* state,
                     const PendingClose& pc);
template <int MODE>
int bsf_launch(pgh_graph_s* g, const EpiParams& ep, const float* xg, const LoopState* state, int* num_partials,
               hipEvent_t before_combine = nullptr);
int bsf_to_internal(pgh_graph_s* g, const float* src, float* dst, bool prescale, float hole);
bool bsf_can_bring_pair(const pgh_graph_s* g);
IsoTail iso_tail_of(const BsfFormat& f);
int iso_flag_release(pgh_graph_s* g);
// (in_norm < 0: the pass computes the L1 norm of v itself -- k_pair_scan sums |v| beside listing its non-zeros -- divides by it and
// leaves it in init_aux->in_norm; init_state / init_aux: the loop state of the run is started by the same launches)
int bsf_bring_pair(pgh_graph_s* g, const float* v, const float* ranks, float* v_int, float* y0, bool want_xg, float in_norm,
                   bool start_from_v, bool watch_iso = false, LoopState* init_state = nullptr, LoopAux* init_aux = nullptr,
                   bool* state_inited = nullptr, const float* pred_deg = nullptr);
bool bsf_can_norm_on_device(const pgh_graph_s* g);
int bsf_out_to_internal(pgh_graph_s* g, const float* src, float* dst, float hole);
int bsf_make_gather(pgh_graph_s* g, const float* y_int, const float* scale_int, float* xg_out = nullptr);   // xg_out: a caller's buffer in the layout of BsfFormat::xg
int bsf_to_original(pgh_graph_s* g, const float* src, float* dst, double factor);
int bsf_build(pgh_graph_s* g, const float* val, const int32_t* mult, const float* src_old, const float* dst_old, bool relabel,
              int force_blocks = 0, BsfFormat* target = nullptr);
int build_count_perm(const unsigned int* cnt, int64_t n, int B, int blk, int32_t* perm, int32_t* iperm, int64_t head = kDealHeadAll);
int bsf_auto_blocks(int64_t n_src);
void bsf_destroy(BsfFormat& f);
// pgh_bsf64.hip: the f64 route of the "chebyshev" recurrence on a blocked image of its own (pgh_graph_s::bsf64)
bool bsf64_usable(const pgh_graph_s* g);
int bsf64_ensure(pgh_graph_s* g);
int64_t bsf64_length(const pgh_graph_s* g);      // length of the loop's internal-space vectors (the gather vector: + 1)
int bsf64_bring(pgh_graph_s* g, const float* p, double c1, double* term, double* res, double* xg, bool keep_flag = false);
int bsf64_take(pgh_graph_s* g, const double* res, double factor, float* out);
int bsf64_take_col(pgh_graph_s* g, const double* vec, float* mat, int ld, int col);
int bsf64_step(pgh_graph_s* g, double a, double b, double c, const double* term, double* term_out, double* result, double* xg,
               int err_linf, const LoopState* state, double* partial_sum, double* partial_delta, int* num_partials, bool every_row = false,
               const double* row_w = nullptr, const double* src_w = nullptr);
int bsf64_walk_operands(pgh_graph_s* g, int mode, const float* p, const float* deg, const float* lam, double inv_norm, double* row_w,
                        double* src_w, double* term);
int bsf64_scale_by(pgh_graph_s* g, double* x, const double* w);
int finish_graph(pgh_graph_s* g);

}  // namespace pgh
