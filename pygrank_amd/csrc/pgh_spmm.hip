// Multi-seed propagation: Y = M^T X over a row-major [n, b] slab of b <= 64 personalization vectors.
//
// Reference counterpart: NodeRanking.propagate (pygrank/core/signals.py:225-226) and the tuner / sweep callers that
// re-run rank() once per vector on the same matrix; the reference never batches.  Here the adjacency
// is streamed ONCE per iteration for the whole batch.  Every column keeps its own L1 quotient, residual and stopping
// point (RecursiveGraphFilter._step + ConvergenceManager per column: abstract_filters.py:126-136, convergence.py:77-101),
// so column j of the batch equals the single-vector run of seed j.
//
// Layout: the blocked segment-flag format with ONE column block (a gathered row of X is b * 4 contiguous bytes, so the
// gather is already line-granular; relabelling keeps the hot rows of X together for the L2 / Infinity Cache).
// One wavefront walks one 512-entry tile: lane = batch column, the column words are wave-uniform (scalar loads),
// a set flag is a wave-uniform branch that closes the running f32 row sum into the [n, b] sum slab.  Cross-tile
// segments leave [tile][64] carries combined in a fixed order by k_mm_fixup.  k_mm_combine applies the PageRank
// epilogue per column and writes the next gather slab; k_mm_residual / k_mm_close keep the per-column loop state.
#include "pgh_kernels.h"

#include <vector>

using namespace pgh;

namespace {

#ifndef PGH_MM_UNROLL
#define PGH_MM_UNROLL 8
#endif
constexpr int kLanes = 64;
constexpr int kTileMM = 64 * PGH_BSF_IPT;        // same tile table as the single-vector layout

struct BatchState {
    double scale[kLanes];
    double err[kLanes];
    double sum[kLanes];
    int    done[kLanes];
    int    steps[kLanes];
    int    converged[kLanes];
    int    all_done;
    int    b;
};

struct MMView {
    const uint32_t* colf;
    const float*    val;
    const int32_t*  seg_row;
    const int4*     tile;
    float*          head;        // [num_tiles][64]
    float*          tail;        // [num_tiles][64]
    int             num_tiles;
};

template <bool HAS_VAL>
__global__ __launch_bounds__(WG) void k_mm_partial(MMView f, const float* __restrict__ xg, int ld, int b, float* __restrict__ sums,
                                                    const BatchState* __restrict__ state) {
    if (state != nullptr && state->all_done) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (WG / 64) + (threadIdx.x >> 6)));
    const int stride = gridDim.x * (WG / 64);
    const bool live = lane < b;
    for (int t = wave; t < f.num_tiles; t += stride) {
        const int4 ti = f.tile[t];
        const uint32_t* __restrict__ cw = f.colf + ti.x;
        const float* __restrict__ vw = HAS_VAL ? f.val + ti.x : nullptr;
        int cur = ti.z;                       // segment open when the tile starts
        bool opened = false;                  // a flag has been seen in this tile (wave-uniform)
        double acc = 0.0;                     // f64: a lane adds up to 512 terms serially (the single-vector kernel adds 8)
        constexpr int U = PGH_MM_UNROLL;                  // row gathers in flight per wavefront
        for (int e0 = 0; e0 < kTileMM; e0 += U) {
            uint32_t w[U];
            float x[U];
#pragma unroll
            for (int j = 0; j < U; ++j) w[j] = cw[e0 + j];                         // wave-uniform: scalar loads
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = live ? xg[(int64_t)(w[j] & 0x7fffffffu) * ld + lane] : 0.f;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (w[j] >> 31) {                                                   // wave-uniform branch: a row segment starts
                    if (!opened) {
                        f.head[(int64_t)t * kLanes + lane] = (float)acc;            // piece of the segment open at tile start
                        opened = true;
                    } else {
                        const int row = f.seg_row[cur];
                        if (row >= 0 && live) sums[(int64_t)row * ld + lane] = (float)acc;
                    }
                    acc = 0.0;
                    ++cur;
                }
                acc += (double)(HAS_VAL ? vw[e0 + j] * x[j] : x[j]);
            }
        }
        f.tail[(int64_t)t * kLanes + lane] = (float)acc;                            // piece of the segment still open
    }
}

// one wavefront per closing tile: fixed-order sum of the chain of tail carries + the head piece
__global__ __launch_bounds__(WG) void k_mm_fixup(MMView f, int ld, int b, float* __restrict__ sums, const BatchState* __restrict__ state) {
    if (state != nullptr && state->all_done) return;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    const int stride = gridDim.x * (WG / 64);
    for (int t = wave; t < f.num_tiles; t += stride) {
        const int4 ti = f.tile[t];
        if (ti.w < 0 || ti.z < 0) continue;
        const int row = f.seg_row[ti.z];
        if (row < 0) continue;
        double total = 0.0;
        for (int s = ti.w; s < t; ++s) total += (double)f.tail[(int64_t)s * kLanes + lane];
        total += (double)f.head[(int64_t)t * kLanes + lane];
        if (lane < b) sums[(int64_t)row * ld + lane] = (float)total;
    }
}

struct CombineParams {
    const float* sums;       // [n, ld] plain row sums (structural zeros never written)
    const float* dst_scale;  // [n] or null
    const float* src_scale;  // [n] or null
    const float* p;          // [n, ld] personalization (AXPBY) or null (PLAIN)
    const float* y_old;      // [n, ld] previous iterate (frozen columns copy it) or null
    float*       y;          // [n, ld]
    float*       xg_out;     // [n, ld] next gather slab (y * src_scale) or null
    double       alpha;
    int          plain;      // 1: y = dst * sum
};

__global__ __launch_bounds__(WG) void k_mm_combine(CombineParams c, int64_t n, int ld, int b, const BatchState* __restrict__ state,
                                                    double* __restrict__ partial_sum /* [grid][64] */) {
    __shared__ double s_red[WG / 64][kLanes];
    if (state != nullptr && state->all_done) return;
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int64_t wave = blockIdx.x * (int64_t)(WG / 64) + wave_in_wg;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64);
    const bool live = lane < b;
    const bool frozen = state != nullptr && live && state->done[lane] != 0;
    const float a = c.plain ? 1.f : (float)(c.alpha * (state != nullptr && live ? state->scale[lane] : 1.0));
    const float bc = (float)(1.0 - c.alpha);
    double colsum = 0.0;
    for (int64_t r = wave; r < n; r += stride) {
        if (!live) continue;
        const int64_t at = r * ld + lane;
        float y;
        if (frozen) {
            y = c.y_old[at];
        } else {
            float s = c.sums[at];
            if (c.dst_scale != nullptr) s *= c.dst_scale[r];
            y = a * s;
            if (!c.plain) y += bc * c.p[at];
        }
        c.y[at] = y;
        if (c.xg_out != nullptr) c.xg_out[at] = c.src_scale != nullptr ? y * c.src_scale[r] : y;
        colsum += (double)y;
    }
    s_red[wave_in_wg][lane] = colsum;
    __syncthreads();
    if (wave_in_wg == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WG / 64; ++w) t += s_red[w][lane];
        partial_sum[(int64_t)blockIdx.x * kLanes + lane] = t;
    }
}

// per-column fold of the block partials: lane = column, the WG / 64 wavefronts take interleaved rows of the partial table
// and are combined in wavefront order (fixed summation order)
__global__ __launch_bounds__(WG) void k_mm_fold(const double* __restrict__ partials, int count, int linf, double* __restrict__ out) {
    __shared__ double s_red[WG / 64][kLanes];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc = 0.0;
    for (int i = w; i < count; i += WG / 64) {
        const double v = partials[(int64_t)i * kLanes + lane];
        acc = linf ? fmax(acc, v) : acc + v;
    }
    s_red[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        double t = s_red[0][lane];
#pragma unroll
        for (int k = 1; k < WG / 64; ++k) t = linf ? fmax(t, s_red[k][lane]) : t + s_red[k][lane];
        out[lane] = t;
    }
}

__global__ __launch_bounds__(WG) void k_mm_residual(const float* __restrict__ y, const float* __restrict__ y_old, int64_t n, int ld, int b,
                                                     int use_quotient, int linf, const BatchState* __restrict__ state,
                                                     double* __restrict__ partial_res) {
    __shared__ double s_red[WG / 64][kLanes];
    if (state->all_done) return;
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int64_t wave = blockIdx.x * (int64_t)(WG / 64) + wave_in_wg;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64);
    const bool live = lane < b;
    const double S = live ? state->sum[lane] : 1.0;
    const double inv = use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;
    const double scale = live ? state->scale[lane] : 1.0;
    double acc = 0.0;
    if (live && !state->done[lane]) {
        for (int64_t r = wave; r < n; r += stride) {
            const double d = fabs((double)y[r * ld + lane] * inv - (double)y_old[r * ld + lane] * scale);
            acc = linf ? fmax(acc, d) : acc + d;
        }
    }
    s_red[wave_in_wg][lane] = acc;
    __syncthreads();
    if (wave_in_wg == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WG / 64; ++w) t = linf ? fmax(t, s_red[w][lane]) : t + s_red[w][lane];
        partial_res[(int64_t)blockIdx.x * kLanes + lane] = t;
    }
}

// closes a batched step: per-column quotient, step count and ConvergenceManager check (convergence.py:96-101)
__global__ void k_mm_close(BatchState* __restrict__ state, int use_quotient, int check, int err_kind, double tol, int64_t n_orig) {
    const int lane = threadIdx.x;
    const bool live = lane < state->b;
    int done = 1;
    if (live && !state->done[lane]) {
        const double S = state->sum[lane];
        state->scale[lane] = use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;
        state->steps[lane] += 1;
        if (check) {
            double e = state->err[lane];
            if (err_kind == PGH_ERR_MABS) e /= (double)n_orig;
            state->err[lane] = e;
            if (e <= tol) {
                state->done[lane] = 1;
                state->converged[lane] = 1;
            }
        }
        done = state->done[lane];
    }
    const unsigned long long all = __ballot(done != 0);
    if (lane == 0) state->all_done = (all == ~0ULL) ? 1 : 0;
}

__global__ void k_mm_state_init(BatchState* state, int b) {
    const int lane = threadIdx.x;
    state->scale[lane] = 1.0;
    state->err[lane] = 0.0;
    state->sum[lane] = 0.0;
    state->done[lane] = lane < b ? 0 : 1;
    state->steps[lane] = 0;
    state->converged[lane] = 0;
    if (lane == 0) {
        state->all_done = 0;
        state->b = b;
    }
}

// caller-space slab -> internal (relabelled) slab, optional per-row scale; holes zero
__global__ void k_mm_permute_in(const float* __restrict__ src, const int32_t* __restrict__ perm, const float* __restrict__ row_scale,
                                int64_t n_int, int64_t n_valid, int ld, float* __restrict__ dst) {
    const int64_t total = n_int * ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int col = (int)(i - r * ld);
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        float v = o >= 0 ? src[o * ld + col] : 0.f;
        if (row_scale) v *= row_scale[r];
        dst[i] = v;
    }
}

__global__ void k_mm_permute_out(const float* __restrict__ src, const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid, int ld,
                                 const double* __restrict__ col_factor, float* __restrict__ dst) {
    const int64_t total = n_int * ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int col = (int)(i - r * ld);
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        if (o >= 0) dst[o * ld + col] = src[i] * (col_factor ? (float)col_factor[col] : 1.f);
    }
}

inline int blocks_for(int64_t n, int cap_mult = 16) {
    int64_t blocks = (n + WG - 1) / WG;
    const int64_t cap = (int64_t)rt().num_cus * cap_mult;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

struct DevBytes {          // slab work buffers from the runtime's stream-ordered pool (eight 2 GB slabs per 64-seed batch)
    void* p = nullptr;
    ~DevBytes() {
        if (p) pool_free(p);
    }
    int alloc(size_t bytes) { return pool_alloc(bytes > 0 ? bytes : 1, &p); }
    template <typename T>
    T* as() { return static_cast<T*>(p); }
};

int ensure_mm_layout(pgh_graph_s* g) {
    if (g->bsf_mm.enabled) return 0;
    PGH_CHECK(g->n_rows == g->n_cols, "the multi-seed path needs a square matrix");
    const bool valfree = g->keep_mult != nullptr;
    PGH_TRY(bsf_build(g, valfree ? nullptr : g->val, g->keep_mult, g->keep_src, g->keep_dst, true, 1, &g->bsf_mm));
    BsfFormat& f = g->bsf_mm;
    // [num_tiles][64] f32 carries live in the (otherwise unused) part / xg slots of the batch layout
    PGH_HIP(hipMalloc(&f.part, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes));
    PGH_HIP(hipMalloc(&f.xg, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes));
    PGH_HIP(hipMemsetAsync(f.part, 0, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes, rt().stream));
    PGH_HIP(hipMemsetAsync(f.xg, 0, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes, rt().stream));
    return 0;
}

MMView mm_view(const BsfFormat& f) {
    MMView v;
    v.colf = f.colf;
    v.val = f.val;
    v.seg_row = f.seg_row;
    v.tile = f.tile;
    v.head = f.part;
    v.tail = f.xg;
    v.num_tiles = f.num_tiles;
    return v;
}

// sums <- M^T-times-gather-slab (plain row sums in the internal id space); sums must have its structural zeros in place
int mm_partial(pgh_graph_s* g, const float* xg, int ld, int b, float* sums, const BatchState* state) {
    Runtime& r = rt();
    const BsfFormat& f = g->bsf_mm;
    const MMView v = mm_view(f);
    const int grid = r.num_cus * 8;
    {
        ProfScope prof(PGH_K_SPMM);
        if (f.val) k_mm_partial<true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state);
        else k_mm_partial<false><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state);
    }
    {
        ProfScope prof(PGH_K_FIXUP);
        k_mm_fixup<<<blocks_for((int64_t)f.num_tiles * 64, 64), WG, 0, r.stream>>>(v, ld, b, sums, state);
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

int combine_grid() { return rt().num_cus * 8; }

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" int pgh_spmm(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y) {
    PGH_CHECK(g && x && y, "pgh_spmm: null argument");
    PGH_CHECK(x->n == g->n_rows && y->n == g->n_cols && x->b == y->b, "pgh_spmm: shape mismatch");
    PGH_CHECK(x->b >= 1 && x->b <= kLanes, "pgh_spmm: the batch width must be in [1, 64]");
    PGH_CHECK(x->data != y->data, "pgh_spmm: conv must be pure (output aliases input)");
    if (g->n_cols == 0) return 0;
    PGH_TRY(ensure_mm_layout(g));
    Runtime& r = rt();
    const BsfFormat& f = g->bsf_mm;
    const int ld = x->b, b = x->b;
    const int64_t n_int = f.n_out;
    DevBytes xg, sums, yint, partial;
    PGH_TRY(xg.alloc(sizeof(float) * (size_t)n_int * ld));
    PGH_TRY(sums.alloc(sizeof(float) * (size_t)n_int * ld));
    PGH_TRY(yint.alloc(sizeof(float) * (size_t)n_int * ld));
    PGH_TRY(partial.alloc(sizeof(double) * (size_t)combine_grid() * kLanes));
    PGH_HIP(hipMemsetAsync(sums.p, 0, sizeof(float) * (size_t)n_int * ld, r.stream));
    k_mm_permute_in<<<blocks_for(n_int * ld), WG, 0, r.stream>>>(x->data, f.perm, f.src_scale, n_int, g->n_rows, ld, xg.as<float>());
    PGH_TRY(mm_partial(g, xg.as<float>(), ld, b, sums.as<float>(), nullptr));
    CombineParams c{};
    c.sums = sums.as<float>();
    c.dst_scale = f.dst_scale;
    c.y = yint.as<float>();
    c.plain = 1;
    {
        ProfScope prof(PGH_K_COMBINE);
        k_mm_combine<<<combine_grid(), WG, 0, r.stream>>>(c, n_int, ld, b, nullptr, partial.as<double>());
    }
    k_mm_permute_out<<<blocks_for(n_int * ld), WG, 0, r.stream>>>(yint.as<float>(), f.perm, n_int, g->n_cols, ld, nullptr, y->data);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    return 0;
}

// Batched PageRank: b independent runs of PageRank(alpha) with the SAME ConvergenceManager settings; column j stops
// at its own iteration (frozen afterwards), exactly as b calls of pgh_ppr_run would.
extern "C" int pgh_ppr_run_batch(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales,
                                 pgh_loop_result* results) {
    PGH_CHECK(g && p && ranks && cfg && results, "pgh_ppr_run_batch: null argument");
    PGH_CHECK(g->n_rows == g->n_cols && p->n == g->n_cols && ranks->n == g->n_cols && p->b == ranks->b, "pgh_ppr_run_batch: shape mismatch");
    PGH_CHECK(p->b >= 1 && p->b <= kLanes, "pgh_ppr_run_batch: the batch width must be in [1, 64]");
    PGH_CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    PGH_TRY(ensure_mm_layout(g));
    Runtime& r = rt();
    const BsfFormat& f = g->bsf_mm;
    const int ld = p->b, b = p->b;
    const int64_t n = g->n_cols, n_int = f.n_out;
    const size_t slab = sizeof(float) * (size_t)n_int * ld;
    DevBytes pint, xg, sums, y0, y1, partial, state_mem, factors;
    PGH_TRY(pint.alloc(slab));
    PGH_TRY(xg.alloc(slab));
    PGH_TRY(sums.alloc(slab));
    PGH_TRY(y0.alloc(slab));
    PGH_TRY(y1.alloc(slab));
    const int cgrid = combine_grid();
    PGH_TRY(partial.alloc(sizeof(double) * (size_t)cgrid * kLanes));
    PGH_TRY(state_mem.alloc(sizeof(BatchState)));
    PGH_TRY(factors.alloc(sizeof(double) * kLanes));
    BatchState* state = state_mem.as<BatchState>();
    hipEvent_t ev_a, ev_b;
    PGH_HIP(hipEventCreate(&ev_a));
    PGH_HIP(hipEventCreate(&ev_b));
    PGH_HIP(hipEventRecord(ev_a, r.stream));
    PGH_HIP(hipMemsetAsync(sums.p, 0, slab, r.stream));
    k_mm_state_init<<<1, kLanes, 0, r.stream>>>(state, b);
    k_mm_permute_in<<<blocks_for(n_int * ld), WG, 0, r.stream>>>(p->data, f.perm, nullptr, n_int, n, ld, pint.as<float>());
    const float* start = cfg->start_from_p ? p->data : ranks->data;          // abstract_filters.py:56 without warm_start
    k_mm_permute_in<<<blocks_for(n_int * ld), WG, 0, r.stream>>>(start, f.perm, nullptr, n_int, n, ld, y0.as<float>());
    k_mm_permute_in<<<blocks_for(n_int * ld), WG, 0, r.stream>>>(start, f.perm, f.src_scale, n_int, n, ld, xg.as<float>());
    float* buf[2] = {y0.as<float>(), y1.as<float>()};
    const int linf = cfg->err_kind == PGH_ERR_LINF;
    const int max_steps = cfg->max_iters - 1 > 0 ? cfg->max_iters - 1 : 0;
    BatchState host_state;
    memset(&host_state, 0, sizeof(host_state));
    bool done = false;
    int enq = 0;
    while (!done && enq < max_steps) {
        const int upto = (enq + 4 < max_steps) ? enq + 4 : max_steps;
        for (; enq < upto; ++enq) {
            const int k = enq + 1;
            float* yin = buf[(k - 1) & 1];
            float* yout = buf[k & 1];
            PGH_TRY(mm_partial(g, xg.as<float>(), ld, b, sums.as<float>(), state));
            CombineParams c{};
            c.sums = sums.as<float>();
            c.dst_scale = f.dst_scale;
            c.src_scale = f.src_scale;
            c.p = pint.as<float>();
            c.y_old = yin;
            c.y = yout;
            c.xg_out = xg.as<float>();
            c.alpha = cfg->alpha;
            {
                ProfScope prof(PGH_K_COMBINE);
                k_mm_combine<<<cgrid, WG, 0, r.stream>>>(c, n_int, ld, b, state, partial.as<double>());
            }
            k_mm_fold<<<1, WG, 0, r.stream>>>(partial.as<double>(), cgrid, 0, reinterpret_cast<double*>(state) + 2 * kLanes);   // -> state.sum
            const int it = k + 1;
            const int check = (cfg->err_kind != PGH_ERR_ITERS) && (it < cfg->max_iters) && (it % cfg->end_modulo == 0);
            if (check) {
                ProfScope prof(PGH_K_RESIDUAL);
                k_mm_residual<<<cgrid, WG, 0, r.stream>>>(yout, yin, n_int, ld, b, cfg->use_quotient, linf, state, partial.as<double>());
                k_mm_fold<<<1, WG, 0, r.stream>>>(partial.as<double>(), cgrid, linf, reinterpret_cast<double*>(state) + kLanes);     // -> state.err
            }
            k_mm_close<<<1, kLanes, 0, r.stream>>>(state, cfg->use_quotient, check, cfg->err_kind, cfg->tol, n);
        }
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipMemcpyAsync(&host_state, state, sizeof(BatchState), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        done = host_state.all_done != 0;
    }
    PGH_HIP(hipMemcpyAsync(&host_state, state, sizeof(BatchState), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    // the iterate of a column that stopped early was copied forward unchanged by every later executed step
    double h_factors[kLanes];
    for (int j = 0; j < kLanes; ++j) h_factors[j] = j < b ? host_state.scale[j] * (out_scales ? out_scales[j] : cfg->out_scale) : 1.0;
    PGH_HIP(hipMemcpyAsync(factors.p, h_factors, sizeof(h_factors), hipMemcpyHostToDevice, r.stream));
    int executed = 0;                                    // steps that ran before every column had stopped
    for (int j = 0; j < b; ++j) executed = host_state.steps[j] > executed ? host_state.steps[j] : executed;
    k_mm_permute_out<<<blocks_for(n_int * ld), WG, 0, r.stream>>>(buf[executed & 1], f.perm, n_int, n, ld, factors.as<double>(), ranks->data);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipEventRecord(ev_b, r.stream));
    PGH_HIP(hipEventSynchronize(ev_b));
    float ms = 0.f;
    PGH_HIP(hipEventElapsedTime(&ms, ev_a, ev_b));
    (void)hipEventDestroy(ev_a);
    (void)hipEventDestroy(ev_b);
    for (int j = 0; j < b; ++j) {
        memset(&results[j], 0, sizeof(pgh_loop_result));
        results[j].iterations = host_state.steps[j] + 1;
        results[j].converged = host_state.converged[j];
        results[j].spmv_count = host_state.steps[j];
        results[j].last_error = host_state.err[j];
        results[j].loop_ms = (double)ms;
    }
    return 0;
}
