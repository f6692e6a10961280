// Small graphs: the whole PageRank run as ONE persistent multi-workgroup kernel.
//
// Reference counterpart: GraphFilter.rank's loop (abstract_filters.py:58-62) with PageRank._formula (adhoc.py:34-36),
// RecursiveGraphFilter._step's L1 quotient (abstract_filters.py:126-136) and ConvergenceManager (convergence.py:77-101).
//
// Why: below ~0.5 M edges an iteration of the blocked loop is 4-5 dependent launches of a few microseconds each -- 20-27 us
// per iteration, 211 / 245 us per run at RMAT scale 10 / 14 (profiles/r02/small_window_sweep.log), the BASELINE.json configs[0]
// graph (10 K nodes, 80 K edges) likewise -- while the arithmetic of an iteration is well under a microsecond of the chip.
// A single workgroup for the whole loop was measured in round 1 and lost (one CU's load latency per dependent phase).  This
// kernel keeps EVERY CU on the loop: num_cus workgroups x 256 threads, resident together, a wavefront per row of CSR(M^T) in
// the caller's id space (no relabelling, so no permute passes either), two grid-wide barriers per iteration (after the
// step, after the residual), the per-workgroup partial sums folded by every workgroup in the same order -- all of them reach
// the same verdict, deterministic, atomic-free arithmetic.  One launch and one 32-byte read-back per run.
#include "pgh_kernels.h"

#include <cstdlib>

using namespace pgh;

namespace {

constexpr int kSmallThreads = 256;
constexpr int kSmallMaxGrid = 256;

struct SmallState {          // the run's outcome, read back by the host
    double scale;
    double err;
    int    steps;
    int    converged;
    int    pad[2];
};

struct SmallArgs {
    const int32_t* rowptr;   // CSR(M^T), caller ids
    const int32_t* col;
    const float*   val;
    const float*   p;        // personalization (caller's; divided by in_norm here)
    float*         ranks;    // in: start iterate unless start_from_p; out: final ranks
    float*         pn;       // [n] work: p / in_norm
    float*         buf0;     // [n] work: iterates
    float*         buf1;
    double*        part_sum; // [grid]
    double*        part_res; // [grid]
    unsigned int*  bar;      // [2] arrival counter, generation
    SmallState*    out;
    double         alpha, tol, out_scale;
    float          in_norm;
    int            n, max_iters, end_modulo, use_quotient, err_kind, start_from_p;
};

// grid-wide barrier for workgroups that are all resident: arrival counter + generation word, agent-scope release /
// acquire around it (the eight XCDs have L2s of their own: what a workgroup wrote before the barrier must be written back,
// what it reads after must not come from a stale line)
__device__ __forceinline__ void grid_barrier(unsigned int* bar, unsigned int nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned int gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// every workgroup folds the same partials in the same order (thread t takes t, t + 256, ...; wavefront shuffles; the four
// wavefronts in order): bitwise the same result everywhere
__device__ __forceinline__ double fold_all(const double* __restrict__ partials, int count, int linf, double* s4) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += kSmallThreads) {
        const double v = __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc = linf ? fmax(acc, v) : acc + v;
    }
    acc = linf ? wave_reduce_max(acc) : wave_reduce_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = acc;
    __syncthreads();
    double r = s4[0];
#pragma unroll
    for (int w = 1; w < kSmallThreads / 64; ++w) r = linf ? fmax(r, s4[w]) : r + s4[w];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kSmallThreads) void k_small_ppr(SmallArgs a) {
    __shared__ double s4[4];
    __shared__ double s_part[kSmallThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave_in = tid >> 6;
    const int nblocks = gridDim.x;
    const int gthread = blockIdx.x * kSmallThreads + tid, gthreads = nblocks * kSmallThreads;
    const int gwave = blockIdx.x * (kSmallThreads / 64) + wave_in, gwaves = nblocks * (kSmallThreads / 64);
    const int n = a.n;
    const int linf = a.err_kind == PGH_ERR_LINF;
    // ---- prologue of GraphFilter.rank (abstract_filters.py:55-56): p / norm, start iterate
    for (int i = gthread; i < n; i += gthreads) {
        const float pv = a.in_norm != 1.f ? a.p[i] / a.in_norm : a.p[i];
        a.pn[i] = pv;
        a.buf0[i] = a.start_from_p ? pv : a.ranks[i];
    }
    grid_barrier(a.bar, nblocks);
    double scale = 1.0, err = 0.0;
    int steps = 0, converged = 0;
    const float bf = (float)(1.0 - a.alpha);
    const int max_steps = a.max_iters - 1 > 0 ? a.max_iters - 1 : 0;
    for (int k = 1; k <= max_steps; ++k) {
        const float* __restrict__ x = (k & 1) ? a.buf0 : a.buf1;
        float* __restrict__ y = (k & 1) ? a.buf1 : a.buf0;
        const float a_eff = (float)(a.alpha * scale);              // the pending quotient rides in the factor (lazy, as the big loops)
        // ---- step: a wavefront per row; f32 products, f64 row sums (the row-major kernel's arithmetic)
        double sum_y = 0.0;
        for (int row = gwave; row < n; row += gwaves) {
            const int lo = a.rowptr[row], hi = a.rowptr[row + 1];
            double acc = 0.0;
            for (int e = lo + lane; e < hi; e += 64) acc += (double)(a.val[e] * x[a.col[e]]);
            acc = wave_reduce_sum(acc);
            if (lane == 0) {
                float v = a_eff * (float)acc;
                v += bf * a.pn[row];
                y[row] = v;
                sum_y += (double)v;
            }
        }
        if (lane == 0) s_part[wave_in] = sum_y;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < kSmallThreads / 64; ++w) t += s_part[w];
            __hip_atomic_store(a.part_sum + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier(a.bar, nblocks);
        const double S = fold_all(a.part_sum, nblocks, 0, s4);
        const double inv = a.use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;      // abstract_filters.py:133-134
        const int it = k + 1;
        const int check = a.err_kind != PGH_ERR_ITERS && it < a.max_iters && it % a.end_modulo == 0;
        if (check) {
            // ---- residual |y * inv - x * scale| (supervised.py:93-138) over this workgroup's share of the rows
            double r = 0.0;
            for (int i = gthread; i < n; i += gthreads) {
                const double d = fabs((double)y[i] * inv - (double)x[i] * scale);
                r = linf ? fmax(r, d) : r + d;
            }
            r = linf ? wave_reduce_max(r) : wave_reduce_sum(r);
            if (lane == 0) s_part[wave_in] = r;
            __syncthreads();
            if (tid == 0) {
                double t = s_part[0];
                for (int w = 1; w < kSmallThreads / 64; ++w) t = linf ? fmax(t, s_part[w]) : t + s_part[w];
                __hip_atomic_store(a.part_res + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            grid_barrier(a.bar, nblocks);
            err = fold_all(a.part_res, nblocks, linf, s4);
            if (a.err_kind == PGH_ERR_MABS) err /= (double)n;
        }
        scale = inv;
        steps = k;
        if (check && err <= a.tol) {                               // convergence.py:96-101: the same verdict in every workgroup
            converged = 1;
            break;
        }
    }
    // ---- the quotient still pending and the preserve_norm factor (abstract_filters.py:63-64)
    const float* __restrict__ fin = (steps & 1) ? a.buf1 : a.buf0;
    const float f = (float)(scale * a.out_scale);
    for (int i = gthread; i < n; i += gthreads) a.ranks[i] = fin[i] * f;
    if (blockIdx.x == 0 && tid == 0) {
        a.out->scale = scale;
        a.out->err = err;
        a.out->steps = steps;
        a.out->converged = converged;
    }
}

struct SmallBuffers {
    double*       parts = nullptr;      // [2 * kSmallMaxGrid]
    unsigned int* bar = nullptr;        // [2]
    SmallState*   out = nullptr;        // device
    SmallState*   out_host = nullptr;   // pinned
    hipEvent_t    ev_a = nullptr, ev_b = nullptr;
};
SmallBuffers g_small;

int ensure_small_buffers() {
    if (g_small.parts != nullptr) return 0;
    PGH_HIP(hipMalloc(&g_small.parts, sizeof(double) * 2 * kSmallMaxGrid));
    PGH_HIP(hipMalloc(&g_small.bar, sizeof(unsigned int) * 2));
    PGH_HIP(hipMemset(g_small.bar, 0, sizeof(unsigned int) * 2));
    PGH_HIP(hipMalloc(&g_small.out, sizeof(SmallState)));
    PGH_HIP(hipHostMalloc(&g_small.out_host, sizeof(SmallState), hipHostMallocDefault));
    PGH_HIP(hipEventCreate(&g_small.ev_a));
    PGH_HIP(hipEventCreate(&g_small.ev_b));
    return 0;
}

}  // namespace

namespace pgh {

// graphs for which one wavefront per row across the chip beats 4-5 launches per iteration (PGH_SMALL=0 switches the path off,
// PGH_SMALL_NNZ moves the line)
bool small_ppr_usable(const pgh_graph_s* g, const pgh_loop_cfg* cfg) {
    static const int off = getenv("PGH_SMALL") != nullptr && atoi(getenv("PGH_SMALL")) == 0;
    static const int64_t max_nnz = getenv("PGH_SMALL_NNZ") != nullptr ? atoll(getenv("PGH_SMALL_NNZ")) : 600000;
    if (off || g->n_rows != g->n_cols || g->rowptr == nullptr || g->col == nullptr || g->val == nullptr) return false;
    if (g->n_cols < 1 || g->n_cols > 65536 || g->nnz > max_nnz) return false;
    return cfg->end_modulo >= 1 && cfg->max_iters >= 1;
}

int small_ppr_run(pgh_graph_s* g, const float* p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res) {
    Runtime& r = rt();
    PGH_TRY(ensure_small_buffers());
    const int n = (int)g->n_cols;
    float *pn = nullptr, *b0 = nullptr, *b1 = nullptr;
    PGH_TRY(pool_alloc(sizeof(float) * (size_t)n, (void**)&pn));
    PGH_TRY(pool_alloc(sizeof(float) * (size_t)n, (void**)&b0));
    PGH_TRY(pool_alloc(sizeof(float) * (size_t)n, (void**)&b1));
    SmallArgs a;
    a.rowptr = g->rowptr;
    a.col = g->col;
    a.val = g->val;
    a.p = p;
    a.ranks = ranks->data;
    a.pn = pn;
    a.buf0 = b0;
    a.buf1 = b1;
    a.part_sum = g_small.parts;
    a.part_res = g_small.parts + kSmallMaxGrid;
    a.bar = g_small.bar;
    a.out = g_small.out;
    a.alpha = cfg->alpha;
    a.tol = cfg->tol;
    a.out_scale = cfg->out_scale;
    a.in_norm = cfg->in_norm != 0.0 ? (float)cfg->in_norm : 1.f;
    a.n = n;
    a.max_iters = cfg->max_iters;
    a.end_modulo = cfg->end_modulo;
    a.use_quotient = cfg->use_quotient;
    a.err_kind = cfg->err_kind;
    a.start_from_p = cfg->start_from_p != 0;
    // every workgroup must be resident for the barrier: one per CU at most, no more than there are rows to hand out
    int grid = r.num_cus < kSmallMaxGrid ? r.num_cus : kSmallMaxGrid;
    const int useful = (n + 3) / 4;                    // four wavefronts = four rows per workgroup and round
    if (grid > useful) grid = useful;
    if (grid < 1) grid = 1;
    PGH_HIP(hipEventRecord(g_small.ev_a, r.stream));
    {
        ProfScope prof(PGH_K_SPMV);
        k_small_ppr<<<grid, kSmallThreads, 0, r.stream>>>(a);
    }
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipEventRecord(g_small.ev_b, r.stream));
    PGH_HIP(hipMemcpyAsync(g_small.out_host, g_small.out, sizeof(SmallState), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    pool_free(pn);
    pool_free(b0);
    pool_free(b1);
    float ms = 0.f;
    PGH_HIP(hipEventElapsedTime(&ms, g_small.ev_a, g_small.ev_b));
    memset(res, 0, sizeof(*res));
    res->iterations = g_small.out_host->steps + 1;            // ConvergenceManager.iteration at loop exit
    res->converged = g_small.out_host->converged;
    res->spmv_count = g_small.out_host->steps;
    res->last_error = g_small.out_host->err;
    res->loop_ms = (double)ms;
    return 0;
}

}  // namespace pgh
