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
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTileMM = 64 * PGH_BSF_IPT;        // same tile table as the single-vector layout

struct BatchState {
    double scale[kLanes];        // quotient of the current iterate (1 / sum(y_k); 1 without the quotient)
    double err[kLanes];
    double sum[kLanes];
    double pred_inv[kLanes];     // PREDICTED quotient of the step being written (in-kernel residual, see k_mm_step)
    double pred_raw[kLanes];     // the uncorrected prediction it came from
    double sum_p[kLanes];        // sum of the column's personalization (measured by the first step)
    int    done[kLanes];
    int    steps[kLanes];
    int    converged[kLanes];
    // the four words the host polls after every step (copied to a pinned ring): every column has stopped / the in-kernel residual
    // of step `steps` could not vouch for its verdict on some column and the step waits for the separate kernel / executed steps
    int    all_done;
    int    paused;
    int    executed;
    int    b;
};

#ifndef PGH_MM_RES_U
#define PGH_MM_RES_U 8
#endif
#ifndef PGH_MM_COMB_U
#define PGH_MM_COMB_U 2
#endif
struct MMView {
    const uint32_t* colf;
    const float*    val;
    const int32_t*  seg_row;
    const int32_t*  close;       // [num_entries] closing row of every entry (k_mm_close_rows)
    const int4*     tile;
    float*          head;        // [num_tiles][64]
    float*          tail;        // [num_tiles][64]
    int             num_tiles;
};

// closing row of every entry of the multi-seed stream (built once, ensure_mm_layout): -1 = the entry's segment goes on,
// -2 = it ends the piece of the segment that was open when the tile started (-> head carry), otherwise the output row
// of the segment that ends here.  A segment still open at the end of a tile leaves a tail carry.  With this word the
// kernel needs no running segment counter and no dependent seg_row lookups.
__global__ __launch_bounds__(WG) void k_mm_close_rows(const uint32_t* __restrict__ colf, const int4* __restrict__ tile,
                                                       const int32_t* __restrict__ seg_row, int num_tiles, int32_t* __restrict__ close) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    const int stride = gridDim.x * (WG / 64);
    for (int t = wave; t < num_tiles; t += stride) {
        const int4 ti = tile[t];
        const int64_t base = ti.x;
        int before = 0;                                   // flags of the tile seen so far
        for (int c = 0; c < kTileMM / 64; ++c) {
            const int e = c * 64 + lane;
            const bool flag = (colf[base + e] >> 31) != 0;
            const unsigned long long mask = __ballot(flag);
            const int k = before + __popcll(mask & (~0ULL >> (63 - lane)));       // flags at positions <= e
            const bool next_flag = e + 1 < kTileMM && (colf[base + e + 1] >> 31) != 0;
            int out = -1;
            if (next_flag) {
                if (k == 0) out = -2;
                else {
                    const int row = seg_row[ti.z + k];
                    out = row >= 0 ? row : -1;
                }
            }
            close[base + e] = out;
            before += __popcll(mask);
        }
    }
}

// rows of M^T that hold at least one entry (static per graph): every closing row of the stream + the rows completed by the
// cross-tile fix-ups.  A row WITHOUT entries whose personalization row is zero stays zero in every iterate of every
// column: the batch loop neither reads nor writes it (k_mm_combine, k_mm_residual) -- 55 % of the rows of the RMAT bench graph.
__global__ __launch_bounds__(WG) void k_mm_mark_rows(const int32_t* __restrict__ close, int64_t num_entries, const int4* __restrict__ tile,
                                                      const int32_t* __restrict__ seg_row, int num_tiles, uint8_t* __restrict__ has) {
    const int64_t stride = (int64_t)gridDim.x * WG;
    for (int64_t e = blockIdx.x * (int64_t)WG + threadIdx.x; e < num_entries; e += stride) {
        const int c = close[e];
        if (c >= 0) has[c] = 1;
    }
    for (int64_t t = blockIdx.x * (int64_t)WG + threadIdx.x; t < num_tiles; t += stride) {
        const int4 ti = tile[t];
        if (ti.w < 0 || ti.z < 0) continue;
        const int row = seg_row[ti.z];
        if (row >= 0) has[row] = 1;
    }
}

// graph_dropout in the batch kernel: the mask of pgh_spmv_dropout -- a hash of (seed, index of the entry in CSR(M^T) order) --
// evaluated per stream entry; the index rides in a word of its own (BsfFormat::mm_edge) that only dropout launches read
struct MMDrop {
    const int32_t* edge;
    uint64_t       seed;
    uint32_t       threshold;    // floor(rate * 2^32)
    float          keep_scale;   // 1 / (1 - rate)
};

// index in CSR(M^T) order of every entry of the multi-seed stream (built once, on the first dropout launch): the entry's
// row is that of its segment, its source is the column word; both go back to the caller's ids and the source is looked up
// in the row (sorted indices: binary search)
__global__ __launch_bounds__(WG) void k_mm_edge_ids(const uint32_t* __restrict__ colf, const int4* __restrict__ tile,
                                                     const int32_t* __restrict__ seg_row, int num_tiles, const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     int32_t* __restrict__ edge) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    const int stride = gridDim.x * (WG / 64);
    for (int t = wave; t < num_tiles; t += stride) {
        const int4 ti = tile[t];
        const int64_t base = ti.x;
        int before = 0;
        for (int c = 0; c < kTileMM / 64; ++c) {
            const int e = c * 64 + lane;
            const uint32_t w = colf[base + e];
            const unsigned long long mask = __ballot((w >> 31) != 0);
            const int k = before + __popcll(mask & (~0ULL >> (63 - lane)));       // flags at positions <= e
            const int seg = ti.z + k;                                             // ti.z = the segment open at the tile start
            const int row_new = seg >= 0 ? seg_row[seg] : -1;
            int out = -1;
            if (row_new >= 0) {
                const int row_old = perm ? perm[row_new] : row_new;
                const int col_old = perm ? perm[w & 0x7fffffffu] : (int)(w & 0x7fffffffu);
                int lo = rowptr[row_old], hi = rowptr[row_old + 1];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (col[mid] < col_old) lo = mid + 1; else hi = mid;
                }
                out = lo;
            }
            edge[base + e] = out;
            before += __popcll(mask);
        }
    }
}

// One wavefront = four groups of 16 lanes; a group walks its own 512-entry tile and a lane holds 4 of the <= 64 batch
// columns, so ONE 16-byte load instruction of the wavefront fetches four 256-byte rows of the gather slab (the first
// version fetched one row per instruction with lane = column and kept 8 of them in flight between dependent scalar
// loads: 7.5 ms per pass at scale 23 whatever the batch width -- latency, not bandwidth).  The stream words of 16
// entries are loaded by the group's 16 lanes and handed round with ds_bpermute; rows are gathered 8 entries ahead of
// the sums that consume them (two register sets), stream words 16 entries ahead of the gathers.
//
// What bounds it now (profiles/r02/spmm_v2_*): of the 34.4 GB of rows gathered per pass at scale 23 / b = 64, 27.2 GB miss
// the L2 and cross the fabric at 6.2 TB/s -- the ceiling the same box gives any L2-missing read stream, whether the
// Infinity Cache or HBM serves it (tools/bw_probe: 5.3-6.2 TB/s re-reading 64-256 MB, 9 TB/s only below 32 MB).  An LRU
// L2 of 4 MB keeps a row only if it comes back within ~10 K gathers: 19 % of the gathers.  Loading the rows of all but
// the hottest sources with the streaming policy (nt) so that they would not evict the hot ones was measured and does
// not change the hit rate (8.9-9.5 ms per step against 8.7).
// Narrow batches: a row of <= 32 (<= 16) columns needs 8 (4) lanes, so the wavefront is cut into 8 (16) groups and one load
// instruction fetches 8 (16) rows; a lane then holds 2 (4) of the 16 stream words of its group's round.
// SPARSE (round 5): the rows of the gather slab that are all zeros are not fetched at all.  A PageRank batch starts from seed sets: the
// first iterate has a few thousand non-zero rows, the second the seeds' out-neighbours -- gathering them costs a 256-byte L2 miss per
// entry for nothing (adding +0 changes no sum, bit for bit).  One byte per row (SparseGate::map, written by whoever writes the slab)
// is looked up with the stream word, 16 entries ahead of the gather; the sources are hot-first, so the bytes of the often-referenced
// ones share a few lines.  The dense and the sparse form are BOTH launched for every step: each evaluates the same test on the
// device (non-zero rows of the slab against the rows that can be non-zero) and the one it does not select returns at once.
struct SparseGate {
    const uint8_t* map;      // [n] 1 = the row of the gather slab holds a non-zero (null: always dense)
    const int*     nz_rows;  // non-zero rows of the slab
    const int*     live_rows;// rows that take part in the run at all
};
__device__ __forceinline__ bool mm_take_sparse(const SparseGate& sg) {
    return sg.map != nullptr && (long long)(*sg.nz_rows) * 4 < (long long)(*sg.live_rows);
}
#ifndef PGH_MM_GATHER_BF
#define PGH_MM_GATHER_BF 1
#endif
template <bool HAS_VAL, int LPR, bool DROP = false, bool SPARSE = false>
__global__ __launch_bounds__(WG) void k_mm_partial(MMView f, const float* __restrict__ xg, int ld, int b, float* __restrict__ sums,
                                                    const BatchState* __restrict__ state, MMDrop drop = MMDrop{}, SparseGate sg = SparseGate{}) {
    if (state != nullptr && (state->all_done | state->paused)) return;
    if (mm_take_sparse(sg) != SPARSE) return;
    constexpr int G = 64 / LPR;                            // groups (tiles in flight) per wavefront
    constexpr int W = 16 / LPR;                            // stream words of a 16-entry round per lane
    const int lane = threadIdx.x & 63;
    const int l = lane & (LPR - 1);
    const int c4 = 4 * l;                                  // first of this lane's four columns
    const int pull = (lane & ~(LPR - 1)) << 2;             // ds_bpermute byte index of lane 0 of this group
    const bool live = c4 < b;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (WG / 64) + (threadIdx.x >> 6)));
    const int stride = gridDim.x * (WG / 64) * G;
    struct Words {
        uint32_t w[W];
        int      c[W];
        float    v[W];               // the entry's value; with DROP: times the mask's factor (0 or 1 / (1 - rate))
    };
    for (int t0 = wave * G; t0 < f.num_tiles; t0 += stride) {
        const int t = t0 + lane / LPR;
        const bool has = t < f.num_tiles;
        const int64_t base = (int64_t)(has ? t : f.num_tiles - 1) * kTileMM;      // tiles are full and consecutive
        Words w0, w1, w2;
        auto words = [&](int e16, Words& q) __attribute__((always_inline)) {
            const int64_t at = base + e16 + l * W;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                q.w[k] = __builtin_nontemporal_load(f.colf + at + k);
                if (SPARSE) q.w[k] |= (uint32_t)sg.map[q.w[k] & 0x3fffffffu] << 30;      // (ids stay below 2^28: bit 30 is free)
                const int cw = __builtin_nontemporal_load(f.close + at + k);      // (the tile index is clamped: unconditional, then a select)
                q.c[k] = has ? cw : -1;
                q.v[k] = HAS_VAL ? __builtin_nontemporal_load(f.val + at + k) : 1.f;
                if (DROP) {
                    const int edge = __builtin_nontemporal_load(drop.edge + at + k);
                    q.v[k] *= edge >= 0 ? dropout_factor(drop.seed, (uint64_t)edge, drop.threshold, drop.keep_scale) : 0.f;
                }
            }
        };
        // entry j of the round lives in word j % W of lane j / W of the group
        auto gather = [&](const Words& q, int half, f32x4 (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * half + j;
                const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(pull + 4 * (e / W), (int)q.w[e % W]);
                const uint32_t src = word & (SPARSE ? 0x3fffffffu : 0x7fffffffu);
                const bool fetch = SPARSE ? (live && (word & 0x40000000u) != 0u) : live;
#if PGH_MM_GATHER_BF
                if (!SPARSE) {
                    // (round 5: unconditional -- a lane beyond the batch's columns re-reads its group's first four; as `live ? load : 0` every
                    // gather sat under a branch and the wait before the sums was for ALL gathers, the eight just issued included)
                    // (no select behind it either -- it would wait for the load where it stands: such a lane's sums are never stored where
                    // anybody reads them)
                    x[j] = *reinterpret_cast<const f32x4*>(xg + (int64_t)src * ld + (live ? c4 : 0));
                } else      // (the sparse form keeps its skipped loads skipped: redirected to a zero row they cost 50 us of a batch step)
#endif
                x[j] = fetch ? *reinterpret_cast<const f32x4*>(xg + (int64_t)src * ld + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;     // f64: a lane adds up to 512 terms serially
        auto consume = [&](const Words& q, int half, const f32x4 (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * half + j;
                const int ends = __builtin_amdgcn_ds_bpermute(pull + 4 * (e / W), q.c[e % W]);
                f32x4 xv = x[j];
                if (HAS_VAL || DROP) xv *= __int_as_float(__builtin_amdgcn_ds_bpermute(pull + 4 * (e / W), __float_as_int(q.v[e % W])));
                a0 += (double)xv.x;
                a1 += (double)xv.y;
                a2 += (double)xv.z;
                a3 += (double)xv.w;
                if (ends != -1) {                          // uniform inside a group
                    const f32x4 s = {(float)a0, (float)a1, (float)a2, (float)a3};
                    if (ends == -2) *reinterpret_cast<f32x4*>(f.head + (int64_t)t * kLanes + c4) = s;
                    else if (live) __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(sums + (int64_t)ends * ld + c4));
                    a0 = a1 = a2 = a3 = 0.0;
                }
            }
        };
        f32x4 xa[8], xb[8];
        words(0, w0);
        words(16, w1);
        gather(w0, 0, xa);
        for (int mr = 0; mr < kTileMM / 16; ++mr) {
            gather(w0, 1, xb);
            consume(w0, 0, xa);
            words(min(16 * (mr + 2), kTileMM - 16), w2);
            if (mr + 1 < kTileMM / 16) gather(w1, 0, xa);
            consume(w0, 1, xb);
            w0 = w1;
            w1 = w2;
        }
        if (has) *reinterpret_cast<f32x4*>(f.tail + (int64_t)t * kLanes + c4) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
    }
}

// cross-tile segments: fixed-order sum of the chain of tail carries + the head piece.  A group of lanes_per_row(ld) lanes per closing
// tile, a float4 per lane (round 4: one wavefront per tile and a float per lane -- 181 us of dependent loads at scale 23 / b = 64)
__host__ __device__ inline int lanes_per_row(int ld) { return ld <= 16 ? 4 : (ld <= 32 ? 8 : 16); }
__global__ __launch_bounds__(WG) void k_mm_fixup(MMView f, int ld, int b, float* __restrict__ sums, const BatchState* __restrict__ state) {
    if (state != nullptr && (state->all_done | state->paused)) return;
    const int lpr = lanes_per_row(ld), per_wave = 64 / lpr;
    const int lane = threadIdx.x & 63, l = lane & (lpr - 1), c4 = 4 * l;
    const int first = (blockIdx.x * (WG / 64) + (threadIdx.x >> 6)) * per_wave + lane / lpr;
    const int stride = gridDim.x * (WG / 64) * per_wave;
    // the kernel is a chain of dependent loads (tile -> segment row -> carries): four tiles per lane group travel together
    constexpr int UF = 4;
    for (int t0 = first; t0 < f.num_tiles; t0 += stride * UF) {
        int4 ti[UF];
        int row[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int t = t0 + u * stride;
            ti[u] = t < f.num_tiles ? f.tile[t] : make_int4(0, 0, -1, -1);
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) row[u] = (ti[u].w >= 0 && ti[u].z >= 0) ? f.seg_row[ti[u].z] : -1;
        f32x4 head[UF], tail0[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int t = t0 + u * stride;
            const bool ok = row[u] >= 0 && c4 < ld;
            head[u] = ok ? *reinterpret_cast<const f32x4*>(f.head + (int64_t)t * kLanes + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            tail0[u] = (ok && ti[u].w < t) ? *reinterpret_cast<const f32x4*>(f.tail + (int64_t)ti[u].w * kLanes + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int t = t0 + u * stride;
            if (row[u] < 0 || c4 >= ld) continue;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int s = ti[u].w;
            if (s < t) {
                a0 = (double)tail0[u].x, a1 = (double)tail0[u].y, a2 = (double)tail0[u].z, a3 = (double)tail0[u].w;
                ++s;
            }
            // a hub row's chain runs over hundreds of tiles (the heaviest row of the bench graph: ~480): eight carries travel together and
            // are added in tile order -- one dependent load per carry was what the 181 us of round 4's fix-up pass were
            constexpr int UC = 8;
            for (; s + UC <= t; s += UC) {
                f32x4 v[UC];
#pragma unroll
                for (int k = 0; k < UC; ++k) v[k] = *reinterpret_cast<const f32x4*>(f.tail + (int64_t)(s + k) * kLanes + c4);
#pragma unroll
                for (int k = 0; k < UC; ++k) {
                    a0 += (double)v[k].x;
                    a1 += (double)v[k].y;
                    a2 += (double)v[k].z;
                    a3 += (double)v[k].w;
                }
            }
            for (; s < t; ++s) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(f.tail + (int64_t)s * kLanes + c4);
                a0 += (double)v.x;
                a1 += (double)v.y;
                a2 += (double)v.z;
                a3 += (double)v.w;
            }
            a0 += (double)head[u].x;
            a1 += (double)head[u].y;
            a2 += (double)head[u].z;
            a3 += (double)head[u].w;
            *reinterpret_cast<f32x4*>(sums + (int64_t)row[u] * ld + c4) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
        }
    }
}

// lanes that move one row of the [n, ld] slabs as float4s: 16 for up to 64 columns, 8 for up to 32, 4 for up to 16 (the
// other kernels of the batch use the same shape as k_mm_partial); a wavefront moves 64 / lanes rows per pass
struct CombineParams {
    const float* sums;       // [n, ld] plain row sums (structural zeros never written)
    const float* dst_scale;  // [n] or null
    const float* src_scale;  // [n] or null
    const float* p;          // [n, ld] personalization (AXPBY) or null (PLAIN)
    const uint8_t* p_row_nz; // [n] or null: row flags (PermuteIn::a_row_nz).  bit 0 clear: the row of p is all zeros and adds
                             // nothing (fma(1 - alpha, 0, v) = v): not read.  bit 1 clear: the row has no entries, its sum is 0:
                             // not read.  Both clear: the row is zero in every iterate -- neither read nor written (y, xg and
                             // the other iterate buffer hold zeros there from the start of the run)
    const float* y_old;      // [n, ld] previous iterate (frozen columns copy it) or null
    float*       y;          // [n, ld]
    float*       xg_out;     // [n, ld] next gather slab (y * src_scale) or null
    double       alpha;
    int          plain;      // 1: y = dst * sum
};

// (16 lanes x float4 per row, four rows per wavefront and pass, two passes in flight: the first version walked one row
// per wavefront with lane = column and took 2.3 ms for its 8.4 GB at scale 23 / b = 64)
__global__ __launch_bounds__(WG) void k_mm_combine(CombineParams c, int64_t n, int ld, int b, const BatchState* __restrict__ state,
                                                    double* __restrict__ partial_sum /* [grid][64] */) {
    __shared__ double s_red[WG / 64][kLanes];
    if (state != nullptr && (state->all_done | state->paused)) return;
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int l = lane & (lpr - 1), c4 = 4 * l;
    const bool live = c4 < b;
    f32x4 a, frozen = {0.f, 0.f, 0.f, 0.f};                // per column: alpha * quotient; 1 = the column has stopped
    bool any_frozen = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool col = c4 + k < b;
        a[k] = c.plain ? 1.f : (float)(c.alpha * (state != nullptr && col ? state->scale[c4 + k] : 1.0));
        if (state != nullptr && col && state->done[c4 + k] != 0) {
            frozen[k] = 1.f;
            any_frozen = true;
        }
    }
    any_frozen = __any(any_frozen);
    const float bc = (float)(1.0 - c.alpha);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int64_t first = (blockIdx.x * (int64_t)(WG / 64) + wave_in_wg) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    constexpr int U = PGH_MM_COMB_U;
    int fl_next[U];                                        // row flags one trip ahead (see k_mm_residual)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t r = first + u * stride;
        fl_next[u] = (live && r < n) ? (c.p_row_nz != nullptr ? (int)c.p_row_nz[r] : 3) : 0;
    }
    for (int64_t r0 = first; r0 < n; r0 += stride * U) {
        f32x4 sum[U], pv[U], yo[U];
        float d[U], sc[U];
        int fl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * stride;
            const bool in_range = live && r < n;
            fl[u] = fl_next[u];
            const int64_t rn = r + stride * U;
            fl_next[u] = (live && rn < n) ? (c.p_row_nz != nullptr ? (int)c.p_row_nz[rn] : 3) : 0;
            const bool ok = in_range && fl[u] != 0;
            const int64_t at = (ok ? r : 0) * ld + c4;
            sum[u] = (ok && (fl[u] & 2)) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(c.sums + at)) : f32x4{0.f, 0.f, 0.f, 0.f};
            const bool with_p = ok && !c.plain && (fl[u] & 1);
            pv[u] = with_p ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(c.p + at)) : f32x4{0.f, 0.f, 0.f, 0.f};
            yo[u] = (ok && any_frozen) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(c.y_old + at)) : f32x4{0.f, 0.f, 0.f, 0.f};
            d[u] = (ok && c.dst_scale != nullptr) ? c.dst_scale[r] : 1.f;
            sc[u] = (ok && c.xg_out != nullptr && c.src_scale != nullptr) ? c.src_scale[r] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * stride;
            if (!live || r >= n || fl[u] == 0) continue;
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s = sum[u][k];
                if (c.dst_scale != nullptr) s *= d[u];
                float v = a[k] * s;
                if (!c.plain) v += bc * pv[u][k];
                y[k] = frozen[k] != 0.f ? yo[u][k] : v;
            }
            const int64_t at = r * ld + c4;
            *reinterpret_cast<f32x4*>(c.y + at) = y;
            if (c.xg_out != nullptr) *reinterpret_cast<f32x4*>(c.xg_out + at) = c.src_scale != nullptr ? y * sc[u] : y;
            s0 += (double)y.x;
            s1 += (double)y.y;
            s2 += (double)y.z;
            s3 += (double)y.w;
        }
    }
    // the lane groups hold the same columns: fold them in group order, then the wavefronts in wavefront order
    {
        double t0 = s0, t1 = s1, t2 = s2, t3 = s3;
        for (int off = lpr; off < 64; off += lpr) {
            t0 += __shfl_down(s0, off, 64);
            t1 += __shfl_down(s1, off, 64);
            t2 += __shfl_down(s2, off, 64);
            t3 += __shfl_down(s3, off, 64);
        }
        s0 = t0, s1 = t1, s2 = t2, s3 = t3;
    }
    for (int j = threadIdx.x; j < (WG / 64) * kLanes; j += WG) (&s_red[0][0])[j] = 0.0;      // columns beyond 4 * lanes
    __syncthreads();
    if (lane < lpr) {
        s_red[wave_in_wg][c4 + 0] = s0;
        s_red[wave_in_wg][c4 + 1] = s1;
        s_red[wave_in_wg][c4 + 2] = s2;
        s_red[wave_in_wg][c4 + 3] = s3;
    }
    __syncthreads();
    if (wave_in_wg == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WG / 64; ++w) t += s_red[w][lane];
        partial_sum[(int64_t)blockIdx.x * kLanes + lane] = t;
    }
}

// per-column fold of the block partials ([count][64] doubles): one workgroup per column, thread t adds rows t, t + 256, ...,
// then the lanes and the four wavefronts are combined in a fixed order (deterministic).  (One workgroup for all 64 columns --
// the first version -- read the 1 MB of partials through a single CU: 128 us per fold, two folds per batch step.)
__global__ __launch_bounds__(WG) void k_mm_fold(const double* __restrict__ partials, int count, int linf, double* __restrict__ out) {
    __shared__ double s_red[WG / 64];
    const int col = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += WG) {
        const double v = partials[(int64_t)i * kLanes + col];
        acc = linf ? fmax(acc, v) : acc + v;
    }
    acc = linf ? wave_reduce_max(acc) : wave_reduce_sum(acc);
    if (lane == 0) s_red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_red[0];
#pragma unroll
        for (int k = 1; k < WG / 64; ++k) t = linf ? fmax(t, s_red[k]) : t + s_red[k];
        out[col] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The batch loop keeps ONE slab per iterate: the GATHER slab xg = y * s' (s' = the source scale of the row, 1 where that is zero:
// such a row is never gathered and its slab row holds y itself).  y is what the epilogue computes in registers and never stores;
// the previous iterate's row comes back as xg_old * (1 / s') (one rounding, 6e-8 relative: it enters the residual only), the ranks
// leave through k_mm_permute_out2 the same way.  Two slabs ping-pong, so a paused step can be re-evaluated from both.
// (Round 4 wrote y AND xg every step and read both iterates again in a residual pass of its own: 4.8 GB per batch step at scale 23 /
// b = 64 against 3.0 GB here.)
//
// The residual of a step inside its epilogue (RecursiveGraphFilter._step's quotient + ConvergenceManager per column,
// abstract_filters.py:126-136, convergence.py:96-101): sum_r |y_r * inv - yold_r * scale| needs inv = 1 / sum(y) of the very step being
// written.  As in the single-vector loop (ResParams, pgh_kernels.h) it is PREDICTED per column -- sum(y) = a * sum_j deg_j yold_j + b *
// sum(p), corrected by the last step's measured / predicted ratio -- the kernel evaluates R' and D = sum sign(.) y against inv', the
// close takes R = R' + (inv - inv') D, exact but for rows whose term changes sign between inv' and inv: bounded by 2 |inv - inv'|
// sum|y|.  A column whose tolerance lies inside that bound (or that met a negative / non-finite y: R' = NaN) PAUSES the step for the
// whole batch: every later launch is a no-op, the host runs the separate residual kernel for the step and goes on without the fusion.
// one byte per row of a wavefront's pass (rows row0 .. row0 + 64 / lpr - 1, row0 a multiple of that count; votes = ballot of "my part
// of my row holds a non-zero") as whole dwords from lane 0; returns the number of rows that hold a non-zero (wave-uniform).  Rows a
// pass does not touch are rows that stay zero for the whole run: their byte may be written as 0.
template <int LPR>
__device__ __forceinline__ int mm_store_row_bytes_t(uint8_t* __restrict__ map, int64_t row0, int64_t n, unsigned long long votes, int lane) {
    constexpr int ROWS = 64 / LPR, WORDS = ROWS / 4;
    constexpr unsigned long long kGroup = LPR == 16 ? 0xffffULL : (LPR == 8 ? 0xffULL : 0xfULL);
    uint32_t w[WORDS];
    int count = 0;
#pragma unroll
    for (int k = 0; k < WORDS; ++k) {
        w[k] = 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool nz = ((votes >> ((4 * k + g) * LPR)) & kGroup) != 0ULL;
            w[k] |= (nz ? 1u : 0u) << (8 * g);
            count += nz ? 1 : 0;
        }
    }
    if (lane == 0 && row0 < n) {                               // (the map is padded to whole passes)
        uint32_t* out = reinterpret_cast<uint32_t*>(map + row0);
#pragma unroll
        for (int k = 0; k < WORDS; ++k) out[k] = w[k];
    }
    return count;
}
__device__ __forceinline__ int mm_store_row_bytes(uint8_t* __restrict__ map, int64_t row0, int64_t n, unsigned long long votes, int lpr, int lane) {
    if (lpr == 16) return mm_store_row_bytes_t<16>(map, row0, n, votes, lane);
    if (lpr == 8) return mm_store_row_bytes_t<8>(map, row0, n, votes, lane);
    return mm_store_row_bytes_t<4>(map, row0, n, votes, lane);
}

struct StepParams {
    const float* sums;       // [n, ld] plain row sums (structural zeros never written)
    const f32x4* rowop;      // [n] {dst scale, s', 1 / s', row sum of M} (k_mm_rowops)
    const float* p;          // [n, ld] personalization
    const uint8_t* row_flags;// [n] (PermuteIn2::row_flags) bit 0: the row of p holds a non-zero; bit 1: the row has entries; 0: zero for ever
    const float* xg_old;     // [n, ld]
    float*       xg_new;     // [n, ld]
    double       alpha;
    int          mode;       // 0: S and T only; 1: + the in-kernel residual against state->pred_inv; 2: first step (D = sum of p)
    uint8_t*     nz_map;     // [n] <- 1 where the written row of xg_new holds a non-zero (k_mm_partial<SPARSE>), or null
    int*         nz_part;    // [grid] <- the number of such rows per workgroup (folded by k_mm_close2)
    const int*   nz_prev;    // non-zero rows of xg_old, and the rows that take part in the run: the map is kept while nz_prev * 4 < live_rows
    const int*   live_rows;
    int          affine;     // 1: the workgroups of an XCD take a contiguous eighth of every window (diagnostic)
    const float* zero;       // [ld] zeros: what a row without the operand reads instead (no load sits under a branch)
};

// TRACK: the form that also writes the non-zero map.  It needs 134+ registers (three wavefronts per SIMD instead of four: 900 us instead of
// 690 for the pass at scale 23), so it is a kernel of its own that runs only while the iterate is sparse; both forms are launched for
// every step and the one the device-side test does not select returns at once (as the two forms of k_mm_partial do).
template <bool TRACK>
__global__ __launch_bounds__(WG) void k_mm_step(StepParams c, int64_t n, int ld, int b, const BatchState* __restrict__ state,
                                                 double* __restrict__ partials /* [4][grid][64]: S, T, R', D */) {
    __shared__ double s_red[4][WG / 64][kLanes];
    if (state->all_done | state->paused) return;
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int l = lane & (lpr - 1), c4 = 4 * l;
    const bool live = c4 < b;
    f32x4 a, frozen = {0.f, 0.f, 0.f, 0.f};
    double inv_p[4], scl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool col = c4 + k < b;
        scl[k] = col ? state->scale[c4 + k] : 1.0;
        a[k] = (float)(c.alpha * scl[k]);
        inv_p[k] = (col && c.mode == 1) ? state->pred_inv[c4 + k] : 1.0;
        if (!col || state->done[c4 + k] != 0) frozen[k] = 1.f;
    }
    const float bc = (float)(1.0 - c.alpha);
    double S[4] = {0.0, 0.0, 0.0, 0.0}, T[4] = {0.0, 0.0, 0.0, 0.0}, R[4] = {0.0, 0.0, 0.0, 0.0}, D[4] = {0.0, 0.0, 0.0, 0.0};
    bool neg = false;
    int nz_count = 0;
    // The non-zero map of the slab this step writes is kept only WHILE THE ITERATE IS SPARSE (the gate of k_mm_partial<SPARSE>): once a
    // quarter of the rows hold a non-zero the pass is dense for the rest of the run (this workgroup reports "every row": sticky), and
    // the epilogue spends nothing on the map -- it costs ~150 us of a 690-us pass at scale 23 while it is written.
    const bool track = c.nz_map != nullptr && (long long)(*c.nz_prev) * 4 < (long long)(*c.live_rows);
    if (track != TRACK) return;
    // passes of 64 / lpr rows, the whole grid streaming through one window of the slabs (a wavefront per 16-KB chunk was measured: 832 us
    // against 689 for this pass).  The workgroups that share an XCD (blockIdx % 8) take a CONTIGUOUS eighth of every window, so the map
    // bytes of one 64-byte line come from one L2.
    const int64_t vb = c.affine ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t first = (vb * (WG / 64) + wave_in_wg) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    constexpr int U = PGH_MM_COMB_U;
    int fl_next[U];                                        // row flags one trip ahead: the loads of a trip do not wait for its flags
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t r = first + u * stride;
        fl_next[u] = (int)c.row_flags[r < n ? r : n - 1];       // (raw: whether the row exists is applied when the flags are USED, a trip later)
    }
    // Round 5: every load of a trip is issued unconditionally -- a row without the operand reads the zero row instead.  As `cond ? load : 0`
    // each of the eight loads of a trip sat in a basic block of its own with a full wait behind it: eight round trips one after the other.
    const float* const zero4 = c.zero + c4;
    for (int64_t r0 = first; r0 < n; r0 += stride * U) {
        f32x4 sum[U], pv[U], xo[U], op[U];
        int fl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * stride;
            fl[u] = (live && r < n) ? fl_next[u] : 0;
            const int64_t rn = r + stride * U;
            fl_next[u] = (int)c.row_flags[rn < n ? rn : n - 1];
            const bool ok = fl[u] != 0;
            const int64_t at = (ok ? r : 0) * ld + c4;
            sum[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((ok && (fl[u] & 2)) ? c.sums + at : zero4));
            pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((ok && (fl[u] & 1)) ? c.p + at : zero4));
            xo[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ok ? c.xg_old + at : zero4));
            const f32x4 ro = c.rowop[ok ? r : 0];
            op[u] = ok ? ro : f32x4{1.f, 1.f, 1.f, 0.f};
        }
        bool nzl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            nzl[u] = false;
            if (fl[u] == 0) continue;
            const int64_t r = r0 + u * stride;
            f32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float yo = xo[u][k] * op[u][2];                       // the previous iterate's y (one rounding)
                const float y = a[k] * (sum[u][k] * op[u][0]) + bc * pv[u][k];
                if (frozen[k] != 0.f) {
                    out[k] = xo[u][k];                                       // a stopped column keeps its row bit for bit
                    continue;
                }
                out[k] = y * op[u][1];
                S[k] += (double)y;
                T[k] += (double)op[u][3] * (double)y;
                neg = neg || !(y >= 0.f);
                if (c.mode == 2) {
                    D[k] += (double)pv[u][k];
                } else if (c.mode == 1) {
                    const double d = (double)y * inv_p[k] - (double)yo * scl[k];
                    R[k] += fabs(d);
                    D[k] += d < 0.0 ? -(double)y : (double)y;
                }
            }
            *reinterpret_cast<f32x4*>(c.xg_new + r * ld + c4) = out;
            if (TRACK) nzl[u] = out.x != 0.f || out.y != 0.f || out.z != 0.f || out.w != 0.f;
        }
        if (TRACK) {
            // one byte per row of the pass: does the row hold a non-zero?  (the lanes of a row vote; rows a pass skips are zero for ever)
#pragma unroll
            for (int u = 0; u < U; ++u) nz_count += mm_store_row_bytes(c.nz_map, r0 - lane / lpr + u * stride, n, __ballot(nzl[u]), lpr, lane);
        }
    }
    // (the count leaves per workgroup and is folded by the step's close: 8192 atomic adds to ONE word cost 250 us of the pass)
    __shared__ int s_nz[WG / 64];
    if (lane == 0) s_nz[wave_in_wg] = nz_count;
    __syncthreads();
    if (c.nz_part != nullptr && threadIdx.x == 0) c.nz_part[blockIdx.x] = TRACK ? s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3] : (blockIdx.x == 0 ? *c.live_rows : 0);
    if (c.mode == 1 && neg) R[0] = R[1] = R[2] = R[3] = __builtin_nan("");      // (poisons the lane's four columns: a pause, never a wrong verdict)
    // the lane groups hold the same columns: fold them in group order, then the wavefronts in wavefront order
    for (int j = threadIdx.x; j < 4 * (WG / 64) * kLanes; j += WG) (&s_red[0][0][0])[j] = 0.0;      // columns beyond 4 * lanes
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double t0 = S[k], t1 = T[k], t2 = R[k], t3 = D[k];
        for (int off = lpr; off < 64; off += lpr) {
            t0 += __shfl_down(S[k], off, 64);
            t1 += __shfl_down(T[k], off, 64);
            t2 += __shfl_down(R[k], off, 64);
            t3 += __shfl_down(D[k], off, 64);
        }
        if (lane < lpr) {
            s_red[0][wave_in_wg][c4 + k] = t0;
            s_red[1][wave_in_wg][c4 + k] = t1;
            s_red[2][wave_in_wg][c4 + k] = t2;
            s_red[3][wave_in_wg][c4 + k] = t3;
        }
    }
    __syncthreads();
    {
        const int q = wave_in_wg;                          // WG / 64 == 4 wavefronts: wavefront q folds quantity q
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WG / 64; ++w) t += s_red[q][w][lane];
        partials[((int64_t)q * gridDim.x + blockIdx.x) * kLanes + lane] = t;
    }
}
static_assert(WG / 64 == 4, "k_mm_step folds its four quantities with four wavefronts");

// per-column fold of the step's partials: workgroup (col, q) adds partials[q][0 .. count)[col] in a fixed order
__global__ __launch_bounds__(WG) void k_mm_fold4(const double* __restrict__ partials, int count, const BatchState* __restrict__ state,
                                                  double* __restrict__ out /* [4][64] */) {
    __shared__ double s_red[WG / 64];
    if (state->all_done | state->paused) return;
    const int col = blockIdx.x, q = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += WG) acc += partials[((int64_t)q * count + i) * kLanes + col];
    acc = wave_reduce_sum(acc);
    if (lane == 0) s_red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_red[0];
#pragma unroll
        for (int k = 1; k < WG / 64; ++k) t += s_red[k];
        out[q * kLanes + col] = t;
    }
}

// the separate residual pass (first step of a run, the max rule, graph_dropout, paused steps): sum_r |y_r / S - yold_r * scale| per
// column from the two gather slabs (y = xg * (1 / s')); S = the step's folded sum(y) (folded[0]), scale = the previous quotient
__global__ __launch_bounds__(WG) void k_mm_residual2(const float* __restrict__ xg_new, const float* __restrict__ xg_old, const f32x4* __restrict__ rowop,
                                                      int64_t n, int ld, int b, int use_quotient, int linf, const BatchState* __restrict__ state,
                                                      const double* __restrict__ folded, double* __restrict__ partial_res,
                                                      const uint8_t* __restrict__ row_flags, int resume) {
    __shared__ double s_red[WG / 64][kLanes];
    if (state->all_done | (resume ? 0 : state->paused)) return;
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int l = lane & (lpr - 1), c4 = 4 * l;
    double inv[4], scale[4], acc[4] = {0.0, 0.0, 0.0, 0.0};
    bool want[4], any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool col = c4 + k < b;
        const double S = col ? folded[c4 + k] : 1.0;
        inv[k] = use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;
        scale[k] = col ? state->scale[c4 + k] : 1.0;
        want[k] = col && !state->done[c4 + k];
        any = any || want[k];
    }
    const int64_t first = (blockIdx.x * (int64_t)(WG / 64) + wave_in_wg) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    constexpr int U = PGH_MM_RES_U;
    if (any) {
        uint8_t fl_next[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = first + u * stride;
            fl_next[u] = r < n ? row_flags[r] : (uint8_t)0;
        }
        for (int64_t r0 = first; r0 < n; r0 += stride * U) {
            f32x4 a[U], o[U];
            float un[U];
            bool use[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + u * stride;
                use[u] = fl_next[u] != 0;                 // rows that are zero in every iterate add |0 - 0|: not read
                const int64_t rn = r + stride * U;
                fl_next[u] = rn < n ? row_flags[rn] : (uint8_t)0;
                const int64_t at = (use[u] ? r : 0) * ld + c4;
                a[u] = use[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xg_new + at)) : f32x4{0.f, 0.f, 0.f, 0.f};
                o[u] = use[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xg_old + at)) : f32x4{0.f, 0.f, 0.f, 0.f};
                un[u] = use[u] ? rowop[r][2] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!use[u]) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double d = fabs((double)(a[u][k] * un[u]) * inv[k] - (double)(o[u][k] * un[u]) * scale[k]);
                    acc[k] = linf ? fmax(acc[k], d) : acc[k] + d;
                }
            }
        }
    }
    for (int j = threadIdx.x; j < (WG / 64) * kLanes; j += WG) (&s_red[0][0])[j] = 0.0;      // columns beyond 4 * lanes
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double t = acc[k];
        for (int off = lpr; off < 64; off += lpr) {
            const double o = __shfl_down(acc[k], off, 64);
            t = linf ? fmax(t, o) : t + o;
        }
        if (lane < lpr) s_red[wave_in_wg][c4 + k] = want[k] ? t : 0.0;
    }
    __syncthreads();
    if (wave_in_wg == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WG / 64; ++w) t = linf ? fmax(t, s_red[w][lane]) : t + s_red[w][lane];
        partial_res[(int64_t)blockIdx.x * kLanes + lane] = t;
    }
}

// per-column fold of one quantity ([count][64] partials; sum or max), resume: also behind a pending pause
__global__ __launch_bounds__(WG) void k_mm_fold1(const double* __restrict__ partials, int count, int linf, const BatchState* __restrict__ state,
                                                  double* __restrict__ out, int resume) {
    __shared__ double s_red[WG / 64];
    if (state->all_done | (resume ? 0 : state->paused)) return;
    const int col = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += WG) {
        const double v = partials[(int64_t)i * kLanes + col];
        acc = linf ? fmax(acc, v) : acc + v;
    }
    acc = linf ? wave_reduce_max(acc) : wave_reduce_sum(acc);
    if (lane == 0) s_red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_red[0];
#pragma unroll
        for (int k = 1; k < WG / 64; ++k) t = linf ? fmax(t, s_red[k]) : t + s_red[k];
        out[col] = t;
    }
}

// closes a batched step: per-column quotient, step count, ConvergenceManager check (convergence.py:96-101) and the NEXT step's predicted
// quotient.  mode 1: the residual came from k_mm_step against the predicted quotient (folded[2], folded[3]); mode 0 / 2: from the
// separate kernel (state->err holds the folded value; 2 = first step: folded[3] is sum(p)).  resume: this close answers a pause.
struct CloseParams {
    double tol, alpha;
    long long n_orig;
    int use_quotient, check, err_kind, mode, resume;
    const int* nz_part;      // [nz_parts] per-workgroup counts of k_mm_step (or null)
    int* nz_total;           // <- their sum
    int nz_parts;
};
__global__ void k_mm_close2(BatchState* __restrict__ state, const double* __restrict__ folded, const double* __restrict__ err_folded, CloseParams cp) {
    const int lane = threadIdx.x;
    if (state->all_done | (cp.resume ? 0 : state->paused)) return;
    const bool live = lane < state->b && !state->done[lane];
    const double S = folded[lane], T = folded[kLanes + lane], R = folded[2 * kLanes + lane], D = folded[3 * kLanes + lane];
    const double scale_new = cp.use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;
    const double sum_p = cp.mode == 2 ? D : state->sum_p[lane];
    // with the factors as k_mm_step forms them: (float)(alpha * scale), (float)(1 - alpha)
    const double next_raw = (double)(float)(cp.alpha * scale_new) * T + (double)(float)(1.0 - cp.alpha) * sum_p;
    const double raw_now = state->pred_raw[lane];
    const double ratio = (cp.mode == 1 && raw_now != 0.0) ? S / raw_now : 1.0;
    const double S_next = next_raw * ((ratio == ratio && fabs(ratio - 1.0) < 1e-4) ? ratio : 1.0);
    const double pred_next = cp.use_quotient ? (S_next != 0.0 ? 1.0 / S_next : 0.0) : 1.0;
    double err = 0.0, slack = 0.0;
    int verdict = 0;
    if (cp.check && live) {
        if (cp.mode == 1) {
            const double ip = state->pred_inv[lane];
            err = R + (scale_new - ip) * D;
            slack = 2.0 * fabs(scale_new - ip) * fabs(S);
        } else {
            err = err_folded[lane];
        }
        if (cp.err_kind == PGH_ERR_MABS) {
            err /= (double)cp.n_orig;
            slack /= (double)cp.n_orig;
        }
        if (cp.mode == 1 && !(fabs(err - cp.tol) > 2.0 * slack)) verdict = 2;       // too close to call (or not finite)
        else if (err <= cp.tol) verdict = 1;
    }
    const unsigned long long pause = __ballot(verdict == 2);
    if (pause != 0ULL) {                                   // nothing of the step is committed: the host re-evaluates it (mode 0, resume)
        if (lane == 0) state->paused = 1;
        return;
    }
    int done = 1;
    if (live) {
        state->scale[lane] = scale_new;
        state->sum[lane] = S;
        state->sum_p[lane] = sum_p;
        state->pred_inv[lane] = pred_next;
        state->pred_raw[lane] = next_raw;
        state->steps[lane] += 1;
        if (cp.check) {
            state->err[lane] = err;
            if (verdict == 1) {
                state->done[lane] = 1;
                state->converged[lane] = 1;
            }
        }
        done = state->done[lane];
    }
    const unsigned long long all = __ballot(done != 0);
    if (lane == 0) {
        state->all_done = (all == ~0ULL) ? 1 : 0;
        state->paused = 0;
        state->executed += 1;
    }
    if (cp.nz_part != nullptr) {                           // non-zero rows of the slab this step wrote: the gate of the next gather pass
        int t = 0;
        for (int i = lane; i < cp.nz_parts; i += kLanes) t += cp.nz_part[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0) *cp.nz_total = t;
    }
}

__global__ void k_mm_state_init(BatchState* state, int b) {
    const int lane = threadIdx.x;
    state->scale[lane] = 1.0;
    state->err[lane] = 0.0;
    state->sum[lane] = 0.0;
    state->pred_inv[lane] = 1.0;
    state->pred_raw[lane] = 0.0;
    state->sum_p[lane] = 0.0;
    state->done[lane] = lane < b ? 0 : 1;
    state->steps[lane] = 0;
    state->converged[lane] = 0;
    if (lane == 0) {
        state->all_done = 0;
        state->paused = 0;
        state->executed = 0;
        state->b = b;
    }
}

// the epilogue's row operands in the multi-seed id space, one 16-byte word per row (ensure_mm_layout)
__global__ __launch_bounds__(WG) void k_mm_rowops(const float* __restrict__ dst_scale, const float* __restrict__ src_scale, const float* __restrict__ degrees,
                                                   const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid, f32x4* __restrict__ out) {
    for (int64_t r = blockIdx.x * (int64_t)WG + threadIdx.x; r < n_int; r += (int64_t)gridDim.x * WG) {
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        const float s = src_scale != nullptr ? src_scale[r] : 1.f;
        const float sp = s != 0.f ? s : 1.f;               // a row nobody gathers keeps y itself in the slab
        out[r] = f32x4{dst_scale != nullptr ? dst_scale[r] : 1.f, sp, 1.f / sp, (o >= 0 && degrees != nullptr) ? degrees[o] : 0.f};
    }
}

// caller-space slabs -> the loop's internal slabs in ONE pass: the personalization (rows that hold a non-zero only: nobody reads the
// others), the first gather slab (start iterate * s', every row), zeros in the rows of the second slab that no step will ever write,
// and the row flags.  (Round 4 wrote three whole slabs here.)
struct PermuteIn2 {
    const float* src_p;      // caller's personalization [n, b]
    const float* src_x;      // caller's start iterate [n, b]; may equal src_p
    float*       out_p;      // [n_int, ld]
    float*       out_xg0;    // [n_int, ld]
    float*       out_xg1;    // [n_int, ld]
    const f32x4* rowop;
    uint8_t*     row_flags;  // [n_int] bit 0: the row of p holds a non-zero; bit 1: the row of M^T holds entries (row_has; all rows when null)
    const uint8_t* row_has;
    uint8_t*     nz_map;     // [n_int] <- 1 where the row of out_xg0 holds a non-zero (k_mm_partial<SPARSE>)
    int*         nz_rows;    // [wavefronts of the grid] <- their number, per wavefront (k_mm_count_fold)
    int*         live_rows;  // [wavefronts of the grid] <- the rows whose flags are not 0 (the rows that take part in the run)
    double*      pred_part;  // [2][grid][64] per-workgroup {sum_r deg_r * x0[r], sum_r p[r]} per column: the first step's predicted quotient
                             // (k_mm_first_pred), or null
};
__global__ __launch_bounds__(WG) void k_mm_permute_in2(PermuteIn2 q, const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid, int b, int ld) {
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int lane = threadIdx.x & 63, l = lane & (lpr - 1), c4 = 4 * l;
    if (c4 >= ld) return;
    const int64_t first = (blockIdx.x * (int64_t)(WG / 64) + (threadIdx.x >> 6)) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    const bool vec = (b & 3) == 0;
    int nz_count = 0, live_count = 0;
    double t0[4] = {0.0, 0.0, 0.0, 0.0}, sp[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t r = first; r < n_int; r += stride) {
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        auto fetch = [&](const float* src) __attribute__((always_inline)) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (o >= 0) {
                if (vec) v = *reinterpret_cast<const f32x4*>(src + o * b + c4);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c4 + k < b) v[k] = src[o * b + c4 + k];
                }
            }
            return v;
        };
        const int64_t at = r * ld + c4;
        const f32x4 vp = fetch(q.src_p);
        const bool nz = vp.x != 0.f || vp.y != 0.f || vp.z != 0.f || vp.w != 0.f;
        const unsigned long long any = __ballot(nz) >> (lane & ~(lpr - 1));       // this row's lanes from bit 0 on
        const int flags = ((any & ((1ULL << lpr) - 1ULL)) != 0ULL ? 1 : 0) | ((q.row_has == nullptr || q.row_has[r] != 0) ? 2 : 0);
        if (l == 0) q.row_flags[r] = (uint8_t)flags;
        if (flags & 1) *reinterpret_cast<f32x4*>(q.out_p + at) = vp;
        const f32x4 vx = q.src_x == q.src_p ? vp : fetch(q.src_x);
        *reinterpret_cast<f32x4*>(q.out_xg0 + at) = vx * q.rowop[r][1];
        if (flags == 0) *reinterpret_cast<f32x4*>(q.out_xg1 + at) = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool xnz = vx.x != 0.f || vx.y != 0.f || vx.z != 0.f || vx.w != 0.f;
        if (q.pred_part != nullptr && (xnz || nz)) {
            const double deg = (double)q.rowop[r][3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t0[k] += deg * (double)vx[k];
                sp[k] += (double)vp[k];
            }
        }
        const unsigned long long anyx = __ballot(xnz) >> (lane & ~(lpr - 1));
        const bool row_nz = (anyx & ((1ULL << lpr) - 1ULL)) != 0ULL;
        (void)row_nz;
        // (wave-uniform counts by ballot: lanes beyond the row length have left the kernel, a shuffle would read their registers)
        nz_count += mm_store_row_bytes(q.nz_map, r - lane / lpr, n_int, __ballot(xnz), lpr, lane);
        live_count += __popcll(__ballot(l == 0 && flags != 0));
    }
    if (lane == 0) {                                       // per wavefront: folded by k_mm_count_fold (no same-address atomics)
        q.nz_rows[blockIdx.x * (WG / 64) + (threadIdx.x >> 6)] = nz_count;
        q.live_rows[blockIdx.x * (WG / 64) + (threadIdx.x >> 6)] = live_count;
    }
    if (q.pred_part != nullptr) {
        // the lane groups hold the same columns: fold them in group order, then the wavefronts in wavefront order (as k_mm_step)
        __shared__ double s_pred[2][WG / 64][kLanes];
        const int wave_in_wg = threadIdx.x >> 6;
        for (int j = threadIdx.x; j < 2 * (WG / 64) * kLanes; j += WG) (&s_pred[0][0][0])[j] = 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double a0 = t0[k], a1 = sp[k];
            for (int off = lpr; off < 64; off += lpr) {
                a0 += __shfl_down(t0[k], off, 64);
                a1 += __shfl_down(sp[k], off, 64);
            }
            if (lane < lpr) {
                s_pred[0][wave_in_wg][c4 + k] = a0;
                s_pred[1][wave_in_wg][c4 + k] = a1;
            }
        }
        __syncthreads();
        if (wave_in_wg < 2) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WG / 64; ++w) t += s_pred[wave_in_wg][w][lane];
            q.pred_part[((int64_t)wave_in_wg * gridDim.x + blockIdx.x) * kLanes + lane] = t;
        }
    }
}
// the first step's predicted quotient per column, from the way in's sums (the single-vector loop's first_prediction): sum(y_1) = a * sum
// deg x0 + b * sum p with the factors as k_mm_step forms them
__global__ void k_mm_first_pred(BatchState* __restrict__ state, const double* __restrict__ t0, const double* __restrict__ sp, double alpha, int use_quotient) {
    const int lane = threadIdx.x;
    const double raw = (double)(float)alpha * t0[lane] + (double)(float)(1.0 - alpha) * sp[lane];
    state->sum_p[lane] = sp[lane];
    state->pred_raw[lane] = raw;
    state->pred_inv[lane] = use_quotient ? (raw != 0.0 ? 1.0 / raw : 0.0) : 1.0;
}
__global__ __launch_bounds__(WG) void k_mm_count_fold(const int* __restrict__ a, const int* __restrict__ b, int count, int* __restrict__ out_a, int* __restrict__ out_b) {
    __shared__ int s_a[WG / 64], s_b[WG / 64];
    int ta = 0, tb = 0;
    for (int i = threadIdx.x; i < count; i += WG) {
        ta += a[i];
        tb += b[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ta += __shfl_xor(ta, d, 64);
        tb += __shfl_xor(tb, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_a[threadIdx.x >> 6] = ta;
        s_b[threadIdx.x >> 6] = tb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *out_a = s_a[0] + s_a[1] + s_a[2] + s_a[3];
        *out_b = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    }
}

// ranks out of the loop: dst[old id] = xg[r] * (1 / s') * column factor (rows whose flag is 0 are zero for ever: written as zeros)
__global__ __launch_bounds__(WG) void k_mm_permute_out2(const float* __restrict__ xg, const f32x4* __restrict__ rowop, const int32_t* __restrict__ perm,
                                                         int64_t n_int, int64_t n_valid, int b, int ld, const double* __restrict__ col_factor,
                                                         float* __restrict__ dst, const uint8_t* __restrict__ row_flags) {
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int lane = threadIdx.x & 63, l = lane & (lpr - 1), c4 = 4 * l;
    if (c4 >= b) return;
    const int64_t first = (blockIdx.x * (int64_t)(WG / 64) + (threadIdx.x >> 6)) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    const bool vec = (b & 3) == 0;
    f32x4 factor = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c4 + k < b) factor[k] = (float)col_factor[c4 + k];
    for (int64_t r = first; r < n_int; r += stride) {
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        if (o < 0) continue;
        // (a row that stayed zero leaves as zeros from here: every caller id has exactly one row, so the destination needs no clearing pass)
        const f32x4 v = row_flags[r] == 0 ? f32x4{0.f, 0.f, 0.f, 0.f}
                                          : (__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xg + r * ld + c4)) * rowop[r][2]) * factor;
        if (vec) *reinterpret_cast<f32x4*>(dst + o * b + c4) = v;
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c4 + k < b) dst[o * b + c4 + k] = v[k];
        }
    }
}

// caller-space slab(s) -> internal (relabelled) slabs, holes zero: the personalization, the start iterate and the first gather
// slab (start * row_scale) leave in ONE pass over the permutation (three separate passes took 1.6 ms each at scale 23 /
// b = 64).  A 16-lane group moves one row; the internal slabs round the row length up to whole float4s (columns b .. ld - 1
// are zero), the caller's rows are b floats long.
struct PermuteIn {
    const float* src_a;      // -> out_a (personalization) or null
    const float* src_b;      // -> out_b, and scaled -> out_bs (start iterate); may equal src_a
    float*       out_a;
    float*       out_b;
    float*       out_bs;     // out_b * row_scale, or null
    const float* row_scale;  // [n_int] or null
    uint8_t*     a_row_nz;   // [n_int] or null: row flags for k_mm_combine / k_mm_residual: bit 0 = row r of out_a holds a
                             // non-zero, bit 1 = row r of M^T holds entries (row_has); 0 = the row is zero in every iterate
    const uint8_t* row_has;  // [n_int] static marks (k_mm_mark_rows) or null
};

__global__ __launch_bounds__(WG) void k_mm_permute_in(PermuteIn q, const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid, int b, int ld) {
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int lane = threadIdx.x & 63, l = lane & (lpr - 1), c4 = 4 * l;
    if (c4 >= ld) return;
    const int64_t first = (blockIdx.x * (int64_t)(WG / 64) + (threadIdx.x >> 6)) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    const bool vec = (b & 3) == 0;
    for (int64_t r = first; r < n_int; r += stride) {
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        auto fetch = [&](const float* src) __attribute__((always_inline)) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (o >= 0) {
                if (vec) v = *reinterpret_cast<const f32x4*>(src + o * b + c4);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c4 + k < b) v[k] = src[o * b + c4 + k];
                }
            }
            return v;
        };
        const int64_t at = r * ld + c4;
        f32x4 va = {0.f, 0.f, 0.f, 0.f};
        if (q.src_a != nullptr) {
            va = fetch(q.src_a);
            *reinterpret_cast<f32x4*>(q.out_a + at) = va;
            if (q.a_row_nz != nullptr) {
                const bool nz = va.x != 0.f || va.y != 0.f || va.z != 0.f || va.w != 0.f;
                const unsigned long long any = __ballot(nz) >> (lane & ~(lpr - 1));       // this row's lanes from bit 0 on
                if (l == 0)
                    q.a_row_nz[r] = (uint8_t)(((any & ((1ULL << lpr) - 1ULL)) != 0ULL ? 1 : 0) | ((q.row_has == nullptr || q.row_has[r] != 0) ? 2 : 0));
            }
        }
        if (q.src_b != nullptr) {
            const f32x4 vb = q.src_b == q.src_a ? va : fetch(q.src_b);
            if (q.out_b != nullptr) *reinterpret_cast<f32x4*>(q.out_b + at) = vb;
            if (q.out_bs != nullptr) *reinterpret_cast<f32x4*>(q.out_bs + at) = q.row_scale != nullptr ? vb * q.row_scale[r] : vb;
        }
    }
}

// (row_flags: rows whose flag is 0 are zero and the caller has cleared dst: they are neither read nor written)
__global__ __launch_bounds__(WG) void k_mm_permute_out(const float* __restrict__ src, const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid,
                                                        int b, int ld, const double* __restrict__ col_factor, float* __restrict__ dst,
                                                        const uint8_t* __restrict__ row_flags = nullptr) {
    const int lpr = lanes_per_row(ld), rows_per_wave = 64 / lpr;
    const int lane = threadIdx.x & 63, l = lane & (lpr - 1), c4 = 4 * l;
    if (c4 >= b) return;
    const int64_t first = (blockIdx.x * (int64_t)(WG / 64) + (threadIdx.x >> 6)) * rows_per_wave + lane / lpr;
    const int64_t stride = (int64_t)gridDim.x * (WG / 64) * rows_per_wave;
    const bool vec = (b & 3) == 0;
    f32x4 factor = {1.f, 1.f, 1.f, 1.f};
    if (col_factor != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c4 + k < b) factor[k] = (float)col_factor[c4 + k];
    }
    for (int64_t r = first; r < n_int; r += stride) {
        if (row_flags != nullptr && row_flags[r] == 0) continue;
        const int64_t o = perm ? perm[r] : (r < n_valid ? r : -1);
        if (o < 0) continue;
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + r * ld + c4)) * factor;
        if (vec) *reinterpret_cast<f32x4*>(dst + o * b + c4) = v;
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c4 + k < b) dst[o * b + c4 + k] = v[k];
        }
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
    PGH_HIP(pooled_malloc(&f.part, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes));
    PGH_HIP(pooled_malloc(&f.xg, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes));
    PGH_HIP(hipMemsetAsync(f.part, 0, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes, rt().stream));
    PGH_HIP(hipMemsetAsync(f.xg, 0, sizeof(float) * (size_t)(f.num_tiles + 1) * kLanes, rt().stream));
    PGH_HIP(pooled_malloc(&f.mm_close, sizeof(int32_t) * (size_t)f.num_entries));
    k_mm_close_rows<<<blocks_for((int64_t)f.num_tiles * 64, 64), WG, 0, rt().stream>>>(f.colf, f.tile, f.seg_row, f.num_tiles, f.mm_close);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    f.device_bytes += (int64_t)f.num_entries * 4;
    PGH_HIP(pooled_malloc(&f.mm_row_has, (size_t)(f.n_out > 0 ? f.n_out : 1)));
    PGH_HIP(hipMemsetAsync(f.mm_row_has, 0, (size_t)(f.n_out > 0 ? f.n_out : 1), rt().stream));
    k_mm_mark_rows<<<blocks_for(f.num_entries), WG, 0, rt().stream>>>(f.mm_close, f.num_entries, f.tile, f.seg_row, f.num_tiles, f.mm_row_has);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    f.device_bytes += f.n_out;
    PGH_HIP(pooled_malloc(&f.mm_rowop, sizeof(float) * 4 * (size_t)(f.n_out > 0 ? f.n_out : 1)));
    k_mm_rowops<<<blocks_for(f.n_out), WG, 0, rt().stream>>>(f.dst_scale, f.src_scale, g->degrees, f.perm, f.n_out, g->n_rows,
                                                              reinterpret_cast<f32x4*>(f.mm_rowop));
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    f.device_bytes += (int64_t)f.n_out * 16;
    return 0;
}

MMView mm_view(const BsfFormat& f) {
    MMView v;
    v.colf = f.colf;
    v.val = f.val;
    v.seg_row = f.seg_row;
    v.close = f.mm_close;
    v.tile = f.tile;
    v.head = f.part;
    v.tail = f.xg;
    v.num_tiles = f.num_tiles;
    return v;
}

// sums <- M^T-times-gather-slab (plain row sums in the internal id space); sums must have its structural zeros in place
int ensure_mm_edge_ids(pgh_graph_s* g) {
    BsfFormat& f = g->bsf_mm;
    if (f.mm_edge != nullptr) return 0;
    PGH_CHECK(g->rowptr != nullptr && g->col != nullptr, "graph_dropout on a batch needs the CSR image of the graph");
    PGH_HIP(pooled_malloc(&f.mm_edge, sizeof(int32_t) * (size_t)f.num_entries));
    k_mm_edge_ids<<<blocks_for((int64_t)f.num_tiles * 64, 64), WG, 0, rt().stream>>>(f.colf, f.tile, f.seg_row, f.num_tiles, f.perm, g->rowptr,
                                                                                      g->col, f.mm_edge);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    f.device_bytes += (int64_t)f.num_entries * 4;
    return 0;
}

int mm_partial(pgh_graph_s* g, const float* xg, int ld, int b, float* sums, const BatchState* state, const MMDrop* drop = nullptr,
               const SparseGate* gate = nullptr) {
    Runtime& r = rt();
    const BsfFormat& f = g->bsf_mm;
    const MMView v = mm_view(f);
    // persistent grid: as many workgroups as the registers let a CU hold (PGH_MM_WGS overrides: diagnostic)
    const int lpr = lanes_per_row(ld);
    const int which = (f.val ? 3 : 0) + (lpr == 16 ? 0 : (lpr == 8 ? 1 : 2));
    const void* kernels[6] = {(const void*)k_mm_partial<false, 16>, (const void*)k_mm_partial<false, 8>, (const void*)k_mm_partial<false, 4>,
                              (const void*)k_mm_partial<true, 16>,  (const void*)k_mm_partial<true, 8>,  (const void*)k_mm_partial<true, 4>};
    static int per_cu_of[6] = {0, 0, 0, 0, 0, 0};
    int& per_cu = per_cu_of[which];
    if (per_cu == 0) {
        if (getenv("PGH_MM_WGS") != nullptr) per_cu = atoi(getenv("PGH_MM_WGS"));
        else PGH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernels[which], WG, 0));
        if (per_cu < 1) per_cu = 1;
    }
    const int grid = r.num_cus * per_cu;
    if (drop != nullptr) {          // graph_dropout: the same shapes with the mask word (the value-free stream gets a factor per entry)
        ProfScope prof(PGH_K_SPMM);
        switch (which) {
            case 0: k_mm_partial<false, 16, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, *drop); break;
            case 1: k_mm_partial<false, 8, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, *drop); break;
            case 2: k_mm_partial<false, 4, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, *drop); break;
            case 3: k_mm_partial<true, 16, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, *drop); break;
            case 4: k_mm_partial<true, 8, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, *drop); break;
            default: k_mm_partial<true, 4, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, *drop); break;
        }
    } else {
        // both forms of the pass for every step: each evaluates the gate on the device, the one it does not select returns at once
        const SparseGate sg = gate != nullptr ? *gate : SparseGate{};
        ProfScope prof(PGH_K_SPMM);
        switch (which) {
            case 0: k_mm_partial<false, 16><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
            case 1: k_mm_partial<false, 8><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
            case 2: k_mm_partial<false, 4><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
            case 3: k_mm_partial<true, 16><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
            case 4: k_mm_partial<true, 8><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
            default: k_mm_partial<true, 4><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
        }
        if (sg.map != nullptr) {
            switch (which) {
                case 0: k_mm_partial<false, 16, false, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
                case 1: k_mm_partial<false, 8, false, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
                case 2: k_mm_partial<false, 4, false, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
                case 3: k_mm_partial<true, 16, false, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
                case 4: k_mm_partial<true, 8, false, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
                default: k_mm_partial<true, 4, false, true><<<grid, WG, 0, r.stream>>>(v, xg, ld, b, sums, state, MMDrop{}, sg); break;
            }
        }
    }
    {
        ProfScope prof(PGH_K_FIXUP);
        k_mm_fixup<<<blocks_for((int64_t)f.num_tiles * 64, 64), WG, 0, r.stream>>>(v, ld, b, sums, state);
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

int combine_grid() { return rt().num_cus * 8; }
int cgrid_of() { return combine_grid(); }

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
namespace {
// the mask of one launch: null for rate == 0
int make_drop(pgh_graph_s* g, double rate, uint64_t seed, MMDrop* out, const MMDrop** use) {
    *use = nullptr;
    if (rate == 0.0) return 0;
    PGH_CHECK(rate > 0.0 && rate < 1.0, "graph_dropout: the rate must lie in [0, 1)");
    PGH_TRY(ensure_mm_edge_ids(g));
    out->edge = g->bsf_mm.mm_edge;
    out->seed = seed;
    out->threshold = (uint32_t)floor(rate * 4294967296.0);
    out->keep_scale = (float)(1.0 / (1.0 - rate));
    *use = out;
    return 0;
}
int spmm_impl(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y, double rate, uint64_t seed);
int batch_impl(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales, double rate, uint64_t seed0,
               pgh_loop_result* results);
}  // namespace

extern "C" int pgh_spmm(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y) { return spmm_impl(g, x, y, 0.0, 0); }
extern "C" int pgh_spmm_dropout(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y, double rate, uint64_t seed) { return spmm_impl(g, x, y, rate, seed); }

namespace {
int spmm_impl(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y, double rate, uint64_t seed) {
    PGH_CHECK(g && x && y, "pgh_spmm: null argument");
    PGH_CHECK(x->n == g->n_rows && y->n == g->n_cols && x->b == y->b, "pgh_spmm: shape mismatch");
    PGH_CHECK(x->b >= 1 && x->b <= kLanes, "pgh_spmm: the batch width must be in [1, 64]");
    PGH_CHECK(x->data != y->data, "pgh_spmm: conv must be pure (output aliases input)");
    if (g->n_cols == 0) return 0;
    PGH_TRY(ensure_mm_layout(g));
    Runtime& r = rt();
    const BsfFormat& f = g->bsf_mm;
    const int b = x->b, ld = (b + 3) & ~3;
    const int64_t n_int = f.n_out;
    DevBytes xg, sums, yint, partial;
    PGH_TRY(xg.alloc(sizeof(float) * (size_t)n_int * ld));
    PGH_TRY(sums.alloc(sizeof(float) * (size_t)n_int * ld));
    PGH_TRY(yint.alloc(sizeof(float) * (size_t)n_int * ld));
    PGH_TRY(partial.alloc(sizeof(double) * (size_t)combine_grid() * kLanes));
    PGH_HIP(hipMemsetAsync(sums.p, 0, sizeof(float) * (size_t)n_int * ld, r.stream));
    {
        PermuteIn q{};
        q.src_b = x->data;
        q.out_bs = xg.as<float>();
        q.row_scale = f.src_scale;
        k_mm_permute_in<<<blocks_for(n_int * lanes_per_row(ld)), WG, 0, r.stream>>>(q, f.perm, n_int, g->n_rows, b, ld);
    }
    MMDrop drop_store;
    const MMDrop* drop = nullptr;
    PGH_TRY(make_drop(g, rate, seed, &drop_store, &drop));
    PGH_TRY(mm_partial(g, xg.as<float>(), ld, b, sums.as<float>(), nullptr, drop));
    CombineParams c{};
    c.sums = sums.as<float>();
    c.dst_scale = f.dst_scale;
    c.y = yint.as<float>();
    c.plain = 1;
    {
        ProfScope prof(PGH_K_COMBINE);
        k_mm_combine<<<combine_grid(), WG, 0, r.stream>>>(c, n_int, ld, b, nullptr, partial.as<double>());
    }
    k_mm_permute_out<<<blocks_for(n_int * lanes_per_row(ld)), WG, 0, r.stream>>>(yint.as<float>(), f.perm, n_int, g->n_cols, b, ld, nullptr, y->data);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    return 0;
}

}  // namespace

// Batched PageRank: b independent runs of PageRank(alpha) with the SAME ConvergenceManager settings; column j stops
// at its own iteration (frozen afterwards), exactly as b calls of pgh_ppr_run would.
extern "C" int pgh_ppr_run_batch(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales,
                                 pgh_loop_result* results) {
    return batch_impl(g, p, ranks, cfg, out_scales, 0.0, 0, results);
}
// ... with graph_dropout (abstract_filters.py:61: a fresh mask for every step): step k of the batch multiplies by the matrix
// masked with seed0 + k - 1, the mask pgh_spmv_dropout(seed0 + k - 1) would apply
extern "C" int pgh_ppr_run_batch_dropout(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales,
                                         double rate, uint64_t seed0, pgh_loop_result* results) {
    return batch_impl(g, p, ranks, cfg, out_scales, rate, seed0, results);
}

namespace {
// what the host polls after every step: the tail of BatchState, copied to a pinned ring by the stream (no queue drain per look)
struct BatchPoll {
    int all_done, paused, executed, b;
};
constexpr int kPollRing = 8;
struct PollRing {
    BatchPoll* host = nullptr;       // pinned [kPollRing]
    hipEvent_t ev[kPollRing] = {nullptr};
};
int poll_ring(PollRing** out) {
    static PollRing ring;
    if (ring.host == nullptr) {
        PGH_HIP(hipHostMalloc(reinterpret_cast<void**>(&ring.host), sizeof(BatchPoll) * kPollRing, hipHostMallocDefault));
        for (int i = 0; i < kPollRing; ++i) PGH_HIP(hipEventCreateWithFlags(&ring.ev[i], hipEventDisableTiming));
    }
    *out = &ring;
    return 0;
}

int batch_impl(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales, double rate, uint64_t seed0,
               pgh_loop_result* results) {
    PGH_CHECK(g && p && ranks && cfg && results, "pgh_ppr_run_batch: null argument");
    PGH_CHECK(g->n_rows == g->n_cols && p->n == g->n_cols && ranks->n == g->n_cols && p->b == ranks->b, "pgh_ppr_run_batch: shape mismatch");
    PGH_CHECK(p->b >= 1 && p->b <= kLanes, "pgh_ppr_run_batch: the batch width must be in [1, 64]");
    PGH_CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    PGH_TRY(ensure_mm_layout(g));
    Runtime& r = rt();
    const BsfFormat& f = g->bsf_mm;
    const int b = p->b, ld = (b + 3) & ~3;
    const int64_t n = g->n_cols, n_int = f.n_out;
    const size_t slab = sizeof(float) * (size_t)n_int * ld;
    const f32x4* rowop = reinterpret_cast<const f32x4*>(f.mm_rowop);
    DevBytes pint, xg0, xg1, sums, partial, folded, state_mem, factors, row_flags, nz_maps, nz_counts, zero_row;
    PGH_TRY(zero_row.alloc(sizeof(float) * (size_t)(ld + 4)));
    PGH_HIP(hipMemsetAsync(zero_row.p, 0, sizeof(float) * (size_t)(ld + 4), r.stream));
    PGH_TRY(row_flags.alloc((size_t)((n_int + 63) / 64 * 64 + 64)));      // whole 64-row chunks (k_mm_step reads a chunk's flags as one line)
    // non-zero rows of the two gather slabs (k_mm_partial<SPARSE>): a byte per row, and {rows that hold a non-zero [2], rows in the run}
    const int64_t map_len = (n_int + 63) / 64 * 64 + 64;         // whole passes of 16 rows, 16-byte aligned halves
    PGH_TRY(nz_maps.alloc(2 * (size_t)map_len));
    const int in_grid = blocks_for(n_int * lanes_per_row(ld));
    PGH_TRY(nz_counts.alloc(sizeof(int) * (size_t)(4 + cgrid_of() + 2 * in_grid * (WG / 64))));
    uint8_t* nz_map[2] = {nz_maps.as<uint8_t>(), nz_maps.as<uint8_t>() + map_len};
    int* nz_cnt = nz_counts.as<int>();
    const bool sparse_gate = rate == 0.0 && !(getenv("PGH_MM_SPARSE") != nullptr && atoi(getenv("PGH_MM_SPARSE")) == 0);
    // rows without entries and without personalization are zero in every iterate when the loop starts from p (they are then
    // zero in the start iterate too): nobody reads or writes them after the way in
    const bool skip_dead = cfg->start_from_p != 0 && f.mm_row_has != nullptr;
    PGH_TRY(pint.alloc(slab));
    PGH_TRY(xg0.alloc(slab));
    PGH_TRY(xg1.alloc(slab));
    PGH_TRY(sums.alloc(slab));
    const int cgrid = combine_grid();
    PGH_TRY(partial.alloc(sizeof(double) * 4 * (size_t)cgrid * kLanes));
    PGH_TRY(folded.alloc(sizeof(double) * 7 * kLanes));                  // S, T, R', D of the step + the separate kernel's residual + the way in's {T0, sum p}
    PGH_TRY(state_mem.alloc(sizeof(BatchState)));
    PGH_TRY(factors.alloc(sizeof(double) * kLanes));
    BatchState* state = state_mem.as<BatchState>();
    PollRing* ring = nullptr;
    PGH_TRY(poll_ring(&ring));
    hipEvent_t ev_a, ev_b;
    PGH_HIP(hipEventCreate(&ev_a));
    PGH_HIP(hipEventCreate(&ev_b));
    PGH_HIP(hipEventRecord(ev_a, r.stream));
    if (!skip_dead) PGH_HIP(hipMemsetAsync(sums.p, 0, slab, r.stream));      // structural zeros of the rows without entries (every row is processed)
    // the in-kernel residual: sum rules, the plain matrix (a dropped matrix has other column sums every step), PGH_MM_FUSED=0 turns it off
    const bool fused_first = (cfg->err_kind == PGH_ERR_L1 || cfg->err_kind == PGH_ERR_MABS) && rate == 0.0 &&
                             !(getenv("PGH_MM_FUSED") != nullptr && atoi(getenv("PGH_MM_FUSED")) == 0);
    bool step1_predicted = false;                // the way in's sums give step 1 its predicted quotient (PGH_MM_FIRST_PRED=0: the separate kernel)
    k_mm_state_init<<<1, kLanes, 0, r.stream>>>(state, b);
    PGH_HIP(hipMemsetAsync(nz_maps.p, 0, 2 * (size_t)map_len, r.stream));       // rows nobody ever writes hold zeros: their bytes stay 0
    PGH_HIP(hipMemsetAsync(nz_counts.p, 0, sizeof(int) * 4, r.stream));
    int* nz_part = nz_cnt + 4;                                           // [cgrid] k_mm_step's per-workgroup counts
    int* in_part = nz_part + cgrid_of();                                 // [2][in_grid * 4] the way in's per-wavefront counts
    {
        PermuteIn2 q{};
        q.nz_map = nz_map[0];
        q.nz_rows = in_part;
        q.live_rows = in_part + in_grid * (WG / 64);
        q.src_p = p->data;
        q.src_x = cfg->start_from_p ? p->data : ranks->data;                  // abstract_filters.py:56 without / with warm_start
        q.out_p = pint.as<float>();
        q.out_xg0 = xg0.as<float>();
        q.out_xg1 = xg1.as<float>();
        q.rowop = rowop;
        q.row_flags = row_flags.as<uint8_t>();
        q.row_has = skip_dead ? f.mm_row_has : nullptr;                       // null: every row counts as holding entries
        // the first step's quotient is predicted from sums of the way in (no separate residual pass for step 1): its per-workgroup
        // partials borrow the step partials' buffer (2 x in_grid x 64 doubles <= 4 x cgrid x 64 as long as in_grid <= 2 cgrid)
        const bool first_pred = fused_first && in_grid <= 2 * cgrid && !(getenv("PGH_MM_FIRST_PRED") != nullptr && atoi(getenv("PGH_MM_FIRST_PRED")) == 0);
        q.pred_part = first_pred ? partial.as<double>() : nullptr;
        k_mm_permute_in2<<<in_grid, WG, 0, r.stream>>>(q, f.perm, n_int, n, b, ld);
        k_mm_count_fold<<<1, WG, 0, r.stream>>>(q.nz_rows, q.live_rows, in_grid * (WG / 64), nz_cnt + 0, nz_cnt + 2);
        if (first_pred) {
            double* fold0 = folded.as<double>();
            k_mm_fold1<<<kLanes, WG, 0, r.stream>>>(partial.as<double>(), in_grid, 0, state, fold0 + 5 * kLanes, 0);
            k_mm_fold1<<<kLanes, WG, 0, r.stream>>>(partial.as<double>() + (int64_t)in_grid * kLanes, in_grid, 0, state, fold0 + 6 * kLanes, 0);
            k_mm_first_pred<<<1, kLanes, 0, r.stream>>>(state, fold0 + 5 * kLanes, fold0 + 6 * kLanes, cfg->alpha, cfg->use_quotient);
        }
        step1_predicted = first_pred;
    }
    float* buf[2] = {xg0.as<float>(), xg1.as<float>()};
    double* fold = folded.as<double>();
    const int linf = cfg->err_kind == PGH_ERR_LINF;
    const int max_steps = cfg->max_iters - 1 > 0 ? cfg->max_iters - 1 : 0;
    bool fused = fused_first;
    int flags = fused ? 2 : 0;
    CloseParams cp{};
    cp.tol = cfg->tol;
    cp.alpha = cfg->alpha;
    cp.n_orig = n;
    cp.use_quotient = cfg->use_quotient;
    cp.err_kind = cfg->err_kind;
    auto check_of = [&](int k) { const int it = k + 1; return (cfg->err_kind != PGH_ERR_ITERS) && (it < cfg->max_iters) && (it % cfg->end_modulo == 0); };
    // the separate residual of step k + its close (first step, max rule, dropout, a paused step)
    auto close_plain = [&](int k, int mode, int resume) -> int {
        const int check = check_of(k) ? 1 : 0;
        if (check) {
            ProfScope prof(PGH_K_RESIDUAL);
            k_mm_residual2<<<cgrid, WG, 0, r.stream>>>(buf[k & 1], buf[(k - 1) & 1], rowop, n_int, ld, b, cfg->use_quotient, linf, state, fold,
                                                       partial.as<double>(), row_flags.as<uint8_t>(), resume);
            k_mm_fold1<<<kLanes, WG, 0, r.stream>>>(partial.as<double>(), cgrid, linf, state, fold + 4 * kLanes, resume);
        }
        CloseParams c2 = cp;
        c2.check = check;
        c2.mode = mode;
        c2.resume = resume;
        c2.nz_part = sparse_gate ? nz_part : nullptr;
        c2.nz_parts = cgrid;
        c2.nz_total = nz_cnt + (k & 1);
        k_mm_close2<<<1, kLanes, 0, r.stream>>>(state, fold, fold + 4 * kLanes, c2);
        PGH_HIP(hipGetLastError());
        return 0;
    };
    auto enqueue_step = [&](int k) -> int {
        MMDrop drop_store;
        const MMDrop* drop = nullptr;
        PGH_TRY(make_drop(g, rate, seed0 + (uint64_t)(k - 1), &drop_store, &drop));
        SparseGate gate{};
        if (sparse_gate) {
            gate.map = nz_map[(k - 1) & 1];
            gate.nz_rows = nz_cnt + ((k - 1) & 1);
            gate.live_rows = nz_cnt + 2;
        }
        PGH_TRY(mm_partial(g, buf[(k - 1) & 1], ld, b, sums.as<float>(), state, drop, sparse_gate ? &gate : nullptr));
        const int mode = (k == 1 && !(fused && step1_predicted)) ? 2 : (fused ? 1 : 0);
        StepParams c{};
        c.sums = sums.as<float>();
        c.rowop = rowop;
        c.p = pint.as<float>();
        c.row_flags = row_flags.as<uint8_t>();
        c.xg_old = buf[(k - 1) & 1];
        c.xg_new = buf[k & 1];
        c.alpha = cfg->alpha;
        c.mode = mode;
        c.nz_map = sparse_gate ? nz_map[k & 1] : nullptr;
        c.nz_part = sparse_gate ? nz_part : nullptr;
        c.nz_prev = nz_cnt + ((k - 1) & 1);
        c.live_rows = nz_cnt + 2;
        c.affine = getenv("PGH_MM_AFFINE") != nullptr && atoi(getenv("PGH_MM_AFFINE")) != 0;
        c.zero = zero_row.as<float>();
        {
            ProfScope prof(PGH_K_COMBINE);
            k_mm_step<false><<<cgrid, WG, 0, r.stream>>>(c, n_int, ld, b, state, partial.as<double>());
            if (sparse_gate) k_mm_step<true><<<cgrid, WG, 0, r.stream>>>(c, n_int, ld, b, state, partial.as<double>());
        }
        k_mm_fold4<<<dim3(kLanes, 4), WG, 0, r.stream>>>(partial.as<double>(), cgrid, state, fold);
        if (mode == 1) {
            CloseParams c2 = cp;
            c2.check = check_of(k) ? 1 : 0;
            c2.mode = 1;
            c2.nz_part = sparse_gate ? nz_part : nullptr;
            c2.nz_parts = cgrid;
            c2.nz_total = nz_cnt + (k & 1);
            k_mm_close2<<<1, kLanes, 0, r.stream>>>(state, fold, fold + 4 * kLanes, c2);
        } else {
            PGH_TRY(close_plain(k, mode, 0));
        }
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipMemcpyAsync(&ring->host[k % kPollRing], &state->all_done, sizeof(BatchPoll), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipEventRecord(ring->ev[k % kPollRing], r.stream));
        return 0;
    };
    // the host keeps up to three steps enqueued beyond the last one whose outcome it has seen; launches behind a stop or a pause are no-ops
    int enq = 0, seen = 0;
    bool stop = false;
    while (!stop) {
        while (enq < max_steps && enq - seen < 3) {
            PGH_TRY(enqueue_step(enq + 1));
            ++enq;
        }
        if (seen == enq) break;
        const int k = seen + 1;
        PGH_HIP(hipEventSynchronize(ring->ev[k % kPollRing]));
        const BatchPoll poll = ring->host[k % kPollRing];
        ++seen;
        if (poll.paused) {
            // step k ran but for its close; the steps enqueued behind it saw the pause and did nothing.  The separate kernel decides, the
            // run goes on without the fusion (a negative or non-finite value would pause every step)
            flags |= 1;
            fused = false;
            PGH_TRY(close_plain(k, 0, 1));
            PGH_HIP(hipMemcpyAsync(&ring->host[0], &state->all_done, sizeof(BatchPoll), hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
            enq = seen;
            stop = ring->host[0].all_done != 0;
        } else if (poll.all_done) {
            stop = true;
        }
    }
    BatchState host_state;
    PGH_HIP(hipMemcpyAsync(&host_state, state, sizeof(BatchState), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    // the slab row of a column that stopped early was copied forward unchanged by every later executed step
    double h_factors[kLanes];
    for (int j = 0; j < kLanes; ++j) h_factors[j] = j < b ? host_state.scale[j] * (out_scales ? out_scales[j] : cfg->out_scale) : 1.0;
    PGH_HIP(hipMemcpyAsync(factors.p, h_factors, sizeof(h_factors), hipMemcpyHostToDevice, r.stream));
    const int executed = host_state.executed;              // steps that ran before every column had stopped
    k_mm_permute_out2<<<blocks_for(n_int * lanes_per_row(ld)), WG, 0, r.stream>>>(buf[executed & 1], rowop, f.perm, n_int, n, b, ld, factors.as<double>(),
                                                                                    ranks->data, row_flags.as<uint8_t>());
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
        results[j].flags = flags;
    }
    return 0;
}
}  // namespace
