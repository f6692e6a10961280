// The f64 route of the reference's "chebyshev" recurrence on the blocked segment-flag stream.
//
// Reference counterpart: ClosedFormGraphFilter._recursion with coefficient_type "chebyshev" (abstract_filters.py:216-224)
// over conv(signal, M) = signal @ M (pygrank/core/backend/numpy.py:64-65).  Why f64: EpiPoly64 in pgh_spmv.hip (the
// recurrence S_k = (2 M^T - I) S_{k-1} amplifies every rounding of a term; an f32 evaluation cannot hold 1e-6).
//
// Round 2 ran this route over the row-major CSR(M^T) (k_spmv_merge<..., double, ...>: 1.3 ms per term at RMAT scale 23, 0.12 of
// the HBM roofline -- every one of the 134 M gathers is an 8-byte request to the L2 / fabric).  This file gives the route the
// layout decisions of the f32 stream (pgh_bsf.hip), with 8-byte operands:
//   * a second blocked image of the graph (pgh_graph_s::bsf64, built on first use like the multi-seed image): sources
//     relabelled by descending reference count, dealt to 8 XCD-affine column blocks, entries sorted by (block, row, col), a
//     flag bit opens every row segment, blocks padded to whole 512-entry wavefront tiles; rounds 3-5 left the cold entries in
//     the stream (an 8-byte gather through the XCD's L2 each); round 6 moves them into a propagation-blocking image of their
//     own (second half of this file), the stream is then hot-only with 2-byte words;
//   * k_bsf64_partial: one 1024-thread workgroup per CU; the first 20 224 doubles of the block's hot-first slice of the
//     gather vector fill the LDS (158 KB), the rest of the slice is gathered through the XCD's own L2 (a block's referenced
//     prefix is ~3.9 MB at scale 23: L2 hit rate 0.88); f64 segmented sums (DPP scans on register pairs), closed segments
//     leave as 8-byte stores into the compact partial-sum array, pieces that cross tiles as carries;
//   * k_bsf64_fixup closes the cross-tile segments in a fixed order; k_bsf64_combine folds a row's <= 8 block segments
//     (SegMeta, as in the f32 layout), applies the recurrence's epilogue in f64 and writes the next gather vector -- for
//     images without a cold tail; with one, the fix-ups ride in k_pb64_gather and k_pb64_finish runs the epilogue.
// Deterministic like the f32 path (the cold image's row sums are integer atomics: order-independent).
#include "pgh_kernels.h"

#include <cstdlib>

using namespace pgh;

namespace {

constexpr int kT = 64 * PGH_BSF_IPT;           // entries per wavefront tile (the tile table of bsf_build)
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kHot64 = 20224;                  // doubles of the gather vector cached in LDS per workgroup (the LDS holds nothing else)
constexpr int kLdsDoubles = kHot64 + 1;
// diagnostic builds only: 1 = no cold gathers, 2 = no hot-cache reads, 4 = no stores of a lane's later segments, 8 = no scans.
// Measured at RMAT scale 23 (profiles/r03/cheb_f64_blocked.log): 376 us as shipped; 190 without the cold gathers (a divergent
// 8-byte gather instruction costs the CU ~50 cycles WHATEVER the number of active lanes: 32 blocks leave 10 % of the lanes
// cold instead of 30 % and take the same time; exec-masked loads likewise), 335 without the stores, 377 without the LDS
// reads, 373 without the scans, 146 with all four off (stream + arithmetic).
#ifndef PGH_B64_PROBE
#define PGH_B64_PROBE 0
#endif
static_assert(PGH_BSF_IPT == 8, "a lane owns 8 consecutive entries (two 16-byte words)");
static_assert(kLdsDoubles * 8 + 16 * 8 <= 160 * 1024, "LDS budget of one CU");

struct View64 {
    const uint32_t* colf;        // [num_entries] source (new id) | bit 31 = first entry of a row segment
    const uint16_t* colf16;      // hot-only streams (every cold entry in the cold image): slot of the hot cache | bit 15 = first entry of a segment
    const float*    val;         // [num_entries] or null (value-free)
    const int4*     tile;        // {entry_start, entry_count, seg_base, chain_first}
    double*         tail;        // [num_tiles] piece of the segment still open at the end of the tile
    double*         head;        // [num_tiles] piece of the segment that was open when the tile started
    double*         psum;        // [num_segs + pad] compact block partial sums
    int             num_blocks;
    int             blk;
    int             hot;         // sources of every block in the LDS hot cache: kHot64 (PGH_HOT64: fewer, so that small test graphs have a cold tail)
    int             tile_begin[kMaxBlocks + 1];
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_inclusive_sum(int v) {
    v += dpp_i32<0x111, 0xf>(0, v);      // row_shr:1
    v += dpp_i32<0x112, 0xf>(0, v);      // row_shr:2
    v += dpp_i32<0x114, 0xf>(0, v);      // row_shr:4
    v += dpp_i32<0x118, 0xf>(0, v);      // row_shr:8
    v += dpp_i32<0x142, 0xa>(0, v);      // row_bcast:15 -> rows 1, 3
    v += dpp_i32<0x143, 0xc>(0, v);      // row_bcast:31 -> rows 2, 3
    return v;
}
// inclusive segmented sum: keep = 0 on lanes that start a new segment, 1 on lanes that continue the previous lane's
__device__ __forceinline__ double wave_segmented_sum64(int keep, double val) {
#define PGH_SEG64(CTRL, MASK)                                      \
    {                                                              \
        const double v2 = dpp_f64<CTRL, MASK>(0.0, val);           \
        const int k2 = dpp_i32<CTRL, MASK>(1, keep);               \
        val += keep ? v2 : 0.0;                                    \
        keep &= k2;                                                \
    }
    PGH_SEG64(0x111, 0xf)
    PGH_SEG64(0x112, 0xf)
    PGH_SEG64(0x114, 0xf)
    PGH_SEG64(0x118, 0xf)
    PGH_SEG64(0x142, 0xa)
    PGH_SEG64(0x143, 0xc)
#undef PGH_SEG64
    return val;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

PGH_STAMP_DECL(g_times_partial64)
// COLD = false (round 6): every cold entry of the image lives in its propagation-blocking image (below): no gather leaves the CU
// NARROW: the hot-only stream as 2-byte words (View64::colf16; COLD = false only)
template <bool HAS_VAL, bool COLD = true, bool NARROW = false>
__global__ __launch_bounds__(kThreads) void k_bsf64_partial(View64 f, const double* __restrict__ xg, const LoopState* __restrict__ state,
                                                            PendingClose pc) {
    __shared__ __attribute__((aligned(16))) double s_lds[kLdsDoubles];
    __shared__ double s_close[16];
    PGH_STAMP_BEGIN(g_times_partial64)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-affine block assignment: workgroups whose dispatch slots share blockIdx % 8 share an XCD (speed only)
    // (more than 8 blocks: an XCD's workgroups share out its num_blocks / 8 blocks -- every block's hot set is a different
    // 11 776 sources, so the LDS of the chip caches num_blocks x 11 776 of them)
    const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int b, rank, stride;
    if (f.num_blocks <= 8) {
        const int per = 8 / f.num_blocks;
        b = label % f.num_blocks;
        rank = (slot * per + label / f.num_blocks) * kWaves + wave;
        stride = (gridDim.x >> 3) * per * kWaves;
    } else {
        const int q = f.num_blocks >> 3;                   // blocks per XCD; (gridDim.x / 8) workgroups per XCD, a multiple of q
        b = label + 8 * (slot % q);
        rank = (slot / q) * kWaves + wave;
        stride = ((gridDim.x >> 3) / q) * kWaves;
    }
    const uint32_t base = (uint32_t)b * (uint32_t)f.blk;
    const uint32_t hot = (uint32_t)min(f.hot, f.blk);
    {
        // the hot cache: all of a thread's loads in flight at once (round 5: one 8-byte load, one wait, one LDS store per stride of 1024 was
        // twenty round trips one after the other at the head of every launch), and the loop-state test + the previous term's close (PendingClose)
        // between their issue and their arrival, on an LDS scratch of their own
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const double* __restrict__ src = xg + base;
        const f64x2* __restrict__ src2 = reinterpret_cast<const f64x2*>(src);      // (blocks start on multiples of 32 slots: 16-byte aligned)
        f64x2* __restrict__ dst2 = reinterpret_cast<f64x2*>(s_lds);
        const uint32_t hot2 = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) ? hot >> 1 : 0u;
        constexpr int FR = (kHot64 / 2 + kThreads - 1) / kThreads;
        f64x2 fr[FR];
        if (hot2 > 0) {
#pragma unroll
            for (int k = 0; k < FR; ++k) fr[k] = src2[min((uint32_t)tid + (uint32_t)k * kThreads, hot2 - 1)];
        }
        if (state != nullptr && state->done) return;
        if (pc.active && run_pending_close(pc, s_close)) return;
        if (hot2 > 0) {
#pragma unroll
            for (int k = 0; k < FR; ++k) {
                const uint32_t i = (uint32_t)tid + (uint32_t)k * kThreads;
                if (i < hot2) dst2[i] = fr[k];
            }
        }
        for (uint32_t i = (hot2 << 1) + tid; i < hot; i += kThreads) s_lds[i] = src[i];
        if (tid == 0) s_lds[hot] = 0.0;
    }
    __syncthreads();
    const int t_end = f.tile_begin[b + 1];
    int t = f.tile_begin[b] + rank;
    if (t >= t_end) return;

    static_assert(!NARROW || !COLD, "2-byte stream words address the hot cache only");
    struct Words {
        u32x4 c[NARROW ? 1 : 2];
        f32x4 v[2];
        int   seg_base;
    };
    auto load_words = [&](int tile, Words& w) __attribute__((always_inline)) {
        if (NARROW) {
            w.c[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(f.colf16 + (int64_t)tile * kT + lane * 8));
        } else {
            const u32x4* p = reinterpret_cast<const u32x4*>(f.colf + (int64_t)tile * kT + lane * 8);
            w.c[0] = __builtin_nontemporal_load(p);
            w.c[1] = __builtin_nontemporal_load(p + 1);
        }
        if (HAS_VAL) {
            const f32x4* q = reinterpret_cast<const f32x4*>(f.val + (int64_t)tile * kT + lane * 8);
            w.v[0] = __builtin_nontemporal_load(q);
            w.v[1] = __builtin_nontemporal_load(q + 1);
        }
        w.seg_base = f.tile[tile].z;
    };
    struct Gathered {
        double       c[COLD ? 8 : 1];   // from the block's cold slice (hot lanes address outside the buffer: 0, no memory access)
        uint32_t     at[4];          // slots of the LDS hot cache (cold lanes: its zero slot) as 16-bit pairs, read when the tile is summed
        float        v[HAS_VAL ? 8 : 1];
        unsigned int bits;
        int          seg_base;
    };
    // the block's cold slice [hot, blk) of the gather vector as a buffer: value = h + c, no select and no divergent branch
    const __amdgpu_buffer_rsrc_t cold_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(xg + base + hot), 0, (int)(((uint32_t)f.blk - hot) << 3), 0x00020000);
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    auto gather = [&](const Words& w, Gathered& g) __attribute__((always_inline)) {
        unsigned int bits = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (NARROW) {
                const uint32_t pair = w.c[0][k >> 1];
                const uint32_t half = (k & 1) ? pair >> 16 : pair & 0xffffu;
                bits |= (half >> 15) << k;
                if (!(k & 1)) g.at[k >> 1] = pair & 0x7fff7fffu;     // both slots of the pair at once
                if (HAS_VAL) g.v[k] = w.v[k >> 2][k & 3];
                continue;
            }
            const uint32_t word = w.c[k >> 2][k & 3];
            bits |= (word >> 31) << k;
            const uint32_t loc = (word & 0x7fffffffu) - base;
            const uint32_t slot16 = min(loc, hot);
            g.at[k >> 1] = (k & 1) ? (g.at[k >> 1] | (slot16 << 16)) : slot16;
            if (!COLD) {
            } else if (PGH_B64_PROBE & 1) g.c[k] = 0.0;
            else {
                const i32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(cold_rsrc, (int)((loc - hot) << 3), 0, 0);
                g.c[k] = __hiloint2double(raw.y, raw.x);
            }
            if (HAS_VAL) g.v[k] = w.v[k >> 2][k & 3];
        }
        g.bits = bits;
        g.seg_base = w.seg_base;
    };
    auto reduce = [&](const Gathered& g, int tile) __attribute__((always_inline)) {
        const unsigned int bits = g.bits;
        double h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t slot16 = (k & 1) ? (g.at[k >> 1] >> 16) : (g.at[k >> 1] & 0xffffu);
            h[k] = (PGH_B64_PROBE & 2) ? (double)slot16 : s_lds[slot16];
        }
        const int mine = __popc(bits);
        const int incl = wave_inclusive_sum(mine);         // flags in lanes <= this one
        const int before = incl - mine;
        // The j-th flag of the tile (j = before + flags seen in this lane) closes the running sum: j = 0 closes the piece of
        // the segment that was open when the tile started (-> head carry), j >= 1 closes segment j - 1 of the tile, whose sum
        // belongs at psum[seg_base + j].  Sums closed by a lane's second and later flags are complete and leave at once
        // (8-byte stores; neighbouring lanes write neighbouring slots); what its FIRST flag closes may have begun in
        // earlier lanes: it waits for the stitch below.  (The first version staged all of this through an LDS strip per
        // wavefront -- 18 more LDS instructions per tile than the 8 hot-cache reads, and the LDS is what bounds this kernel.)
        double* __restrict__ dst = f.psum + g.seg_base + before;
        int seen = 0;
        double acc = 0.0, head_sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((bits >> k) & 1u) {
                if (seen != 0 && !(PGH_B64_PROBE & 4)) dst[seen] = acc;
                else head_sum = acc;
                ++seen;
                acc = 0.0;
            }
            double x = COLD ? h[k] + g.c[k] : h[k];
            if (HAS_VAL) x *= (double)g.v[k];
            acc += x;
        }
        // stitch across lanes: what follows a lane's last flag continues into the next lanes up to their first flag
        const double val = (PGH_B64_PROBE & 8) ? acc : wave_segmented_sum64(mine ? 0 : 1, acc);
        const double ev = dpp_f64<0x138, 0xf>(0.0, val);   // wave_shr:1: what the earlier lanes hold of the segment this
        if (bits != 0u) {                                  // lane's first flag closes
            const double first = head_sum + ev;
            if (before == 0) f.head[tile] = first;
            else dst[0] = first;
        }
        if (lane == 63) f.tail[tile] = val;
    };

    // Software pipeline per wavefront (loads are issued unconditionally on clamped tile indices, clamped results are never
    // consumed).  What bounds the kernel is the number of 8-byte cold gathers the L2 has in flight, so the value-free shape
    // keeps the gathers of TWO tiles in flight behind the sums (stream words one tile ahead of the gathers); valued streams
    // carry 8 more registers per set and keep one.
    const int t_last = t_end - 1;
    if (!HAS_VAL) {
        Words wa, wb;
        Gathered g0, g1, g2;
        load_words(t, wa);
        load_words(min(t + stride, t_last), wb);
        gather(wa, g0);
        load_words(min(t + 2 * stride, t_last), wa);
        gather(wb, g1);
#define PGH_STEP64(WL, WG_, GN, GC)                          \
    load_words(min(t + 3 * stride, t_last), WL);             \
    gather(WG_, GN);                                         \
    reduce(GC, t);                                           \
    t += stride;                                             \
    if (t >= t_end) break;
        for (;;) {
            PGH_STEP64(wb, wa, g2, g0)
            PGH_STEP64(wa, wb, g0, g1)
            PGH_STEP64(wb, wa, g1, g2)
            PGH_STEP64(wa, wb, g2, g0)
            PGH_STEP64(wb, wa, g0, g1)
            PGH_STEP64(wa, wb, g1, g2)
        }
#undef PGH_STEP64
    } else {
        Words wa, wb;
        Gathered g0, g1;
        load_words(t, wa);
        load_words(min(t + stride, t_last), wb);
        gather(wa, g0);
        for (;;) {
            load_words(min(t + 2 * stride, t_last), wa);
            gather(wb, g1);
            reduce(g0, t);
            t += stride;
            if (t >= t_end) break;
            load_words(min(t + 2 * stride, t_last), wb);
            gather(wa, g0);
            reduce(g1, t);
            t += stride;
            if (t >= t_end) break;
        }
    }
    PGH_STAMP_END(g_times_partial64)
}

// hot-only stream words -> 2 bytes: slot of the block's hot cache (pad / sentinel entries: the zero slot `hot`) | flag << 15
__global__ void k_bsf64_narrow(const uint32_t* __restrict__ colf, int64_t entries, View64 f, uint32_t hot, uint16_t* __restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < entries; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e / kT);
        int b = 0;
        for (int k = 1; k < f.num_blocks; ++k) b += t >= f.tile_begin[k] ? 1 : 0;
        const uint32_t word = colf[e];
        const uint32_t loc = (word & 0x7fffffffu) - (uint32_t)b * (uint32_t)f.blk;
        out[e] = (uint16_t)(min(loc, hot) | ((word >> 31) << 15));
    }
}

// where the fix-up of tile t goes (index into psum, -1 = nothing to fix): the segment open at the tile start closes here
__global__ void k_bsf64_fixlist(const int4* __restrict__ tile, const int32_t* __restrict__ seg_row, int num_tiles, int32_t* __restrict__ fix_seg) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < num_tiles; t += gridDim.x * blockDim.x) {
        const int4 ti = tile[t];
        fix_seg[t] = (ti.w >= 0 && ti.z >= 0 && seg_row[ti.z] >= 0) ? ti.z : -1;
    }
}

// segments that cross tiles: carries of the chain's tiles in ascending order + the head piece (fixed order); chains of 32+
// tiles (hub rows) are summed by the whole wavefront
struct Fix64 {
    const int32_t* fix_seg;
    const int4*    tile;
    const double*  tail;
    const double*  head;
    double*        psum;
    int            num_tiles;
};
// (a device body: its own launch, or the first instructions of phase A of the cold image -- it touches nothing that kernel reads, and the
// finishing pass is the first to read its results)
__device__ __forceinline__ void bsf64_fixup_tiles(const Fix64& f, const int begin, const int stride) {
    const int32_t* __restrict__ fix_seg = f.fix_seg;
    const int4* __restrict__ tile = f.tile;
    const double* __restrict__ tail = f.tail;
    const double* __restrict__ head = f.head;
    double* __restrict__ psum = f.psum;
    const int num_tiles = f.num_tiles;
    const int lane = threadIdx.x & 63;
    for (int t0 = begin; t0 < num_tiles; t0 += stride) {
        const int t = t0 + (int)threadIdx.x;
        const int dst = t < num_tiles ? fix_seg[t] : -1;
        const int first = dst >= 0 ? tile[t].w : 0;
        const int len = dst >= 0 ? t - first : 0;
        const bool is_long = len >= 32;
        if (dst >= 0 && !is_long) {
            double total = 0.0;
            for (int s = first; s < t; ++s) total += tail[s];
            psum[dst] = total + head[t];
        }
        unsigned long long todo = __ballot(is_long);
        while (todo != 0ULL) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ULL;
            const int c_first = __shfl(first, src, 64), c_t = __shfl(t, src, 64);
            double part = 0.0;
            for (int s = c_first + lane; s < c_t; s += 64) part += tail[s];
            part = wave_reduce_sum(part);
            const double total = __shfl(part, 0, 64) + head[c_t];
            if (lane == src) psum[dst] = total;
        }
    }
}
__global__ __launch_bounds__(WG) void k_bsf64_fixup(Fix64 f, const LoopState* __restrict__ state) {
    if (state != nullptr && state->done) return;
    bsf64_fixup_tiles(f, blockIdx.x * WG, gridDim.x * WG);
}

struct Epi64 {
    double        a, b, c;       // term_out = a * (M^T term) + b * term;  result += c * term_out
    const double* term;
    double*       term_out;
    double*       r;
    double*       xg_out;        // next gather vector: term_out * src_scale
    const float*  src_scale;     // or null
    const float*  dst_scale;     // or null
    int           err_linf;
    // per-row weights of the recursive filters in f64 (round 6; AbsorbingWalks adhoc.py:166-169, SymmetricAbsorbingRandomWalks :362-364):
    // term_out = a * row_w * (M^T term) + b * term, and the next gather vector carries src_w (the walk's pre-scale of the iterate)
    const double* row_w;         // or null (= 1)
    const double* src_w;         // or null (= 1)
};

// rows [iso_from, blk) of every block are isolated (no entries, referenced by nobody); flag = 0: the personalization is zero
// on all of them, so they stay zero in every term and are passed over
struct IsoRows {
    const int* flag;
    int        blk;
    int        iso_from;
};

// Round 5: the row loop without run-time branches around its loads.  The round-3 form tested `mask & bit` before every one of a row's
// eight segment loads, `dst_scale != nullptr` / `b != 0` before the operand loads: ten basic blocks with a drained wait each, ten memory
// round trips per row one after the other (114 us for 253 MB).  Here a missing segment reads the zero slot of the partial sums, a missing
// operand reads it too (its eight zero bytes are a float 0 as well) and is dropped by a select: two round trips per row (map words,
// then everything else).  PGH_B64_COMBINE_BF=0: the old form.
#ifndef PGH_B64_COMBINE_BF
#define PGH_B64_COMBINE_BF 1
#endif
__global__ __launch_bounds__(WG) void k_bsf64_combine(const SegMeta* __restrict__ meta, int64_t words, const double* __restrict__ psum,
                                                       int num_blocks, int64_t n_out, Epi64 ep, const LoopState* __restrict__ state,
                                                       double* __restrict__ partial_sum, double* __restrict__ partial_delta, IsoRows iso,
                                                       unsigned int zero_at) {
    __shared__ double s_red[4];
    if (state != nullptr && state->done) return;
    double sum_y = 0.0, delta = 0.0;
    const int64_t stride = (int64_t)gridDim.x * WG;
    const bool skip_iso = iso.flag != nullptr && *iso.flag == 0;
    if (PGH_B64_COMBINE_BF && num_blocks <= 8) {
        const bool has_ds = ep.dst_scale != nullptr, has_src = ep.src_scale != nullptr, has_b = ep.b != 0.0;
        const char* const zero_base = reinterpret_cast<const char*>(psum + zero_at);
        const char* const ds_base = has_ds ? reinterpret_cast<const char*>(ep.dst_scale) : zero_base;
        const char* const src_base = has_src ? reinterpret_cast<const char*>(ep.src_scale) : zero_base;
        const char* const term_base = has_b ? reinterpret_cast<const char*>(ep.term) : zero_base;
        const bool has_rw = ep.row_w != nullptr, has_sw = ep.src_w != nullptr;
        const char* const rw_base = has_rw ? reinterpret_cast<const char*>(ep.row_w) : zero_base;
        const char* const sw_base = has_sw ? reinterpret_cast<const char*>(ep.src_w) : zero_base;
        for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n_out; i += stride) {
            if (skip_iso && (int)((uint32_t)i % (uint32_t)iso.blk) >= iso.iso_from) continue;
            const int64_t w = i >> 6;
            const unsigned int lane_bit = 1u << (i & 31);
            const bool upper = (i & 32) != 0;
            // first round trip: the map words of the eight blocks (blocks past num_blocks: word 0 of block 0, dropped below), and
            // every operand that does not depend on them
            SegMeta m[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) m[b] = meta[b < num_blocks ? (int64_t)b * words + w : 0];
            const int64_t at4 = i << 2, at8 = i << 3;
            const float dsc = *reinterpret_cast<const float*>(ds_base + (has_ds ? at4 : 0));
            const float ssc = *reinterpret_cast<const float*>(src_base + (has_src ? at4 : 0));
            const double tv = *reinterpret_cast<const double*>(term_base + (has_b ? at8 : 0));
            const double rwv = *reinterpret_cast<const double*>(rw_base + (has_rw ? at8 : 0));
            const double swv = *reinterpret_cast<const double*>(sw_base + (has_sw ? at8 : 0));
            const double r_old = ep.r[i];
            // second: the row's segment in every block
            double v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned int lo = (unsigned int)m[b].mask, hi = (unsigned int)(m[b].mask >> 32);
                const unsigned int word = upper ? hi : lo;
                const unsigned int before = upper ? __popc(lo) + __popc(hi & (lane_bit - 1u)) : __popc(lo & (lane_bit - 1u));
                const bool has = b < num_blocks && (word & lane_bit) != 0u;
                v[b] = psum[has ? (unsigned int)m[b].base + before : zero_at];
            }
            double sum = 0.0;
#pragma unroll
            for (int b = 0; b < 8; ++b) sum += v[b];
            sum = has_ds ? sum * (double)dsc : sum;
            double y = ep.a * sum;
            y = has_rw ? y * rwv : y;
            y = has_b ? y + ep.b * tv : y;
            ep.term_out[i] = y;
            double gv = has_src ? y * (double)ssc : y;
            gv = has_sw ? gv * swv : gv;
            ep.xg_out[i] = gv;
            sum_y += y;
            const double r_new = r_old + ep.c * y;
            ep.r[i] = r_new;
            const double d = fabs(r_new - r_old);
            delta = ep.err_linf ? fmax(delta, d) : delta + d;
        }
    } else {
        for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n_out; i += stride) {
            if (skip_iso && (int)((uint32_t)i % (uint32_t)iso.blk) >= iso.iso_from) continue;
            const int64_t w = i >> 6;
            const unsigned long long bit = 1ULL << (i & 63);
            // the row's segment of every column block, eight blocks at a time: map words first, then the sums (block order)
            double s = 0.0;
            for (int b0 = 0; b0 < num_blocks; b0 += 8) {
                SegMeta m[8];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (b0 + b < num_blocks) m[b] = meta[(int64_t)(b0 + b) * words + w];
                double v[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    v[b] = 0.0;
                    if (b0 + b < num_blocks && (m[b].mask & bit)) v[b] = psum[m[b].base + __popcll(m[b].mask & (bit - 1ULL))];
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) s += v[b];
            }
            if (ep.dst_scale != nullptr) s *= (double)ep.dst_scale[i];
            double y = ep.a * s;
            if (ep.row_w != nullptr) y *= ep.row_w[i];
            if (ep.b != 0.0) y += ep.b * ep.term[i];
            ep.term_out[i] = y;
            double gv = ep.src_scale != nullptr ? y * (double)ep.src_scale[i] : y;
            if (ep.src_w != nullptr) gv *= ep.src_w[i];
            ep.xg_out[i] = gv;
            sum_y += y;
            const double r_old = ep.r[i];
            const double r_new = r_old + ep.c * y;
            ep.r[i] = r_new;
            const double d = fabs(r_new - r_old);
            delta = ep.err_linf ? fmax(delta, d) : delta + d;
        }
    }
    const double bs = block_reduce_256<0>(sum_y, s_red);
    if (threadIdx.x == 0) partial_sum[blockIdx.x] = bs;
    const double bd = ep.err_linf ? block_reduce_256<1>(delta, s_red) : block_reduce_256<0>(delta, s_red);
    if (threadIdx.x == 0) partial_delta[blockIdx.x] = bd;
}

// caller-space f32 personalization -> internal-space f64 vectors of the loop: term_0 = p, result_1 = c1 * p, gather = p * src_scale
__global__ void k_bsf64_bring(const float* __restrict__ p, const int32_t* __restrict__ perm, const float* __restrict__ src_scale,
                              int64_t n_int, int64_t n_valid, double c1, double* __restrict__ term, double* __restrict__ res,
                              double* __restrict__ xg, int* __restrict__ iso_flag, int blk, int iso_from) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_int; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = perm ? perm[i] : (i < n_valid ? i : -1);
        const double v = o >= 0 ? (double)p[o] : 0.0;
        if (iso_flag != nullptr && v != 0.0 && (int)((uint32_t)i % (uint32_t)blk) >= iso_from) atomicOr(iso_flag, 1);
        term[i] = v;
        res[i] = c1 * v;
        xg[i] = src_scale != nullptr ? v * (double)src_scale[i] : v;
    }
}

// one term of the recurrence as an f32 column of a row-major [n, ld] slab in the caller's ids (the chebyshev slab of
// optimization_dict users: filters._PowerSlab)
__global__ void k_bsf64_take_col(const double* __restrict__ vec, const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid,
                                 float* __restrict__ mat, int ld, int col) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_int; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = perm ? perm[i] : (i < n_valid ? i : -1);
        if (o >= 0) mat[o * ld + col] = (float)vec[i];
    }
}

__global__ void k_bsf64_take(const double* __restrict__ res, const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid, double factor,
                             float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_int; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = perm ? perm[i] : (i < n_valid ? i : -1);
        if (o >= 0) out[o] = (float)(res[i] * factor);
    }
}

// ------------------------------------------------------------------------------------------------- the cold tail of the f64 image (round 6)
// What bounded k_bsf64_partial were its cold gathers: 47 M 8-byte requests per term at RMAT scale 23, each answered with a 128-byte line
// from the XCD's L2 (186 of its 373 us).  The cure is the f32 path's (pgh_pb.hip): the cold entries leave the stream for a
// propagation-blocking image of their own -- the SAME builder (pb_plan / pb_build: bins, cells, both orders, the work list of the finishing
// pass) with this image's parameters: the f64 hot cache's 20 224 sources per block stay in the stream, a chunk of phase A holds 16 384
// sources (128 KB of doubles in LDS), and the values handed from A to B are doubles:
//   k_pb64_gather   phase A: per group of 8 entries 16 B of source indices + 4 B of target group read, 8 LDS gathers of 8 bytes, 64 B
//                   written to the group's place in B order; max |value| of the launch as a 64-bit pattern
//   k_pb64_finish   phase B + the recurrence's epilogue (what k_bsf64_combine does for images without a cold tail) over the work list
//                   of PbFormat: a bin's row sums live in LDS as 64-bit fixed point, every entry is one integer LDS atomic
//                   (order-independent: deterministic); the scale is a power of two per bin from the launch's largest value and the
//                   bin's largest row: an entry keeps min(51, 62 - log2(largest row)) bits below that value -- an absolute error of
//                   2^-52 of the launch's largest gathered value per entry, where a sum of doubles carries 2^-53 of its largest term.
//                   Hub rows are summed in f64 registers, the pieces of a split row folded in index order by the last arriver.
// ~20.5 sequential bytes per cold entry instead of a 128-byte line.  PGH_PB64=0 keeps the cold entries in the stream (round 5's route).
constexpr int kChunk64 = 16384;
#ifndef PGH_PB64_P
#define PGH_PB64_P 4          // groups per lane and round of phase A on long pieces
#endif
#ifndef PGH_PB64_FP
#define PGH_PB64_FP 1         // groups per thread and stream round of the finishing pass
#endif
#ifndef PGH_PB64_G
#define PGH_PB64_G 1          // groups of 64 rows per wavefront in flight in the finishing pass's epilogue (2: 44 bytes of scratch, 181 against 172 us)
#endif
constexpr int kGather64Threads = 1024;
typedef double f64x2 __attribute__((ext_vector_type(2)));

// tmp: the four 16-byte pairs of a group live in four PLANES of every block of 64 groups (pb_tmp_quad's layout for doubles): a wavefront's
// store (phase A) or load (phase B) of pair j covers the contiguous 16-byte slots of its consecutive groups instead of 16 bytes out of
// every 64 -- a quarter of the lines per instruction
__device__ __forceinline__ int64_t pb64_pair(uint32_t group, int j) {
    return ((int64_t)(group >> 6) << 9) + (j << 7) + ((group & 63u) << 1);
}

struct Pb64View {
    const uint16_t*     sloc;
    const float*        val;
    const uint32_t*     dstg;
    const int4*         task;
    const int*          task_range;
    const int4*         first_task;
    double*             tmp;
    unsigned long long* amax;
    const int4*         item_a;
    const int4*         item_b;
    const int*          sched;
    const int*          sched_begin;
    const int4*         first_a;
    const int4*         first_b;
    const int*          first_item;
    uint32_t*           work_counter;
    int                 tail_begin, tail_count;
    uint32_t*           hub_ticket;
    double*             hub_part;
    const uint16_t*     drow;
    const int*          iso_flag;
    int64_t             cold_prefix[9];
    int64_t             xg_base[8];
    int                 num_blocks, hot, chunk, short_piece;
    int64_t             num_cold;
};

// one piece of the A-order stream (whole groups of 8 entries inside the chunk whose slice of the gather vector sits in s_x); the shape of
// pb_stream_piece (pgh_pb_gather.h): loads of round i + 1 issued before the gathers and stores of round i, clamped unconditional loads
template <bool HAS_VAL, int P>
__device__ __forceinline__ void pb64_stream_piece(const double* __restrict__ s_x, const Pb64View& f, const int64_t body_begin, const int64_t body_end,
                                                  unsigned long long& amax) {
    struct Round {
        u16x8    s8[P];
        uint32_t to[P];
        f32x4    w0[HAS_VAL ? P : 1], w1[HAS_VAL ? P : 1];
    };
    constexpr int kRound = kGather64Threads * 8 * P;
    const int span = (int)(body_end - body_begin);
    const int last = span - 8;
    const uint16_t* __restrict__ sl = f.sloc + body_begin;
    const uint32_t* __restrict__ dg = f.dstg + (body_begin >> 3);
    const float* __restrict__ vl = HAS_VAL ? f.val + body_begin : nullptr;
    auto fetch = [&](Round& r, int rb) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int e = min(rb + (int)threadIdx.x * 8 + q * (kGather64Threads * 8), last);
            r.s8[q] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(sl + e));
            r.to[q] = __builtin_nontemporal_load(dg + (e >> 3));
            if (HAS_VAL) {
                r.w0[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vl + e));
                r.w1[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vl + e + 4));
            }
        }
    };
    auto emit = [&](const Round& r, int rb) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (rb + (int)threadIdx.x * 8 + q * (kGather64Threads * 8) > last) continue;
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = s_x[r.s8[q][k]];
                if (HAS_VAL) v[k] *= (double)(k < 4 ? r.w0[q][k & 3] : r.w1[q][k & 3]);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v[k]) & 0x7fffffffffffffffULL;
                amax = bits > amax ? bits : amax;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f64x2*>(f.tmp + pb64_pair(r.to[q], j)) = f64x2{v[2 * j], v[2 * j + 1]};
        }
    };
    Round r0, r1;
    int rb = 0;
    fetch(r0, rb);
    for (;;) {
        fetch(r1, rb + kRound);
        emit(r0, rb);
        rb += kRound;
        if (rb >= span) break;
        fetch(r0, rb + kRound);
        emit(r1, rb);
        rb += kRound;
        if (rb >= span) break;
    }
}

PGH_STAMP_DECL(g_times_gather64)
template <bool HAS_VAL>
__global__ __launch_bounds__(kGather64Threads) void k_pb64_gather(Pb64View f, const double* __restrict__ xg, const LoopState* __restrict__ state,
                                                                  Fix64 fix) {
    __shared__ __attribute__((aligned(16))) double s_x[kChunk64];
    __shared__ unsigned long long s_amax;
    if (state != nullptr && state->done) return;
    PGH_STAMP_BEGIN(g_times_gather64)
    // the cross-tile fix-ups of the blocked stream ride along (one launch and one dependent boundary fewer per term)
    bsf64_fixup_tiles(fix, blockIdx.x * kGather64Threads, gridDim.x * kGather64Threads);
    if (threadIdx.x == 0) s_amax = 0ULL;
    unsigned long long amax = 0ULL;
    const int piece_begin = f.task_range[blockIdx.x], piece_end = f.task_range[blockIdx.x + 1];
    const int4 first_task = f.first_task[blockIdx.x];
    int loaded = -1;
    for (int piece = piece_begin; piece < piece_end; ++piece) {
        const int4 task = piece == piece_begin ? first_task : f.task[piece];
        if (task.x != loaded) {
            __syncthreads();
            // cold ids [first_id, first_id + chunk) -> their doubles in the gather vector, block by block, as LDS-direct loads of 4-byte
            // words (a wavefront's instruction copies 256 contiguous bytes; the whole slice is in flight at once)
            const int64_t first_id = (int64_t)task.x * f.chunk;
            const int64_t last_id = min(first_id + f.chunk, f.num_cold);
            float* __restrict__ s_w = reinterpret_cast<float*>(s_x);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b >= f.num_blocks) continue;
                const int64_t lo = max(first_id, f.cold_prefix[b]), hi = min(last_id, f.cold_prefix[b + 1]);
                if (lo >= hi) continue;                     // wavefront-uniform
                const float* __restrict__ src = reinterpret_cast<const float*>(xg + f.xg_base[b] + f.hot - f.cold_prefix[b]);   // src[2 id] = low word of cold id
                const int64_t lo2 = 2 * lo, hi2 = 2 * hi, first2 = 2 * first_id;
                for (int64_t w0 = lo2 + (threadIdx.x & ~63); w0 < hi2; w0 += kGather64Threads) {
                    const int64_t at = w0 + (threadIdx.x & 63);
                    if (at < hi2) __builtin_amdgcn_global_load_lds(src + at, (__attribute__((address_space(3))) void*)(s_w + (w0 - first2)), 4, 0, 0);
                }
            }
            __syncthreads();
            loaded = task.x;
        }
        const int64_t body_begin = task.y, body_end = task.z;
        if (body_end <= body_begin) continue;
        if (body_end - body_begin >= (int64_t)f.short_piece) pb64_stream_piece<HAS_VAL, HAS_VAL ? 2 : PGH_PB64_P>(s_x, f, body_begin, body_end, amax);
        else pb64_stream_piece<HAS_VAL, 1>(s_x, f, body_begin, body_end, amax);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long other = (unsigned long long)__shfl_xor((long long)amax, d, 64);
        amax = other > amax ? other : amax;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && amax != 0ULL) atomicMax(&s_amax, amax);
    __syncthreads();
    if (threadIdx.x == 0 && s_amax != 0ULL) atomicMax(f.amax, s_amax);
    PGH_STAMP_END(g_times_gather64)
}

struct Rows64 {
    const SegMeta* meta;
    int64_t        words;
    const double*  psum;
    int            num_blocks;
    unsigned int   zero_at;      // a slot of psum that holds 0.0 for good
};

PGH_STAMP_DECL(g_times_finish64)
template <int ROWS, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void k_pb64_finish(Pb64View f, Rows64 rs, Epi64 ep, const LoopState* __restrict__ state,
                                                             double* __restrict__ partial_sum, double* __restrict__ partial_delta) {
    constexpr int WAVES = THREADS / 64;
    constexpr int NB = 8;
    constexpr int WORDS = ROWS / 64 + 1;                   // map words an item can touch per block (unaligned first row)
    __shared__ unsigned long long s_row[ROWS];
    __shared__ unsigned long long s_mask[NB * WORDS];
    __shared__ int s_base[NB * WORDS];
    __shared__ double s_red[2 * WAVES];
    __shared__ double s_hub;
    __shared__ int s_last;
    __shared__ int s_next;
    // start-up words: asked for together, used afterwards
    const unsigned long long amax_raw = __builtin_nontemporal_load(f.amax);
    const int sb0_raw = f.sched_begin[blockIdx.x], sb1_raw = f.sched_begin[blockIdx.x + 1];
    const int4 fa_raw = f.first_a[blockIdx.x], fb_raw = f.first_b[blockIdx.x];
    const int fi_raw = f.first_item[blockIdx.x];
    const int iso_raw = f.iso_flag != nullptr ? *f.iso_flag : 1;
    if (state != nullptr && state->done) return;
    PGH_STAMP_BEGIN(g_times_finish64)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni4 = [&](const int4& v) __attribute__((always_inline)) { return make_int4(uni(v.x), uni(v.y), uni(v.z), uni(v.w)); };
    const unsigned long long amax = ((unsigned long long)(unsigned int)uni((int)(amax_raw >> 32)) << 32) | (unsigned int)uni((int)amax_raw);
    const bool finite = amax < 0x7ff0000000000000ULL;      // inf / NaN among the values: the sums are not representable
    // |value| <= amax < 2^e (values below 2^-960 count as that: what they lose lies 2^-1011 below anything a result can show)
    const int e = max((int)(amax >> 52) - 1022, -960);
    constexpr double kMagic = 6755399441055744.0;          // 1.5 * 2^52: fma(v, S, magic) holds round(v * S) in its low bits
    double sum_y = 0.0, delta = 0.0;
    const bool has_ds = ep.dst_scale != nullptr, has_src = ep.src_scale != nullptr, has_b = ep.b != 0.0;
    const bool has_rw = ep.row_w != nullptr, has_sw = ep.src_w != nullptr;
    const char* const zero_base = reinterpret_cast<const char*>(rs.psum + rs.zero_at);
    const char* const ds_base = has_ds ? reinterpret_cast<const char*>(ep.dst_scale) : zero_base;
    const char* const src_base = has_src ? reinterpret_cast<const char*>(ep.src_scale) : zero_base;
    const char* const term_base = has_b ? reinterpret_cast<const char*>(ep.term) : zero_base;
    const char* const rw_base = has_rw ? reinterpret_cast<const char*>(ep.row_w) : zero_base;
    const char* const sw_base = has_sw ? reinterpret_cast<const char*>(ep.src_w) : zero_base;
    struct Ops {
        float  dsc, ssc;
        double tv, rwv, swv, r_old;
    };
    // the operands of one row (a missing operand reads the zero slot: no load under a run-time branch)
    auto load_ops = [&](int row) __attribute__((always_inline)) {
        Ops o;
        const int64_t at4 = (int64_t)row << 2, at8 = (int64_t)row << 3;
        o.dsc = *reinterpret_cast<const float*>(ds_base + (has_ds ? at4 : 0));
        o.ssc = *reinterpret_cast<const float*>(src_base + (has_src ? at4 : 0));
        o.tv = *reinterpret_cast<const double*>(term_base + (has_b ? at8 : 0));
        o.rwv = *reinterpret_cast<const double*>(rw_base + (has_rw ? at8 : 0));
        o.swv = *reinterpret_cast<const double*>(sw_base + (has_sw ? at8 : 0));
        o.r_old = ep.r[row];
        return o;
    };
    // k_bsf64_combine's epilogue for one row whose sum over the column blocks and the cold image is `sum`
    auto apply = [&](int row, double sum, const Ops& o, bool live) __attribute__((always_inline)) {
        sum = has_ds ? sum * (double)o.dsc : sum;
        double y = ep.a * sum;
        y = has_rw ? y * o.rwv : y;
        y = has_b ? y + ep.b * o.tv : y;
        double gv = has_src ? y * (double)o.ssc : y;
        gv = has_sw ? gv * o.swv : gv;
        const double r_new = o.r_old + ep.c * y;
        if (live) {
            ep.term_out[row] = y;
            ep.xg_out[row] = gv;
            ep.r[row] = r_new;
        }
        const double d = live ? fabs(r_new - o.r_old) : 0.0;
        sum_y += live ? y : 0.0;
        delta = ep.err_linf ? fmax(delta, d) : delta + d;
    };
    // the row's segments in the column blocks, straight from the map (hub rows: one thread, one row)
    auto block_row_sum = [&](int row) __attribute__((always_inline)) {
        const int64_t w = row >> 6;
        const unsigned long long bit = 1ULL << (row & 63);
        double s = 0.0;
        for (int b = 0; b < rs.num_blocks; ++b) {
            const SegMeta m = rs.meta[(int64_t)b * rs.words + w];
            if (m.mask & bit) s += rs.psum[m.base + __popcll(m.mask & (bit - 1ULL))];
        }
        return s;
    };
    constexpr int FP = PGH_PB64_FP;
    struct Round {
        u16x8 r8[FP];
        f64x2 v[FP][4];
    };
    // groups tid + (round * FP + q) * THREADS of the bin (pad rows 0xffff beyond the bin's range)
    auto fetch = [&](const int4& bin, int round, Round& R) __attribute__((always_inline)) {
        const bool hub = ((bin.y >> 21) & 1) != 0;
        const int groups = finite || hub ? bin.w : 0;
#pragma unroll
        for (int q = 0; q < FP; ++q) {
            const int g = tid + (round * FP + q) * THREADS;
            const bool ok = g < groups;
            const uint32_t grp = (uint32_t)(bin.z + g);
            R.r8[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(f.drow + (int64_t)grp * 8))
                         : u16x8{0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                R.v[q][j] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(f.tmp + pb64_pair(grp, j))) : f64x2{0.0, 0.0};
        }
        asm volatile("; stream round issued");
    };
    auto flush = [&](int where) __attribute__((always_inline)) {
        const double v0 = wave_reduce_sum(sum_y);
        const double v1 = ep.err_linf ? wave_reduce_max(delta) : wave_reduce_sum(delta);
        if (lane == 0) {
            s_red[wave] = v0;
            s_red[WAVES + wave] = v1;
        }
        __syncthreads();
        if (tid == 0) {
            double t0 = 0.0, t1 = 0.0;
            for (int w = 0; w < WAVES; ++w) {
                t0 += s_red[w];
                t1 = ep.err_linf ? fmax(t1, s_red[WAVES + w]) : t1 + s_red[WAVES + w];
            }
            partial_sum[where] = t0;
            partial_delta[where] = t1;
        }
        __syncthreads();
        sum_y = 0.0;
        delta = 0.0;
    };
    // item schedule: k_pb_finish's (pgh_pb.hip) -- a static slice per workgroup, the tail of the list handed out by a device counter;
    // partial slot = the workgroup's (static items) or the tail item's own, so that the fold does not depend on who processed what
    const int tail_count = f.tail_count;
    int at = uni(sb0_raw);
    const int at_end = uni(sb1_raw);
    int slot = blockIdx.x, item = -1;
    bool in_tail = at >= at_end;
    auto take_tail = [&]() __attribute__((always_inline)) {
        if (tail_count == 0) return -1;
        __syncthreads();
        if (tid == 0) s_next = (int)atomicAdd(f.work_counter, 1u);
        __syncthreads();
        const int k = uni(s_next);
        if (k >= tail_count) return -1;
        slot = gridDim.x + k;
        return uni(f.sched[f.tail_begin + k]);
    };
    bool flushed_head = false;
    int4 bin = make_int4(0, 0, 0, 0), epi = make_int4(0, 0, -1, 0);
    Round R;
    if (!in_tail) {
        item = uni(fi_raw);
        bin = uni4(fa_raw);
        epi = uni4(fb_raw);
    } else {
        flush(blockIdx.x);
        flushed_head = true;
        item = take_tail();
        if (item >= 0) {
            bin = uni4(f.item_a[item]);
            epi = uni4(f.item_b[item]);
        }
    }
    if (item >= 0) fetch(bin, 0, R);
    const bool skip_iso = f.iso_flag != nullptr && uni(iso_raw) == 0;
    while (item >= 0) {
        if (epi.z == -2) {
            // isolated rows: no entry, no segment in any block.  Passed over while the run's operands are zero there; otherwise the
            // epilogue of an empty row sum
            if (!skip_iso) {
                const int row_end = epi.x + epi.y;
                for (int row0 = epi.x + tid; row0 < row_end; row0 += THREADS) {
                    const Ops o = load_ops(row0);
                    apply(row0, 0.0, o, true);
                }
            }
            epi.y = 0;
        }
        const int rows = bin.y & 0xffff, count_bits = (bin.y >> 16) & 0x1f;
        const bool hub = ((bin.y >> 21) & 1) != 0;
        const int pieces = (int)((unsigned)bin.y >> 22) + 1;
        const int E = min(51, 62 - count_bits);
        const double S = __longlong_as_double((long long)(1023 + E - e) << 52);
        const double inv_S = __longlong_as_double((long long)(1023 - E + e) << 52);
        // the item's slice of the row -> segment map: asked for now, parked in LDS after the stream
        const int word0 = epi.x >> 6;
        const int words = (hub || epi.y <= 0) ? 0 : ((epi.x + epi.y - 1) >> 6) - word0 + 1;
        constexpr int MPT = (NB * WORDS + THREADS - 1) / THREADS;
        SegMeta mreg[MPT];
#pragma unroll
        for (int u = 0; u < MPT; ++u) {
            const int j = tid + u * THREADS;
            const int b = j / WORDS, w = j - b * WORDS;
            mreg[u].mask = 0ULL;
            mreg[u].base = 0;
            if (b < rs.num_blocks && w < words) mreg[u] = rs.meta[(int64_t)b * rs.words + word0 + w];
        }
        const bool next_static = !in_tail && at + 1 < at_end;
        int next_id_raw = -1;
        if (next_static) next_id_raw = f.sched[at + 1];
        double hub_sum = 0.0;
        if (!hub)
            for (int i = tid; i < rows; i += THREADS) s_row[i] = 0ULL;
        __syncthreads();
        const int groups = finite || hub ? bin.w : 0;
        const int nrounds = (groups + THREADS * FP - 1) / (THREADS * FP);
        if (hub) {
            for (int round = 0; round < nrounds; ++round) {
#pragma unroll
                for (int q = 0; q < FP; ++q)
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (R.r8[q][k] == 0) hub_sum += R.v[q][k >> 1][k & 1];
                if (round + 1 < nrounds) fetch(bin, round + 1, R);
            }
        } else {
            for (int round = 0; round < nrounds; ++round) {
#pragma unroll
                for (int q = 0; q < FP; ++q)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = (int)R.r8[q][k];
                        if (r < rows) {
                            const long long fixed = __double_as_longlong(__builtin_fma(R.v[q][k >> 1][k & 1], S, kMagic)) - __double_as_longlong(kMagic);
                            atomicAdd(&s_row[r], (unsigned long long)fixed);
                        }
                    }
                if (round + 1 < nrounds) fetch(bin, round + 1, R);
            }
        }
        // the next item: a static one's id was asked for above, the tail's ticket is taken here
        const int cur_slot = slot;
        int next = -1;
        if (next_static) {
            ++at;
            next = uni(next_id_raw);
        } else {
            in_tail = true;
            next = take_tail();
        }
        int4 next_bin = make_int4(0, 0, 0, 0), next_epi = make_int4(0, 0, -1, 0);
        if (next >= 0) {
            next_bin = f.item_a[next];        // (made uniform behind the epilogue: the wait for them belongs there)
            next_epi = f.item_b[next];
        }
#pragma unroll
        for (int u = 0; u < MPT; ++u) {
            const int j = tid + u * THREADS;
            if (j < NB * WORDS) {
                s_mask[j] = mreg[u].mask;
                s_base[j] = mreg[u].base;
            }
        }
        if (hub) {
            // one row: fixed-order reduction of the threads' sums; of a split row only the last arriver continues
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) hub_sum += __shfl_xor(hub_sum, d, 64);
            if (lane == 0) s_red[wave] = hub_sum;
            __syncthreads();
            if (tid == 0) {
                double total = 0.0;
                for (int w = 0; w < WAVES; ++w) total += s_red[w];
                int last = 1;
                if (pieces > 1) {
                    unsigned long long* part = reinterpret_cast<unsigned long long*>(f.hub_part) + item;
                    (void)atomicExch(part, (unsigned long long)__double_as_longlong(total));
                    __threadfence();
                    last = atomicAdd(f.hub_ticket + epi.z, 1u) == (unsigned)(pieces - 1) ? 1 : 0;
                }
                s_hub = total;
                s_last = last;
            }
            __syncthreads();
            if (pieces > 1 && s_last) {
                __threadfence();
                unsigned long long* part = reinterpret_cast<unsigned long long*>(f.hub_part) + epi.w;
                double* s_piece = reinterpret_cast<double*>(s_row);
                for (int k = tid; k < pieces; k += THREADS) s_piece[k] = __longlong_as_double((long long)atomicAdd(part + k, 0ULL));
                __syncthreads();
                if (tid == 0) {
                    double total = 0.0;
                    for (int k = 0; k < pieces; ++k) total += s_piece[k];
                    s_hub = total;
                    (void)atomicExch(f.hub_ticket + epi.z, 0u);       // re-arm for the next launch
                }
                __syncthreads();
            }
            if (s_last && tid == 0) {
                const int row = epi.x;
                const double sum = block_row_sum(row) + s_hub;
                const Ops o = load_ops(row);
                apply(row, sum, o, true);
            }
        } else {
            __syncthreads();
            // the item's rows in ALIGNED groups of 64 (lane = row % 64): the map word of a (block, group) is wavefront-uniform -- a
            // broadcast read of its LDS copy, the lane's bit a shift, its rank among the group's segments v_mbcnt.  Loads are
            // branch-free: lanes outside the item repeat a row of it, a row without a segment in a block reads the zero slot
            const int row_lo = epi.x, row_hi = epi.x + epi.y - 1;
            const int g_hi = row_hi >> 6;                       // (epi.y == 0: row_hi < row_lo, no group)
            constexpr int G = PGH_PB64_G;                       // groups per wavefront in flight
            for (int g0 = (row_lo >> 6) + wave; g0 <= g_hi && epi.y > 0; g0 += WAVES * G) {
                double v[G][NB];
                Ops o[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int g = min(g0 + u * WAVES, g_hi);    // wavefront-uniform; a group past the end repeats the last one (dropped below)
                    const int row = min(max((g << 6) + lane, row_lo), row_hi);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const unsigned long long mask = s_mask[b * WORDS + (g - word0)];
                        const unsigned int first = (unsigned int)s_base[b * WORDS + (g - word0)];
                        const unsigned int lo = (unsigned int)mask, hi = (unsigned int)(mask >> 32);
                        const unsigned int rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                        const bool has = (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) != 0u;
                        const unsigned int where = has ? first + rank : rs.zero_at;
                        v[u][b] = rs.psum[where];
                    }
                    o[u] = load_ops(row);
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int g = g0 + u * WAVES;
                    const int at_row = (g << 6) + lane;
                    const bool live = g <= g_hi && at_row >= row_lo && at_row <= row_hi;
                    const int row = live ? at_row : row_lo;
                    const int i = row - row_lo;
                    const double c = finite ? (double)(long long)s_row[i < rows ? i : 0] * inv_S : __longlong_as_double(0x7ff8000000000000LL);
                    double sum = 0.0;
#pragma unroll
                    for (int b = 0; b < NB; ++b) sum += v[u][b];
                    sum += i < rows ? c : 0.0;
                    apply(row, sum, o[u], live);
                }
            }
        }
        __syncthreads();                                   // s_row / s_hub are reused by the next item
        if (cur_slot != (int)blockIdx.x) flush(cur_slot);
        else if (in_tail) {
            flush(blockIdx.x);
            flushed_head = true;
        }
        if (next >= 0) {
            next_bin = uni4(next_bin), next_epi = uni4(next_epi);
            fetch(next_bin, 0, R);
        }
        item = next;
        bin = next_bin;
        epi = next_epi;
    }
    if (!flushed_head) flush(blockIdx.x);
    // the last workgroup to leave re-arms the words for the next launch
    if (tid == 0 && atomicAdd(f.amax + 1, 1ULL) == (unsigned long long)(gridDim.x - 1)) {
        f.amax[0] = 0ULL;
        f.amax[1] = 0ULL;
        if (tail_count > 0) *f.work_counter = 0u;
    }
    PGH_STAMP_END(g_times_finish64)
}

inline int grid_for(int64_t n, int per_cu) {
    int64_t blocks = (n + WG - 1) / WG;
    const int64_t cap = (int64_t)rt().num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

namespace pgh {

// first slot from which the rows of EVERY block are isolated (ranks >= live_nodes; deal_rank_of)
static int iso_from_of(const BsfFormat& f) {
    if (f.live_nodes < 0) return f.blk_size;
    if (f.pb.enabled) {
        // the finishing pass of the cold image passes over the work list's isolated items: rows [iso_begin[b], blk) of block b (pb_build).
        // The flag watches from the FIRST of those thresholds on (a superset of every block's stretch)
        int from = f.blk_size;
        for (int b = 0; b < f.num_blocks && b < 8; ++b) from = std::min(from, b < f.iso_row_blocks ? f.iso_begin[b] : f.blk_size);
        return f.has_iso ? from : f.blk_size;
    }
    int64_t from = 0;                                    // one line for all blocks: the last of their first isolated slots
    for (int b = 0; b < f.num_blocks; ++b) from = std::max(from, deal_first_slot(f.live_nodes, b, f.num_blocks, f.blk_size, f.deal_head));
    return (int)(from < f.blk_size ? from : f.blk_size);
}

bool bsf64_usable(const pgh_graph_s* g) {
    const char* off = getenv("PGH_CHEB_CSR");          // 1: the round-2 route over the row-major CSR (checker / measurements)
    if (off != nullptr && atoi(off) != 0) return false;
    return g->n_rows == g->n_cols && g->n_cols > 0 && g->nnz > 0 && g->part_perm == nullptr;
}

int bsf64_ensure(pgh_graph_s* g) {
    BsfFormat& f = g->bsf64;
    if (f.enabled) return 0;
    Runtime& r = rt();
    const bool valfree = g->keep_mult != nullptr;
    // eight XCD-affine blocks once a block's slice is worth an L2 of its own; small graphs: one block, all of it in LDS
    // Column blocks: every block's hot set is another kHot64 sources in some CU's LDS, and a cold gather is what the kernel
    // pays for (a divergent 8-byte load keeps a CU's address unit busy for ~4 cycles per lane) -- so many blocks, as long as
    // a block still feeds 256 / B workgroups of 16 wavefronts and the (row, block) segments stay few next to the entries.
    // Eight XCD-affine blocks once the graph outgrows one hot cache.  More blocks (16 / 32 / 64 are supported: an XCD's
    // workgroups share out its blocks) put more sources into some CU's LDS, but every block is another lookup per row in
    // the combine: measured at scale 23, 8 / 16 / 32 / 64 blocks: combine 116 / 192 / 344 / 682 us for 40 / 70 / 100 us less
    // in the partial sums.
    // (PGH_HOT64=<sources>: a smaller hot cache -- tests: graphs of a few thousand nodes then have cold entries, hub bins, heavy rows)
    int hot64 = kHot64;
    if (const char* h = getenv("PGH_HOT64")) hot64 = std::max(32, std::min(kHot64, atoi(h) / 32 * 32));
    int B = g->n_rows > (int64_t)hot64 ? 8 : 1;
    const char* forced = getenv("PGH_BLOCKS64");
    if (forced != nullptr) {
        const int fb = atoi(forced);
        if (fb == 1 || fb == 8 || fb == 16 || fb == 32 || fb == 64) B = fb;
    }
    f.want_meta = true;
    // round 6: the cold tail in a propagation-blocking image of its own (8 blocks at most: the finishing pass reads a row's segments of
    // eight blocks at once); PGH_PB64=0: round 5's route, cold gathers through the L2
    {
        const char* pb64 = getenv("PGH_PB64");
        f.pb64 = B <= 8 && (pb64 == nullptr || atoi(pb64) != 0);
        f.pb_hot = hot64;
        f.pb_chunk = kChunk64;
    }
    PGH_TRY(bsf_build(g, valfree ? nullptr : g->val, g->keep_mult, g->keep_src, g->keep_dst, true, B, &f));
    f.pb_hot = hot64;                                      // (kept with the image: View64::hot)
    if (f.pb.enabled) {
        for (int b = 0; b < 8; ++b) f.xg_base[b] = f.xg_base_cold[b] = (int64_t)b * f.blk_size;
        f.device_bytes += f.pb.device_bytes;
        const char* s16 = getenv("PGH_STREAM16");
        if (!f.pb.k1_cold && kHot64 < 32768 && (s16 == nullptr || atoi(s16) != 0)) {       // hot-only stream: 2 bytes per entry
            View64 v{};
            v.num_blocks = f.num_blocks;
            v.blk = f.blk_size;
            v.hot = hot64;
            for (int i = 0; i <= kMaxBlocks; ++i) v.tile_begin[i] = f.tile_begin[i];
            PGH_HIP(pooled_malloc(&f.colf16, sizeof(uint16_t) * (size_t)f.num_tiles * kT + 64));
            k_bsf64_narrow<<<grid_for(f.num_entries, 16), WG, 0, r.stream>>>(f.colf, f.num_entries, v, (uint32_t)std::min(hot64, f.blk_size), f.colf16);
            PGH_HIP(hipGetLastError());
            PGH_HIP(hipStreamSynchronize(r.stream));
            (void)pooled_free(f.colf);
            f.colf = nullptr;
            f.device_bytes -= f.num_entries * 2;
        }
    }
    PGH_HIP(pooled_malloc(&f.psum64, sizeof(double) * (size_t)(f.num_segs + kT + 64)));
    PGH_HIP(hipMemsetAsync(f.psum64, 0, sizeof(double) * (size_t)(f.num_segs + kT + 64), r.stream));
    PGH_HIP(pooled_malloc(&f.fix_seg, sizeof(int32_t) * (size_t)(f.num_tiles + 1)));
    k_bsf64_fixlist<<<grid_for(f.num_tiles, 16), WG, 0, r.stream>>>(f.tile, f.seg_row, f.num_tiles, f.fix_seg);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    f.device_bytes += (int64_t)(f.num_segs + kT + 64) * 8 + (int64_t)f.num_tiles * 4;
    if (f.live_nodes >= 0) PGH_HIP(pooled_malloc(&f.iso_flag, sizeof(int)));
    return 0;
}

int64_t bsf64_length(const pgh_graph_s* g) { return g->bsf64.n_out; }

int bsf64_bring(pgh_graph_s* g, const float* p, double c1, double* term, double* res, double* xg, bool keep_flag) {
    const BsfFormat& f = g->bsf64;
    // (keep_flag: a second operand of the same run -- the warm start -- adds its non-zeros on isolated rows to the first one's)
    if (f.iso_flag != nullptr && !keep_flag) PGH_HIP(hipMemsetAsync(f.iso_flag, 0, sizeof(int), rt().stream));
    k_bsf64_bring<<<grid_for(f.n_out, 16), WG, 0, rt().stream>>>(p, f.perm, f.src_scale, f.n_out, g->n_cols, c1, term, res, xg, f.iso_flag,
                                                                  f.blk_size, iso_from_of(f));
    PGH_HIP(hipGetLastError());
    return 0;
}

int bsf64_take(pgh_graph_s* g, const double* res, double factor, float* out) {
    const BsfFormat& f = g->bsf64;
    k_bsf64_take<<<grid_for(f.n_out, 16), WG, 0, rt().stream>>>(res, f.perm, f.n_out, g->n_cols, factor, out);
    PGH_HIP(hipGetLastError());
    return 0;
}

int bsf64_take_col(pgh_graph_s* g, const double* vec, float* mat, int ld, int col) {
    const BsfFormat& f = g->bsf64;
    k_bsf64_take_col<<<grid_for(f.n_out, 16), WG, 0, rt().stream>>>(vec, f.perm, f.n_out, g->n_cols, mat, ld, col);
    PGH_HIP(hipGetLastError());
    return 0;
}

// one term of the recurrence: term_out = a * (M^T term) + b * term, result += c * term_out; xg holds term * src_scale on
// entry and term_out * src_scale on return.  Block partials of sum(term_out) / delta land in partial_sum / partial_delta.
// Operands of the f64 recursive filters in the image's id space, from caller-space f32 vectors (padding slots: zeros).
//   mode 1, AbsorbingWalks (adhoc.py:166-169): y = (s * deg + p * lam) / (lam + deg) = s * row_w + term with row_w = deg / (lam + deg),
//           term = p_n * lam / (lam + deg)
//   mode 2, SymmetricAbsorbingRandomWalks (adhoc.py:348-364): d = deg, a = (1 + sqrt(1 + 4 d)) / 2; the iterate is pre-scaled by 1 / a on the
//           way into the product (src_w), row_w = d / (a + d), term = p_n * a / (a + d)
__global__ void k_bsf64_walk_operands(int mode, const float* __restrict__ p, const float* __restrict__ deg, const float* __restrict__ lam,
                                      const int32_t* __restrict__ perm, int64_t n_int, int64_t n_valid, double inv_norm,
                                      double* __restrict__ row_w, double* __restrict__ src_w, double* __restrict__ term) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_int; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = perm ? perm[i] : (i < n_valid ? i : -1);
        if (o < 0) {
            row_w[i] = 0.0;
            term[i] = 0.0;
            if (src_w != nullptr) src_w[i] = 1.0;
            continue;
        }
        const double pv = (double)p[o] * inv_norm, d = (double)deg[o];
        if (mode == 1) {
            const double l = (double)lam[o];
            row_w[i] = d / (l + d);
            term[i] = pv * l / (l + d);
        } else {
            const double a = (sqrt(d * 4.0 + 1.0) + 1.0) / 2.0;
            src_w[i] = 1.0 / a;
            row_w[i] = d / (a + d);
            term[i] = pv * (a / (a + d));
        }
    }
}
__global__ void k_bsf64_scale_by(double* __restrict__ x, const double* __restrict__ w, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= w[i];
}
int bsf64_walk_operands(pgh_graph_s* g, int mode, const float* p, const float* deg, const float* lam, double inv_norm, double* row_w,
                        double* src_w, double* term) {
    const BsfFormat& f = g->bsf64;
    k_bsf64_walk_operands<<<grid_for(f.n_out, 16), WG, 0, rt().stream>>>(mode, p, deg, lam, f.perm, f.n_out, g->n_cols, inv_norm, row_w, src_w, term);
    PGH_HIP(hipGetLastError());
    return 0;
}
int bsf64_scale_by(pgh_graph_s* g, double* x, const double* w) {
    k_bsf64_scale_by<<<grid_for(g->bsf64.n_out, 16), WG, 0, rt().stream>>>(x, w, g->bsf64.n_out);
    PGH_HIP(hipGetLastError());
    return 0;
}

int bsf64_step(pgh_graph_s* g, double a, double b, double c, const double* term, double* term_out, double* result, double* xg,
               int err_linf, const LoopState* state, double* partial_sum, double* partial_delta, int* num_partials, bool every_row,
               const double* row_w, const double* src_w) {
    Runtime& r = rt();
    BsfFormat& f = g->bsf64;
    View64 v;
    v.colf = f.colf;
    v.colf16 = f.colf16;
    v.val = f.val;
    v.tile = f.tile;
    v.tail = f.tail_carry;
    v.head = f.head_partial;
    v.psum = f.psum64;
    v.num_blocks = f.num_blocks;
    v.blk = f.blk_size;
    v.hot = f.pb_hot > 0 ? f.pb_hot : kHot64;
    for (int i = 0; i <= kMaxBlocks; ++i) v.tile_begin[i] = f.tile_begin[i];
    const int unit = f.num_blocks > 8 ? f.num_blocks : 8;                // whole XCD rounds, and whole rounds of an XCD's blocks
    const int main_grid = r.num_cus >= unit ? r.num_cus / unit * unit : unit;
    {
        ProfScope prof(PGH_K_SPMV);
        PendingClose pc = pending_close_slot();
        if (state == nullptr || pc.state != state) pc.active = 0;
        else pending_close_slot().active = 0;              // consumed
        const bool cold = !f.pb.enabled || f.pb.k1_cold;    // cold gathers in the stream: no cold image, or rows too heavy for its bins
        const bool narrow = f.colf16 != nullptr;
        if (f.val && cold) k_bsf64_partial<true, true><<<main_grid, kThreads, 0, r.stream>>>(v, xg, state, pc);
        else if (f.val && narrow) k_bsf64_partial<true, false, true><<<main_grid, kThreads, 0, r.stream>>>(v, xg, state, pc);
        else if (f.val) k_bsf64_partial<true, false><<<main_grid, kThreads, 0, r.stream>>>(v, xg, state, pc);
        else if (cold) k_bsf64_partial<false, true><<<main_grid, kThreads, 0, r.stream>>>(v, xg, state, pc);
        else if (narrow) k_bsf64_partial<false, false, true><<<main_grid, kThreads, 0, r.stream>>>(v, xg, state, pc);
        else k_bsf64_partial<false, false><<<main_grid, kThreads, 0, r.stream>>>(v, xg, state, pc);
    }
    PGH_STAMP_DUMP(g_times_partial64, main_grid, "k_bsf64_partial")
    Fix64 fix;
    fix.fix_seg = f.fix_seg;
    fix.tile = f.tile;
    fix.tail = f.tail_carry;
    fix.head = f.head_partial;
    fix.psum = f.psum64;
    fix.num_tiles = f.num_tiles;
    static const bool fold_env = getenv("PGH_PB64_FOLD") == nullptr || atoi(getenv("PGH_PB64_FOLD")) != 0;
    const bool fix_rides = fold_env && f.pb.enabled && f.pb.num_tasks > 0;       // inside phase A of the cold image
    if (!fix_rides) {
        ProfScope prof(PGH_K_FIXUP);
        int fix_grid = (f.num_tiles + WG - 1) / WG;
        fix_grid = fix_grid < 1 ? 1 : (fix_grid > 1024 ? 1024 : fix_grid);
        k_bsf64_fixup<<<fix_grid, WG, 0, r.stream>>>(fix, state);
    }
    Epi64 ep;
    ep.a = a;
    ep.b = b;
    ep.c = c;
    ep.term = term;
    ep.term_out = term_out;
    ep.r = result;
    ep.xg_out = xg;
    ep.src_scale = f.src_scale;
    ep.dst_scale = f.dst_scale;
    ep.err_linf = err_linf;
    ep.row_w = row_w;
    ep.src_w = src_w;
    if (f.pb.enabled) {
        // phase A of the cold image, then phase B + the epilogue over its work list (every output row belongs to one item)
        const PbFormat& p = f.pb;
        Pb64View pv;
        pv.sloc = p.sloc;
        pv.val = p.val;
        pv.dstg = p.dstg;
        pv.task = p.task;
        pv.task_range = p.task_range;
        pv.first_task = p.first_task;
        pv.tmp = p.tmp64;
        pv.amax = p.amax64;
        pv.item_a = p.item_a;
        pv.item_b = p.item_b;
        pv.sched = p.sched;
        pv.sched_begin = p.sched_begin;
        pv.first_a = p.first_a;
        pv.first_b = p.first_b;
        pv.first_item = p.first_item;
        pv.work_counter = p.work_counter;
        pv.tail_begin = p.tail_begin;
        pv.tail_count = p.tail_count;
        pv.hub_ticket = p.hub_ticket;
        pv.hub_part = p.hub_part;
        pv.drow = p.drow;
        pv.iso_flag = every_row ? nullptr : f.iso_flag;
        for (int i = 0; i < 9; ++i) pv.cold_prefix[i] = p.cold_prefix[i];
        for (int i = 0; i < 8; ++i) pv.xg_base[i] = f.xg_base_cold[i];
        pv.num_blocks = f.num_blocks;
        pv.hot = p.hot;
        pv.chunk = p.chunk;
        pv.short_piece = p.short_piece;
        pv.num_cold = p.cold_prefix[f.num_blocks];
        {
            ProfScope prof(PGH_K_PB_GATHER);
            if (p.num_tasks > 0) {
                Fix64 ride = fix;
                if (!fix_rides) ride.num_tiles = 0;
                if (p.val) k_pb64_gather<true><<<p.num_tasks, kGather64Threads, 0, r.stream>>>(pv, xg, state, ride);
                else k_pb64_gather<false><<<p.num_tasks, kGather64Threads, 0, r.stream>>>(pv, xg, state, ride);
            }
        }
        PGH_STAMP_DUMP(g_times_gather64, p.num_tasks, "k_pb64_gather")
        Rows64 rs;
        rs.meta = f.meta;
        rs.words = f.meta_words;
        rs.psum = f.psum64;
        rs.num_blocks = f.num_blocks;
        rs.zero_at = (unsigned int)(f.num_segs + kT + 63);
        const int grid = p.sched_groups;
        {
            ProfScope prof(PGH_K_PB_ACCUM);
            if (p.bin_rows > 8192) k_pb64_finish<16384, 1024><<<grid, 1024, 0, r.stream>>>(pv, rs, ep, state, partial_sum, partial_delta);
            else if (p.bin_rows > 4096) k_pb64_finish<8192, 512><<<grid, 512, 0, r.stream>>>(pv, rs, ep, state, partial_sum, partial_delta);
            else k_pb64_finish<4096, 256><<<grid, 256, 0, r.stream>>>(pv, rs, ep, state, partial_sum, partial_delta);
        }
        PGH_STAMP_DUMP(g_times_finish64, grid, "k_pb64_finish")
        PGH_HIP(hipGetLastError());
        if (num_partials) *num_partials = grid + p.tail_count;
        return 0;
    }
    const int cgrid = grid_for((f.n_out + 3) / 4, 8);
    {
        ProfScope prof(PGH_K_COMBINE);
        IsoRows iso;
        iso.flag = every_row ? nullptr : f.iso_flag;       // every_row: term_out is read in full afterwards (pgh_poly_terms)
        iso.blk = f.blk_size;
        iso.iso_from = iso_from_of(f);
        // (the pad behind the segments is cleared at build time and never written: its last word is the zero slot)
        k_bsf64_combine<<<cgrid, WG, 0, r.stream>>>(f.meta, f.meta_words, f.psum64, f.num_blocks, f.n_out, ep, state, partial_sum, partial_delta, iso,
                                                    (unsigned int)(f.num_segs + kT + 63));
    }
    PGH_HIP(hipGetLastError());
    if (num_partials) *num_partials = cgrid;
    return 0;
}

}  // namespace pgh

PGH_WARM_KERNEL(k_bsf64_fixlist)
