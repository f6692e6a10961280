// Column-blocked segment-flag format (BSF) of CSR(M^T): builder and the two propagation kernels that stream it.
//
// Why (MI355X, measured -- profiles/r01): the row-major merge-path SpMV moves 4.7 GB per launch for 1.18 GB of
// algorithmic bytes at RMAT scale 23; 3.5 GB of that is 128-byte line over-fetch for 4-byte x[col] gathers that miss
// the 4 MB per-XCD L2.  The fix is a layout decision, made once at upload time:
//   1. relabel: sources sorted by descending reference count and dealt round-robin to B column blocks, so every
//      block's slice of the gather vector is one contiguous hot-first range;
//   2. block b's entries are streamed only by workgroups whose dispatch slot maps to one XCD group
//      (blockIdx % 8, a speed-only affinity), so that slice stays resident in that XCD's L2 while the matrix
//      stream passes through with non-temporal loads;
//   3. per block, entries are sorted by (row, col) and a row segment is marked by bit 31 of its first column
//      index: the SpMV pass reads no row pointers and never touches empty rows (k_bsf_pack then digests the stream for
//      the kernel: block-local byte offsets, one flag byte per lane, tiles transposed for coalesced 16-byte loads);
//   4. when M^T = diag(dst) * W * diag(src) with small integer W (the preprocessor's "col"/"symmetric"
//      normalisations of an unweighted or multi-edge graph, preprocessing.py:109-138) the values disappear:
//      multiplicities become repeated entries (4 B/edge), src moves into the gather vector, dst into the epilogue;
//   5. entries whose source is outside a block's LDS hot cache ("cold") move to the propagation-blocking image of
//      pgh_pb.hip when there are enough of them; the stream then holds hot entries only and is narrowed to 16-bit
//      words (2 B/edge, k_bsf_narrow).
// Block partial sums go to B dense f32 vectors; k_bsf_combine folds them with the filter's epilogue
// (apply_epilogue: adhoc.py:34-36,166-169; abstract_filters.py:215-230) and writes the next gather vector.
//
// Reference counterpart of the arithmetic: conv(signal, M) = signal @ M (pygrank/core/backend/numpy.py:64-65).
#include "pgh_kernels.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <vector>

using namespace pgh;

namespace {

constexpr int kBlock = 256;
constexpr int kIPT = PGH_BSF_IPT;          // entries per lane of a wavefront tile
constexpr int kTile = 64 * kIPT;            // one tile = one wavefront
constexpr int kBsfThreads = 1024;           // one 16-wavefront workgroup per CU shares the hot cache
constexpr int kBsfHot = PGH_BSF_HOT;        // f32 entries of the gather vector cached in LDS per workgroup
constexpr uint64_t kLow29 = (1ULL << 29) - 1;
constexpr uint64_t kRowSentinel = kLow29;        // sorts after every real row of a block

inline int blocks_for(int64_t n, int cap_mult = 16) {
    int64_t b = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rt().num_cus * cap_mult;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() {
        if (p) (void)pooled_free(p);
    }
    int alloc(size_t count, bool zero = false) {
        PGH_HIP(pooled_malloc(&p, sizeof(T) * (count > 0 ? count : 1)));
        if (zero) PGH_HIP(hipMemsetAsync(p, 0, sizeof(T) * (count > 0 ? count : 1), rt().stream));
        return 0;
    }
    T* release() {
        T* q = p;
        p = nullptr;
        return q;
    }
};

// ------------------------------------------------------------------------------------------------- builder kernels
// reference count of every source (weights count as multiplicities when the format is value-free)
__global__ void k_source_counts(const int32_t* __restrict__ col, const int32_t* __restrict__ mult, int64_t nnz,
                                unsigned int* __restrict__ cnt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride)
        atomicAdd(&cnt[col[k]], mult ? (unsigned int)mult[k] : 1u);
}

// sort key of the relabelling: reference count, ties broken in favour of ids whose ROW holds entries (square graphs: rows
// and sources share the id space).  Ids that nobody references and whose row is empty -- isolated nodes -- then form the
// tail of every block; `live` counts the others.
__global__ void k_relabel_keys(const unsigned int* __restrict__ cnt, const int32_t* __restrict__ rowptr, int64_t n, unsigned int* __restrict__ key,
                               unsigned int* __restrict__ live) {
    unsigned int mine = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int c = cnt[i] < 0x7fffffffu ? cnt[i] : 0x7fffffffu;
        const unsigned int has_row = rowptr[i + 1] > rowptr[i] ? 1u : 0u;
        key[i] = (c << 1) | has_row;
        mine += (c | has_row) != 0u ? 1u : 0u;
    }
    if (mine) atomicAdd(live, mine);
}

__global__ void k_iota(int32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int32_t)i;
}

// rank r (descending count) -> new id = (r % B) * blk + r / B
__global__ void k_make_perm(const int32_t* __restrict__ sorted_ids, int64_t n, int B, int blk, int64_t head, int32_t* __restrict__ perm,
                            int32_t* __restrict__ iperm) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int old = sorted_ids[r];
        const int nw = (int)deal_new_id(r, B, blk, head);
        iperm[old] = nw;
        perm[nw] = old;
    }
}

__global__ void k_fill_perm_pad(int32_t* __restrict__ perm, int64_t n_pad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) perm[i] = -1;
}

// expand CSR(M^T) rows into sort keys (block << 58 | new_row << 29 | new_col); `offs` = first output slot of entry k
__global__ void k_bsf_keys(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
                           const int32_t* __restrict__ mult, const int64_t* __restrict__ offs, int64_t n_out,
                           const int32_t* __restrict__ iperm_rows, const int32_t* __restrict__ iperm_cols, int blk,
                           uint64_t* __restrict__ keys, float* __restrict__ vals_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_out; r += nwaves) {
        const uint64_t nr = (uint64_t)(iperm_rows ? iperm_rows[r] : (int32_t)r);
        for (int64_t k = rowptr[r] + lane; k < rowptr[r + 1]; k += 64) {
            const int c = iperm_cols ? iperm_cols[col[k]] : col[k];
            const uint64_t key = ((uint64_t)(c / blk) << 58) | (nr << 29) | (uint64_t)c;
            const int64_t o = offs ? offs[k] : k;
            const int m = mult ? mult[k] : 1;
            for (int q = 0; q < m; ++q) keys[o + q] = key;
            if (vals_out)                                  // (value AND multiplicity: every copy carries the value -- the entry-index
                for (int q = 0; q < m; ++q) vals_out[o + q] = val[k];       // words of graph_dropout are built that way)
        }
    }
}

// ---- the same expansion, one thread per ENTRY (round 6).  k_bsf_keys gives a wavefront a row: a hub row of 10^5 entries is 1 600 serial
// trips of one wavefront while half the rows are empty -- 50 ms for 10 GB at scale 23, a third of the build.  Here the row
// of every entry is materialised first (its index scattered to the row's first entry, a running maximum spreads it: one scan), then
// the expansion is a plain stream over the entries.
__global__ void k_row_heads(const int32_t* __restrict__ rowptr, int64_t n_out, int32_t* __restrict__ row_of) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_out; r += (int64_t)gridDim.x * blockDim.x)
        if (rowptr[r + 1] > rowptr[r]) row_of[rowptr[r]] = (int32_t)r;        // (empty rows own no entry)
}
__global__ void k_bsf_keys_flat(const int32_t* __restrict__ row_of, const int32_t* __restrict__ col, const float* __restrict__ val,
                                const int32_t* __restrict__ mult, const int64_t* __restrict__ offs, int64_t nnz,
                                const int32_t* __restrict__ iperm_rows, const int32_t* __restrict__ iperm_cols, int blk,
                                uint64_t* __restrict__ keys, float* __restrict__ vals_out) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t r = row_of[k];
        const uint64_t nr = (uint64_t)(iperm_rows ? iperm_rows[r] : r);
        const int c = iperm_cols ? iperm_cols[col[k]] : col[k];
        const uint64_t key = ((uint64_t)(c / blk) << 58) | (nr << 29) | (uint64_t)c;
        const int64_t o = offs ? offs[k] : k;
        const int m = mult ? mult[k] : 1;
        for (int q = 0; q < m; ++q) keys[o + q] = key;
        if (vals_out) {
            const float v = val[k];
            for (int q = 0; q < m; ++q) vals_out[o + q] = v;
        }
    }
}

__global__ void k_bsf_sentinels(uint64_t* __restrict__ keys, int64_t first, int B, int n_src_pad) {
    const int b = threadIdx.x;
    // the sentinel's segment is never closed, so its product is irrelevant: it gathers from the block's first slot
    if (b < B) keys[first + b] = ((uint64_t)b << 58) | (kRowSentinel << 29) | (uint64_t)((int64_t)b * (n_src_pad / B));
}

// sorted keys -> colf (+flag), int flags for the segment scan
__global__ void k_bsf_flags(const uint64_t* __restrict__ keys, int64_t E, uint32_t* __restrict__ colf, int* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += stride) {
        const uint64_t key = keys[e];
        const int f = (e == 0) || ((key >> 29) != (keys[e - 1] >> 29));
        colf[e] = (uint32_t)(key & kLow29) | ((uint32_t)f << 31);
        flags[e] = f;
    }
}

// final pass of the build: digest the entry stream for k_bsf_partial.  Thread = (tile, lane): its 8 column words become
// block-local byte offsets, their bit-31 flags become one byte.
struct PackLayout {
    int tile_begin[9];
    int num_blocks;
};
__global__ void k_bsf_pack(uint32_t* __restrict__ colf, uint32_t* __restrict__ val, PackLayout pl, int num_tiles, int blk,
                           uint8_t* __restrict__ flags8, int32_t* __restrict__ live) {
    // one WAVEFRONT per tile (blockDim is a multiple of 64): every lane reads its 8 logical words before any lane
    // writes, so the in-place transposition below is safe
    const int64_t total = (int64_t)num_tiles * 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tile = (int)(i >> 6), lane = (int)(i & 63);
        int b = 0;
        while (b + 1 < pl.num_blocks && tile >= pl.tile_begin[b + 1]) ++b;
        const uint32_t base = (uint32_t)b * (uint32_t)blk;
        uint32_t* w = colf + i * 8;
        uint32_t x[8], v[8];
        unsigned int bits = 0;
        uint32_t top = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x[k] = w[k];
            if (val) v[k] = val[i * 8 + k];
            bits |= (x[k] >> 31) << k;
            const uint32_t loc = (x[k] & 0x7fffffffu) - base;
            top = loc > top ? loc : top;           // sentinels and pads point at slot 0
            x[k] = loc << 2;
        }
        flags8[i] = (uint8_t)bits;
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_down(top, off, 64);
            top = o > top ? o : top;
        }
        if (lane == 0) atomicMax(&live[b], (int32_t)top + 1);
        // physical order inside the tile: [q = k / 4][lane][k % 4], so that the q-th 16-byte load of the 64 lanes
        // covers one contiguous KB (PGH_TILE_TRANSPOSE = 0 keeps the logical order: lane-contiguous 32 bytes)
        const int64_t t0 = (int64_t)tile * 512;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t dst = PGH_TILE_TRANSPOSE ? t0 + (k >> 2) * 256 + lane * 4 + (k & 3) : t0 + lane * 8 + k;
            colf[dst] = x[k];
            if (val) val[dst] = v[k];
        }
    }
}

// Hot-only streams (every cold entry lives in the propagation-blocking image): a source is one of the <= 29 696 slots of
// the block's LDS hot cache, so its byte offset / 2 fits 16 bits.  [tile][lane][8] halfwords: one 16-byte load per lane.
// Sentinels and pads (slot 0 of the block: any value, they never close a segment) keep pointing at a valid slot.
__global__ void k_bsf_narrow(const uint32_t* __restrict__ colf, int num_tiles, uint32_t hot4, uint16_t* __restrict__ colf16) {
    const int64_t total = (int64_t)num_tiles * 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t0 = (i >> 6) * 512;
        const int lane = (int)(i & 63);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t src = PGH_TILE_TRANSPOSE ? t0 + (k >> 2) * 256 + lane * 4 + (k & 3) : t0 + lane * 8 + k;
            const uint32_t w = colf[src];
            colf16[t0 + lane * 8 + k] = (uint16_t)((w < hot4 ? w : hot4) >> 1);
        }
    }
}

// 1 + highest referenced slot of every block, over all entries of the sorted stream (sentinels point at slot 0)
// (round 6: the keys are sorted by block, so a thread keeps the running maximum of the block it is in and a wavefront whose lanes end in
// one block reports ONE value -- the per-entry test against live[b] plus its atomics took 5.9 ms for a 1-GB read at scale 23)
__global__ void k_bsf_live(const uint64_t* __restrict__ keys, int64_t E, int blk, int32_t* __restrict__ live) {
    int cur_b = -1, cur = 0;
    // (a wavefront owns a CONTIGUOUS stretch of the sorted keys: it meets a block boundary only when its stretch holds one -- with a
    // grid-stride loop every thread crosses every boundary and flushes there: 83 ms)
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t len = ((E + waves - 1) / waves + 63) & ~(int64_t)63;
    const int64_t e_end = min(E, (wave + 1) * len);
    for (int64_t e = wave * len + (threadIdx.x & 63); e < e_end; e += 64) {
        const uint64_t key = keys[e];
        const int b = (int)(key >> 58);
        const int loc = (int)((int64_t)(key & kLow29) - (int64_t)b * blk);
        if (b != cur_b) {
            if (cur_b >= 0) atomicMax(&live[cur_b], cur);
            cur_b = b;
            cur = 0;
        }
        cur = max(cur, loc + 1);
    }
    const int first_b = __shfl(cur_b, 0, 64);
    if (__all(cur_b == first_b)) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cur = max(cur, __shfl_xor(cur, d, 64));
        if ((threadIdx.x & 63) == 0 && cur_b >= 0) atomicMax(&live[cur_b], cur);
    } else if (cur_b >= 0) {
        atomicMax(&live[cur_b], cur);
    }
}

__global__ void k_bsf_is_hot(const uint64_t* __restrict__ keys, int64_t E, int blk, int hot, unsigned char* __restrict__ flag) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[e];
        const int b = (int)(key >> 58);
        flag[e] = ((int64_t)(key & kLow29) - (int64_t)b * blk) < hot ? 1 : 0;
    }
}

// diagnostics (PGH_DEBUG=1): entries whose column falls inside the per-workgroup hot cache
__global__ void k_bsf_count_hot(const uint32_t* __restrict__ colf, int64_t E, int blk, int hot, unsigned long long* __restrict__ out) {
    unsigned long long local = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += stride) {
        const uint32_t c = colf[e] & 0x7fffffffu;
        if ((int)(c % (uint32_t)blk) < hot) ++local;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// segid = inclusive scan of flags; segment s = segid - 1 starts at the flagged entry.  With `meta` (SpMV layout) the
// segment also enters the row -> segment map of its block: bit (row % 64) of word row / 64, base = the word's lowest
// segment index (a block's segments are sorted by row, so the segments of one word are consecutive).
__global__ void k_bsf_seg_rows(const uint64_t* __restrict__ keys, const int* __restrict__ segid, int64_t E,
                               int32_t* __restrict__ seg_row, SegMeta* __restrict__ meta, int64_t meta_words) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += stride) {
        const bool f = (e == 0) || (segid[e] != segid[e - 1]);
        if (f) {
            const uint64_t row = (keys[e] >> 29) & kLow29;
            const int seg = segid[e] - 1;
            seg_row[seg] = row == kRowSentinel ? -1 : (int32_t)row;
            if (meta != nullptr && row != kRowSentinel) {
                SegMeta* m = meta + (int64_t)(keys[e] >> 58) * meta_words + (int64_t)(row >> 6);
                atomicOr(&m->mask, 1ULL << (row & 63));
                atomicMin(&m->base, seg);
            }
        }
    }
}
__global__ void k_bsf_meta_init(SegMeta* __restrict__ meta, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        meta[i].mask = 0ULL;
        meta[i].base = 0x7fffffff;
        meta[i].pad = 0;
    }
}

// first entry of every block (lower bound of b << 58), E for b == B
__global__ void k_bsf_block_starts(const uint64_t* __restrict__ keys, int64_t E, int B, int64_t* __restrict__ starts) {
    const int b = threadIdx.x;
    if (b > B) return;
    const uint64_t target = b < 64 ? (uint64_t)b << 58 : ~0ULL;        // b == B == 64: past every key
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    starts[b] = lo;
}

struct PadLayout {
    int64_t src_start[kMaxBlocks + 1];     // first sorted entry of every block (src_start[B] = E)
    int64_t dst_start[kMaxBlocks + 1];     // first padded slot of every block (multiples of the tile size)
    int     B;
};

// sorted keys -> tile-padded keys; pad slots repeat the block's sentinel key (same (block, row): no flag)
__global__ void k_bsf_pad(PadLayout pl, const uint64_t* __restrict__ keys, const float* __restrict__ vals, int64_t EP, int blk,
                          uint64_t* __restrict__ keys_out, float* __restrict__ vals_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < EP; e += stride) {
        int b = 0;
        while (b + 1 < pl.B && e >= pl.dst_start[b + 1]) ++b;
        const int64_t src = pl.src_start[b] + (e - pl.dst_start[b]);
        const bool real = src < pl.src_start[b + 1];
        keys_out[e] = real ? keys[src] : (((uint64_t)b << 58) | (kRowSentinel << 29) | (uint64_t)((int64_t)b * blk));
        if (vals_out) vals_out[e] = real ? vals[src] : 0.f;
    }
}

struct TileBuild {
    int64_t block_start[kMaxBlocks + 1];
    int     tile_begin[kMaxBlocks + 1];
    int     B;
};

__global__ void k_bsf_tiles(TileBuild tb, const int* __restrict__ segid, int4* __restrict__ tile) {
    const int total = tb.tile_begin[tb.B];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int b = 0;
        while (b + 1 < tb.B && t >= tb.tile_begin[b + 1]) ++b;
        const int64_t start = tb.block_start[b] + (int64_t)(t - tb.tile_begin[b]) * kTile;
        const int64_t end = min(start + (int64_t)kTile, tb.block_start[b + 1]);
        const int seg_before = start > 0 ? segid[start - 1] : 0;       // segments started before this tile
        const int flags_here = segid[end - 1] - seg_before;
        int first = -1;
        if (flags_here > 0) {
            // chain: tiles first..t-1 carry pieces of the segment that is open when tile t starts
            int s = t;
            while (s > tb.tile_begin[b]) {
                const int64_t ps = tb.block_start[b] + (int64_t)(s - 1 - tb.tile_begin[b]) * kTile;
                const int64_t pe = ps + kTile;                           // previous tiles are always full
                const int pflags = segid[pe - 1] - (ps > 0 ? segid[ps - 1] : 0);
                --s;
                if (pflags > 0) break;
            }
            first = s;
        }
        tile[t] = make_int4((int)start, (int)(end - start), seg_before - 1, first);
    }
}

__global__ void k_permute_in(const float* __restrict__ src, const int32_t* __restrict__ perm, const float* __restrict__ scale,
                             int64_t n_pad, int64_t n_valid, float hole, float* __restrict__ dst, int xg_blk = 0, int xg_live = 0) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = xg_slot((int)i, xg_blk, xg_live);
        if (slot < 0) continue;
        const int o = perm ? perm[i] : (i < n_valid ? (int)i : -1);
        float v = o >= 0 ? src[o] : hole;
        if (scale) v *= scale[i];
        dst[slot] = v;
    }
}

// personalization and start vector of a recursive loop in one pass over the permutation (square, relabelled graphs):
// v_int = v[perm], y0 = ranks[perm], xg = ranks[perm] * src_scale
// Seed-set operands (a few non-zeros among millions of zeros -- what a personalization usually is): instead of gathering every
// slot through the permutation (one L2 request per slot: 70-80 us at scale 23), one streaming pass in the caller's order
// clears the internal vectors and lists the non-zeros (k_pair_scan), and the listed entries are scattered (k_pair_scatter).
// More non-zeros than the list holds: the gather pass below runs instead (it returns at once otherwise).
constexpr int kSeedListCap = 1 << 16;

__global__ void k_pair_scan(const float* __restrict__ v, const float* __restrict__ ranks, int64_t n_orig, int64_t n_pad, int64_t xg_len,
                            float* __restrict__ v_int, float* __restrict__ y0, float* __restrict__ xg, int32_t* __restrict__ list,
                            int* __restrict__ count, double* __restrict__ norm_partials, LoopAux* __restrict__ stamp) {
    __shared__ double s_red[4];
    // the run's first kernel: its start on the device's clock (LoopAux::t_begin; the closing launch leaves the field alone)
    if (stamp != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamp->t_begin = __builtin_amdgcn_s_memrealtime();
    double abs_sum = 0.0;                        // this thread's share of sum |v| (GraphFilter.rank's norm, abstract_filters.py:52)
    const int64_t span = n_pad > xg_len ? n_pad : xg_len;
    // four ids per thread and round, their loads issued together (one id per round left every load alone in flight: the scan took
    // 21-24 us at scale 23 once it also summed |v|, 14 us with the loads side by side)
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < span; i0 += stride * U) {
        float vi[U], ri[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            vi[u] = i < n_orig ? v[i] : 0.f;
            ri[u] = (ranks != nullptr && i < n_orig) ? ranks[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n_pad) {
                v_int[i] = 0.f;
                if (y0 != nullptr) y0[i] = 0.f;
            }
            if (xg != nullptr && i < xg_len) xg[i] = 0.f;
            abs_sum += fabs((double)vi[u]);
            if (vi[u] != 0.f || ri[u] != 0.f) {
                const int pos = atomicAdd(count, 1);
                if (pos < kSeedListCap) list[pos] = (int32_t)i;
            }
        }
    }
    if (norm_partials != nullptr) {              // fixed order: a thread's ids, the workgroup's tree, the workgroups in k_scan_close
        const double total = block_reduce_256<0>(abs_sum, s_red);
        if (threadIdx.x == 0) norm_partials[blockIdx.x] = total;
    }
}

// one workgroup behind the scan: starts the loop state of the run that follows (k_state_init's work) and folds the scan's
// partials of sum |v| into LoopAux::in_norm
// ... and clears two words for the launches that follow: the isolated-row flag (the scatter / gather pass raises it) and the list
// counter of the NEXT run (the counters alternate, so no memset stands in front of a run's scan)
__global__ __launch_bounds__(kBlock) void k_scan_close(const double* __restrict__ norm_partials, int count, LoopState* init_state,
                                                       LoopAux* init_aux, int* iso_flag, int* next_seed_count) {
    __shared__ double s4[4];
    if (threadIdx.x == 0) {
        if (iso_flag != nullptr) *iso_flag = 0;
        if (next_seed_count != nullptr) *next_seed_count = 0;
    }
    if (init_state != nullptr && threadIdx.x == 0) {
        if (init_aux != nullptr) {
            init_aux->pred_inv[0] = init_aux->pred_inv[1] = 1.0;
            init_aux->pred_raw[0] = init_aux->pred_raw[1] = 0.0;
            init_aux->sum_p = 0.0;
            init_aux->worst_miss = 0.0;
            init_aux->t0_fix = init_aux->sp_fix = 0;
        }
        init_state->scale = 1.0;
        init_state->err = 0.0;
        init_state->sum = 0.0;
        init_state->done = 0;
        init_state->steps = 0;
        init_state->converged = 0;
        init_state->pad = 0;
    }
    if (norm_partials != nullptr && init_aux != nullptr) {
        const double norm = fold_partials_wide(norm_partials, count, 0, s4);
        if (threadIdx.x == 0) init_aux->in_norm = norm;
    }
}

__device__ __forceinline__ void pair_scatter(const int32_t* __restrict__ list, int total, const float* __restrict__ v,
                                             const float* __restrict__ ranks, const int32_t* __restrict__ iperm, const float* __restrict__ scale,
                                             float* __restrict__ v_int, float* __restrict__ y0, float* __restrict__ xg, int xg_blk, int xg_live,
                                             float in_norm, int start_from_v, const IsoTail& iso, const float* __restrict__ pred_deg,
                                             double& pred_t, double& pred_p) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int old = list[k];
        const int i = iperm[old];
        float a = v[old];
        float b = start_from_v ? 0.f : ranks[old];
        if (in_norm != 1.f) a = a / in_norm;                // the same f32 division as the backend's `p / norm`
        if (start_from_v) b = a;
        v_int[i] = a;
        if (y0 != nullptr) y0[i] = b;
        if (pred_deg != nullptr) {
            pred_t += (double)pred_deg[i] * (double)b;
            pred_p += (double)a;
        }
        if (iso.flag != nullptr && (a != 0.f || b != 0.f) && iso.holds(i)) atomicOr(iso.flag, 1);
        if (xg) {
            const int slot = xg_slot(i, xg_blk, xg_live);
            if (slot >= 0) xg[slot] = scale ? b * scale[i] : b;
        }
    }
}

// (seed_count != null: k_pair_scan has run.  Few enough non-zeros: they are scattered from its list -- the first blocks of this
// launch do it, the others leave at once; else the gather pass over every slot, as without the scan)
__global__ void k_permute_in_pair(const float* __restrict__ v, const float* __restrict__ ranks, const int32_t* __restrict__ perm,
                                  const float* __restrict__ scale, int64_t n_pad, float* __restrict__ v_int, float* __restrict__ y0,
                                  float* __restrict__ xg, int xg_blk, int xg_live, float in_norm, int start_from_v,
                                  IsoTail iso = IsoTail{}, const int* __restrict__ seed_count = nullptr,
                                  const int32_t* __restrict__ seed_list = nullptr, const int32_t* __restrict__ iperm = nullptr,
                                  const LoopAux* __restrict__ norm_from = nullptr, const float* __restrict__ pred_deg = nullptr,
                                  LoopAux* __restrict__ pred_aux = nullptr) {
    // pred_deg (row sums of M, internal ids): the sums of the first step's prediction go to pred_aux->t0_fix / sp_fix
    __shared__ double s_pred[4];
    double pred_t = 0.0, pred_p = 0.0;
    auto pred_publish = [&]() __attribute__((always_inline)) {          // (called by whole workgroups)
        const double bt = block_reduce_256<0>(pred_t, s_pred);
        __syncthreads();
        const double bp = block_reduce_256<0>(pred_p, s_pred);
        if (threadIdx.x == 0 && (bt != 0.0 || bp != 0.0)) {
            // (sums beyond the fixed point's range leave a prediction that misses: the close's bound notices and pauses)
            const double lim = 4.0e6;
            atomicAdd(reinterpret_cast<unsigned long long*>(&pred_aux->t0_fix), (unsigned long long)(long long)llrint(fmin(fmax(bt, -lim), lim) * kPredFix));
            atomicAdd(reinterpret_cast<unsigned long long*>(&pred_aux->sp_fix), (unsigned long long)(long long)llrint(fmin(fmax(bp, -lim), lim) * kPredFix));
        }
    };
    // the norm the scan has just summed (a zero personalization divides by 1: every vector of the run is zero, the caller is told)
    if (norm_from != nullptr) in_norm = norm_from->in_norm != 0.0 ? (float)norm_from->in_norm : 1.f;
    if (seed_count != nullptr) {
        const int total = *seed_count;
        if (total <= kSeedListCap) {
            if (blockIdx.x * (int64_t)blockDim.x >= total) return;      // (workgroup-uniform)
            pair_scatter(seed_list, total, v, ranks, iperm, scale, v_int, y0, xg, xg_blk, xg_live, in_norm, start_from_v, iso, pred_deg, pred_t, pred_p);
            if (pred_deg != nullptr) pred_publish();
            return;
        }
    }
    // four independent (index -> gather) chains per thread and round: one chain per round leaves the loop latency-bound
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < n_pad; i0 += stride * U) {
        int o[U];
        float a[U], b[U], sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            o[u] = i < n_pad ? perm[i] : -1;
            sc[u] = scale && i < n_pad ? scale[i] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = o[u] >= 0 ? v[o[u]] : 0.f;
            b[u] = !start_from_v && o[u] >= 0 ? ranks[o[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n_pad) continue;
            if (in_norm != 1.f) a[u] = a[u] / in_norm;    // the same f32 division as the backend's `p / norm`
            if (start_from_v) b[u] = a[u];
            v_int[i] = a[u];
            if (y0 != nullptr) y0[i] = b[u];
            if (pred_deg != nullptr) {
                pred_t += (double)pred_deg[i] * (double)b[u];
                pred_p += (double)a[u];
            }
            // an operand that is not zero on an isolated row: the run cannot pass over those rows
            if (iso.flag != nullptr && (a[u] != 0.f || b[u] != 0.f) && iso.holds(i)) atomicOr(iso.flag, 1);
            if (xg) {
                const int slot = xg_slot((int)i, xg_blk, xg_live);
                if (slot >= 0) xg[slot] = scale ? b[u] * sc[u] : b[u];
            }
        }
    }
    if (pred_deg != nullptr) pred_publish();
}

// dst[old] = src[iperm[old]] * factor
// (isolated rows that the loop passed over hold zeros: not gathered)
// A workgroup walks a CONTIGUOUS chunk of old ids.  Ties of the relabelling keep ascending ids (build_count_perm: a stable sort), and
// most ids of a power-law graph share a handful of small reference counts, so the new ids of consecutive old ids are a few
// ascending sequences per column block: the lines of src a chunk touches are its own.  (Round 3 walked the ids with a grid-wide
// stride; measured the same at scale 23, 54-55 us either way: see profiles/r04/default_rule.log for what the counters say.)
#ifndef PGH_OUT_CHUNK
#define PGH_OUT_CHUNK 4096
#endif
__global__ __launch_bounds__(kBlock) void k_permute_out_gather(const float* __restrict__ src, const int32_t* __restrict__ iperm, int64_t n,
                                                                float factor, float* __restrict__ dst, IsoTail iso = IsoTail{}) {
    constexpr int U = PGH_OUT_CHUNK / kBlock;              // index -> gather chains in flight per thread
    const bool skip_iso = iso.flag != nullptr && *iso.flag == 0;
    for (int64_t base = (int64_t)blockIdx.x * PGH_OUT_CHUNK; base < n; base += (int64_t)gridDim.x * PGH_OUT_CHUNK) {
        // (round 5: both rounds of loads unconditional -- an id past the end re-reads the last one, a passed-over row reads slot 0 and
        // is replaced by a select.  As `i < n ? iperm[i] : 0` and `dead ? 0 : src[at]` every one of the 2 x 16 loads of a thread sat under a
        // branch with a full wait behind it: 32 round trips one after the other, 42-46 us for a pass that moves 90 MB)
        int at[U];
        float x[U];
        bool dead[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * kBlock + threadIdx.x;
            at[u] = iperm[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dead[u] = skip_iso && iso.holds(at[u]);
            x[u] = src[dead[u] ? 0 : at[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * kBlock + threadIdx.x;
            if (i < n) dst[i] = dead[u] ? 0.f : x[u] * factor;
        }
    }
}
__global__ void k_permute_out(const float* __restrict__ src, const int32_t* __restrict__ perm, int64_t n_pad, int64_t n_valid,
                              float factor, float* __restrict__ dst) {
    constexpr int U = 4;                                   // independent loads per thread and round (see k_permute_in_pair)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < n_pad; i0 += stride * U) {
        int o[U];
        float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            o[u] = i < n_pad ? (perm ? perm[i] : (i < n_valid ? (int)i : -1)) : -1;
            x[u] = i < n_pad ? src[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (o[u] >= 0) dst[o[u]] = x[u] * factor;
    }
}

// ------------------------------------------------------------------------------------------------- SpMV kernels
struct BsfView {
    const int32_t*  fix_seg;   // [num_tiles] segment (index into `psum`) that receives tile t's fix-up, or -1
    const uint32_t* colf;      // packed: byte offset of the source inside its block
    const uint16_t* colf16;    // hot-only streams: byte offset / 2, [tile][lane][8] (colf is null then)
    const uint8_t*  flags8;    // [num_tiles * 64] segment-start flags of each lane's 8 entries
    const float*    val;
    const int4*     tile;
    double*         tail_carry;
    double*         head_partial;
    float*          psum;      // [num_segs] compact block partial sums: segment s of the stream -> psum[s]
    int             num_blocks;
    int             blk_size;
    int64_t         xg_base[8];
    int             tile_begin[9];
};

#ifndef PGH_SEG_DPP_FUSED
#define PGH_SEG_DPP_FUSED 1
#endif
// ---- wavefront scans on the DPP crossbar (no LDS traffic): Hillis-Steele inside the 16-lane rows, then the two
// row broadcasts (lane 15 -> next row on rows 1/3, lane 31 -> rows 2/3) that complete a 64-lane inclusive scan
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                 ROW_MASK, 0xf, false));
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
// inclusive segmented sum with head flags: keep = 0 on lanes whose value starts a new segment, 1 on lanes that continue
// the previous lane's segment.  Per step: val += keep * val[l - d]; keep *= keep[l - d]  (lanes without a source read
// 0 / 1), i.e. two DPP operand fetches, one fma and one multiply.
__device__ __forceinline__ float wave_segmented_sum(float keep, float val) {
#if PGH_SEG_DPP_FUSED
    // Round 5: the DPP operand rides in the arithmetic instruction itself.  `v_fmac_f32_dpp val, val, keep` adds keep * val[l - d];
    // `v_mul_f32_dpp keep, keep, keep` multiplies by keep[l - d]; a lane WITHOUT a source lane is not written at all (bound_ctrl off),
    // which is the "+ 0" / "* 1" the scan wants there -- 2 vector instructions per step instead of 6 (two v_mov identities, two
    // v_mov_dpp, fma, mul: 36 of the kernel's 127 vector instructions per tile, profiles/r05/bsf_partial_valu.log).  Inline asm is
    // opaque to the hazard recognizer: a VGPR written by a VALU instruction needs 2 wait states before a DPP read of it
    // (CDNA3 ISA 4.5) -- the s_nop in front of every fmac covers val (one instruction in between) and keep (two).
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(val), "+v"(keep));
    return val;
#else
#define PGH_SEG_STEP(CTRL, MASK)                                   \
    {                                                              \
        const float v2 = dpp_f32<CTRL, MASK>(0.f, val);            \
        const float k2 = dpp_f32<CTRL, MASK>(1.f, keep);           \
        val = __builtin_fmaf(v2, keep, val);                       \
        keep *= k2;                                                \
    }
    PGH_SEG_STEP(0x111, 0xf)
    PGH_SEG_STEP(0x112, 0xf)
    PGH_SEG_STEP(0x114, 0xf)
    PGH_SEG_STEP(0x118, 0xf)
    PGH_SEG_STEP(0x142, 0xa)
    PGH_SEG_STEP(0x143, 0xc)
#undef PGH_SEG_STEP
    return val;
#endif
}

// One tile = 64 * IPT consecutive entries of one column block, owned by ONE wavefront: no workgroup barrier is
// needed inside the tile loop, so the 16 wavefronts of a 1024-thread workgroup (one workgroup per CU) run
// 16 independent pipelines and hide each other's latency.
//
// What the workgroup shares is the HOT CACHE: the first kBsfHot entries of the block's (hot-first ordered)
// slice of the gather vector, copied into LDS once per launch.  On a power-law graph they serve the bulk of
// the gathers at LDS speed; only the cold remainder goes through the vector memory pipe, whose divergent-
// address miss rate (not the L2 hit rate) bounds this kernel (profiles/r01/bsf_v2_probe_scale23.log).
//
// Every lane owns IPT = 8 CONSECUTIVE entries and fetches them with 16-byte non-temporal loads (blocks are padded
// to whole tiles at build time, so every load is aligned and in range).  The stream is pre-digested at build time
// (k_bsf_pack) so that the per-entry instruction count is minimal: a column word is the BYTE offset of the source
// inside its block (LDS address = min(word, 4 * hot); cold buffer offset = word - 4 * hot, which wraps out of range
// for hot lanes so their load returns 0 without touching memory), and the segment-start flags of a lane's entries
// come as one byte per lane.  Per entry the arithmetic stage issues 6 vector ALU instructions and one LDS write
// (PGH_ENTRY below).  Software pipeline per wavefront, two register sets used ping-pong (loop unrolled twice, no
// register moves): column stream of tile t+2, gathers of tile t+1, arithmetic of tile t.  In-tile sums are f32 (a
// lane adds at most IPT terms, the 64-lane stitch is a log-depth DPP segmented scan); pieces of segments that cross
// tiles are carried in f64 and combined in a fixed order by k_bsf_fixup (deterministic, atomic-free).
// Output: the sum of the j-th segment closed inside tile t goes to psum[seg_base(t) + 1 + j] -- consecutive lanes write
// consecutive floats, no row indices are read (round 1 wrote part[block][row] through a prefetched row list: 74 MB of
// row indices per launch at scale 23 and scattered 4-byte stores; the flag byte and the row list were fetched only one
// tile ahead, which left every wavefront waiting on them -- SQ_WAIT_ANY 62 %, profiles/r02/r01_kernels_sq_counters.json).
PGH_STAMP_DECL(g_times_partial)

// LDS floats of the partial-sum body: [0, kBsfHot + 1) hot cache + one permanent zero (the slot cold lanes read); then per
// wavefront a strip: slot 0 = piece of the segment open at the tile start, slot 1 + j = closed segment j, slot T + 1 + lane =
// scratch for predicated-off writes.  One array, so that every LDS address is an offset from LDS address 0.
constexpr int kBsfLdsFloats = kBsfHot + 1 + (kBsfThreads / 64) * (64 * kIPT + 1 + 64);

// The body of the block partial sums for the workgroup `vblock` of `vgrid` (its own launch: blockIdx / gridDim; inside the
// merged front kernel of a step: the workgroup's index among the partial-sum workgroups).
// `mid`: what the kernel has to do before it may touch the stream (the loop-state test, the deferred close) runs BETWEEN the issue of
// the hot cache's loads and their arrival in LDS: its own loads are queued behind them, so the workgroup pays the longer of the two
// round trips instead of their sum (true = the loop has ended or paused: nothing to do).
#ifndef PGH_FILL_OVERLAP
#define PGH_FILL_OVERLAP 1
#endif
template <int IPT, bool HAS_VAL, bool COLD, bool W16 = false, bool DROP = false, typename Mid>
__device__ __forceinline__ void bsf_partial_body(float* __restrict__ s_lds, const BsfView& f, const float* __restrict__ xg,
                                                 const unsigned int vblock, const unsigned int vgrid, const DropView dv, Mid mid) {
    static_assert(IPT == 8, "one flag byte per lane; lanes fetch their entries as 16-byte words");
    static_assert(!(W16 && COLD), "the 16-bit stream addresses the hot cache only");
    constexpr int T = 64 * IPT;
    constexpr int WAVES = kBsfThreads / 64;
    constexpr int Q = IPT / 4;
    constexpr int STRIP = T + 1 + 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-affine block assignment: workgroups whose dispatch slots share blockIdx % 8 share an XCD (speed only)
#ifndef PGH_XCD_AFFINE
#define PGH_XCD_AFFINE 1
#endif
    const int label = PGH_XCD_AFFINE ? (vblock & 7) : (int)((vblock * 8u) / vgrid);
    const int slot = PGH_XCD_AFFINE ? (vblock >> 3) : (int)(vblock % (vgrid >> 3));
    const int b = label % f.num_blocks;
    const int per = 8 / f.num_blocks;
    const int rank = (slot * per + label / f.num_blocks) * WAVES + wave;
    const int stride = (vgrid >> 3) * per * WAVES;
    float* __restrict__ psum = f.psum;
    const int64_t base = f.xg_base[b];
    const uint32_t hot = (uint32_t)min(kBsfHot, f.blk_size);
    const uint32_t hot4 = hot << 2;
    // hot cache: 16-byte loads when the slice starts on a 16-byte boundary (one round of 8 loads per thread instead of two
    // rounds of 16)
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const bool aligned = (reinterpret_cast<uintptr_t>(xg + base) & 15) == 0;
        const f32x4* __restrict__ src4 = reinterpret_cast<const f32x4*>(xg + base);
        f32x4* __restrict__ dst4 = reinterpret_cast<f32x4*>(s_lds);
        const uint32_t hot4v = aligned ? hot >> 2 : 0u;        // 16-byte words of the slice that go through registers
        constexpr int FR = (kBsfHot / 4 + kBsfThreads - 1) / kBsfThreads;
        f32x4 fr[FR];
        const bool ahead = PGH_FILL_OVERLAP && hot4v > 0;
        if (ahead) {
#pragma unroll
            for (int k = 0; k < FR; ++k) fr[k] = src4[min((uint32_t)tid + (uint32_t)k * kBsfThreads, hot4v - 1)];
        }
        if (mid()) return;                                     // (ONE call site: the close is a few hundred instructions)
        if (ahead) {
#pragma unroll
            for (int k = 0; k < FR; ++k) {
                const uint32_t i = (uint32_t)tid + (uint32_t)k * kBsfThreads;
                if (i < hot4v) dst4[i] = fr[k];
            }
        } else {
            for (uint32_t i = tid; i < hot4v; i += kBsfThreads) dst4[i] = src4[i];
        }
        for (uint32_t i = (hot4v << 2) + tid; i < hot; i += kBsfThreads) s_lds[i] = xg[base + i];
    }
    if (tid == 0) s_lds[hot] = 0.f;
    __syncthreads();
    char* __restrict__ lds = reinterpret_cast<char*>(s_lds);
    const int strip4 = (kBsfHot + 1 + wave * STRIP) << 2;        // byte offset of this wavefront's strip
    const float* __restrict__ seg = s_lds + (kBsfHot + 1) + wave * STRIP;
    const int t_end = f.tile_begin[b + 1];
    // the block's cold slice [hot, blk_size) as a buffer: offsets outside it (every hot lane) read 0
    const __amdgpu_buffer_rsrc_t cold_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + base + hot), 0, (int)(((uint32_t)f.blk_size - hot) << 2), 0x00020000);

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    struct Stream {
        u32x4        c[Q];
        u32x4        v[Q];
        u32x4        e[DROP ? Q : 1];    // graph_dropout: the entries' indices in CSR(M^T) order (laid out like the values)
        unsigned int bits;           // segment-start flags of the lane's IPT entries
        int          seg_base;
    };
    // The stream is read through buffer descriptors rebased to the block's first tile: the lane part of an address is a
    // constant VGPR, the tile part a scalar offset (tile indices are wavefront-uniform), so the loop needs no vector
    // address arithmetic and no address temporaries (a temporary that reuses the destination of an outstanding load
    // forces a full s_waitcnt vmcnt(0)).  Offsets are 32-bit: a block holds < 2^30 entries (bsf_build checks).
    const int tb = f.tile_begin[b];
    const int64_t blk_tiles = (int64_t)(t_end - tb);
    const auto clamp32 = [](int64_t bytes) { return (int)(bytes > 0xffffffffLL ? 0xffffffffLL : bytes); };
    const __amdgpu_buffer_rsrc_t col_rsrc =
        W16 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(f.colf16 + (int64_t)tb * T), 0, clamp32(blk_tiles * T * 2), 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(f.colf + (int64_t)tb * T), 0, clamp32(blk_tiles * T * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(HAS_VAL ? f.val + (int64_t)tb * T : nullptr), 0, HAS_VAL ? clamp32(blk_tiles * T * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t edge_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(DROP ? dv.edge + (int64_t)tb * T : nullptr), 0, DROP ? clamp32(blk_tiles * T * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flag_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(f.flags8 + (int64_t)tb * 64), 0, clamp32(blk_tiles * 64), 0x00020000);
    const int lane32 = PGH_TILE_TRANSPOSE ? lane * 16 : lane * (IPT * 4);
    constexpr int kQStep = PGH_TILE_TRANSPOSE ? 1024 : 16;
    constexpr int kNT = PGH_STREAM_AUX;                    // buffer aux: 2 = non-temporal (streamed once per launch)
    auto load_stream = [&](int tile, Stream& st) __attribute__((always_inline)) {
        const int rel = tile - tb;
        if (W16) {
            st.c[0] = __builtin_amdgcn_raw_buffer_load_b128(col_rsrc, lane * 16, rel * (T * 2), kNT);
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) st.c[q] = __builtin_amdgcn_raw_buffer_load_b128(col_rsrc, lane32 + kQStep * q, rel * (T * 4), kNT);
        }
        if (HAS_VAL) {
#pragma unroll
            for (int q = 0; q < Q; ++q) st.v[q] = __builtin_amdgcn_raw_buffer_load_b128(val_rsrc, lane32 + kQStep * q, rel * (T * 4), kNT);
        }
        if (DROP) {
#pragma unroll
            for (int q = 0; q < Q; ++q) st.e[q] = __builtin_amdgcn_raw_buffer_load_b128(edge_rsrc, lane32 + kQStep * q, rel * (T * 4), kNT);
        }
        st.bits = __builtin_amdgcn_raw_buffer_load_b8(flag_rsrc, lane, rel * 64, 0);
        st.seg_base = f.tile[tile].z;
    };
    // gather stage: issue the LDS read and the buffer load of every entry WITHOUT consuming them (they are summed one
    // tile later, in the arithmetic stage), so both stay in flight across the loop back-edge.  value = h + c, no select.
    struct Gathered {
        float        h[IPT];
        float        c[IPT];
        float        v[(HAS_VAL || DROP) ? IPT : 1];      // the entry's value; with DROP: times its mask factor (0 or 1 / (1 - rate))
        unsigned int bits;
        int          seg_base;
    };
    auto gather = [&](const Stream& st, Gathered& g) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            if (DROP) {
                const float m = dropout_factor(dv.seed, (uint64_t)st.e[k >> 2][k & 3], dv.threshold, dv.keep_scale);
                g.v[k] = HAS_VAL ? __uint_as_float(st.v[k >> 2][k & 3]) * m : m;
            }
            if (W16) {                    // halfword k of the lane's 16 bytes = byte offset / 2 into the hot cache
                const uint32_t pair = st.c[0][k >> 1];
#if PGH_PROBE_GATHER == 9        // the hot gathers without bank conflicts (wrong sums): lane l reads word l of a 256-byte row, still behind the stream word
                const uint32_t off = (((k & 1) ? pair >> 31 : (pair >> 15) & 1u) << 2) + (lane << 2) + (k << 8);
#else
                uint32_t off;            // halfword k & 1 of the pair, times 2: ONE sub-dword-addressed shift (the compiler finds it for the high half only)
                if (k & 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(off) : "v"(1), "v"(pair));
                else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(off) : "v"(1), "v"(pair));
#endif
                g.h[k] = *reinterpret_cast<const float*>(lds + off);
                g.c[k] = 0.f;
                if (HAS_VAL && !DROP) g.v[k] = __uint_as_float(st.v[k >> 2][k & 3]);
                continue;
            }
            const uint32_t w = st.c[k >> 2][k & 3];
#if PGH_PROBE_GATHER == 6        // everything from the hot cache: the kernel without vector-memory gathers
            g.h[k] = *reinterpret_cast<const float*>(lds + (w % hot4 & ~3u));
            g.c[k] = 0.f;
#elif PGH_PROBE_GATHER == 8      // no gathers at all: the stream, the arithmetic and the output stage
            g.h[k] = __uint_as_float(w & 0xffffu);
            g.c[k] = 0.f;
#else
            g.h[k] = *reinterpret_cast<const float*>(lds + min(w, hot4));
            // COLD = false: the stream holds hot entries only (the cold ones live in the propagation-blocking image)
            g.c[k] = COLD ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cold_rsrc, w - hot4, 0, PGH_COLD_AUX)) : 0.f;
#endif
            if (HAS_VAL && !DROP) g.v[k] = __uint_as_float(st.v[k >> 2][k & 3]);
        }
        g.bits = st.bits;            // fetched with the stream, three tiles ahead: as old as the column words used above
        g.seg_base = st.seg_base;
    };
    const int scr4 = strip4 + ((T + 1 + lane) << 2);
    // arithmetic of one tile: lane-local flags, branch-free f32 segmented sum
    auto reduce = [&](const Gathered& g0, int t) __attribute__((always_inline)) {
        const int bits = (int)g0.bits;
        const int mine = __popc(g0.bits);
        const int incl = wave_inclusive_sum(mine);        // flags in lanes <= this one
        const int before = incl - mine;
        const int closed = __builtin_amdgcn_readlane(incl, 63) - 1;      // segments that start and end inside the tile
        // The j-th flag of the tile (j = before + #flags below entry k) opens segment j and closes segment j - 1: the
        // running sum goes to slot j (slot 0 = the segment that was open when the tile started).  A lane's first flag
        // closes a segment that may have begun in earlier lanes; their contribution is added after the stitch below.
        // Writes of entries without a flag are steered to the lane's scratch slot instead of being branched around.
        const int slot0 = strip4 + (before << 2);
        int o4 = slot0;
        int accb = 0;
        // per entry: m = flag ? -1 : 0; address = flag ? o4 : scratch; store acc; o4 += 4 * flag; acc = flag ? 0 : acc
#define PGH_ENTRY(K)                                                                                   \
        {                                                                                              \
            int m, a;                                                                                  \
            asm("v_bfe_i32 %0, %1, " #K ", 1" : "=v"(m) : "v"(bits));                                  \
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(a) : "v"(m), "v"(o4), "v"(scr4));                    \
            if (!(PGH_PROBE_SKIP & 2) || K == 7) *reinterpret_cast<int*>(lds + a) = accb;              \
            asm("v_mad_i32_i24 %0, %1, -4, %2" : "=v"(o4) : "v"(m), "v"(o4));                          \
            asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(accb) : "v"(m), "v"(accb));                           \
            float xv = COLD ? g0.h[K] + g0.c[K] : g0.h[K];      /* (x + 0.f is not folded: -0) */     \
            if (HAS_VAL || DROP) xv *= g0.v[K];                                                        \
            accb = __builtin_bit_cast(int, __builtin_bit_cast(float, accb) + xv);                      \
        }
        PGH_ENTRY(0) PGH_ENTRY(1) PGH_ENTRY(2) PGH_ENTRY(3) PGH_ENTRY(4) PGH_ENTRY(5) PGH_ENTRY(6) PGH_ENTRY(7)
#undef PGH_ENTRY
        // ---- stitch segments that cross lane boundaries: the sum after a lane's last flag continues into the next
        // lane unless that lane starts with ... any flag of its own (head flag = lane has a flag)
        const float acc = __builtin_bit_cast(float, accb);
        const float val = wave_segmented_sum(mine ? 0.f : 1.f, acc);
        const float ev = dpp_f32<0x138, 0xf>(0.f, val);   // wave_shr:1: what the earlier lanes hold of the segment this
        if (bits != 0) {                                  // lane's first flag closes (lane l-1 ends in segment before-1)
            float* slot = reinterpret_cast<float*>(lds + slot0);
            *slot += ev;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#if !(PGH_PROBE_SKIP & 4)
        if (lane == 0 && closed >= 0) f.head_partial[t] = (double)seg[0];   // the segment that was open at the tile start
        if (lane == 63) f.tail_carry[t] = (double)val;    // piece of the segment still open at the end of the tile
#else
        if (val == 123.456f) f.tail_carry[t] = 0.0;
#endif
        // ---- closed segments -> their slots of the compact partial-sum array: segment seg_base + 1 + j <- strip slot 1 + j
        // (a fixed number of predicated stores: no loop of unknown length between the counted waits)
#if !(PGH_PROBE_SKIP & 1)
        float* __restrict__ dst = psum + (g0.seg_base + 1);
        // (a tile closes ~45 segments on the bench graph: one predicated store serves it; the seven others hide behind ONE
        // wave-uniform test instead of a vector compare and an exec branch each)
        if (lane < closed) dst[lane] = seg[1 + lane];
        if (closed > 64) {
#pragma unroll
            for (int k = 1; k < IPT; ++k)
                if (64 * k + lane < closed) dst[64 * k + lane] = seg[1 + 64 * k + lane];
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    int t = f.tile_begin[b] + rank;
    if (t >= t_end) return;
    // The pipeline issues its loads UNCONDITIONALLY (tile indices are clamped to the block's last tile; the clamped
    // results are never consumed): a fixed number of vector-memory operations per iteration lets the compiler emit
    // counted s_waitcnt vmcnt(N) instead of draining the queue.  In flight per wavefront at iteration i: the column
    // streams of tiles i+2 and i+3 (what bounds the kernel without gathers is bytes in flight x latency, so the stream
    // runs two tiles ahead), the gathers of tile i+1, the arithmetic of tile i.  Three stream register sets and two
    // gather sets rotate without moves: the loop is unrolled six times.
    const int t_last = t_end - 1;
    // stream register sets (tiles in flight ahead of the gather stage); build parameter for measurements
#ifndef PGH_STREAM_SETS
#define PGH_STREAM_SETS 3
#endif
    constexpr int NS = PGH_STREAM_SETS;
    static_assert(NS == 3 || NS == 4 || NS == 6, "the unrolled rotation below is written for 3, 4 or 6 stream sets");
    Stream s0, s1, s2, s3, s4, s5;
    Gathered g0, g1;
    load_stream(t, s0);
    gather(s0, g0);
    load_stream(min(t + stride, t_last), s1);
    load_stream(min(t + 2 * stride, t_last), s2);
    if (NS >= 4) load_stream(min(t + 3 * stride, t_last), s3);
    if (NS >= 6) {
        load_stream(min(t + 4 * stride, t_last), s4);
        load_stream(min(t + 5 * stride, t_last), s5);
    }
#define PGH_STEP(SL, SG, GN, GC)                             \
    load_stream(min(t + NS * stride, t_last), SL);           \
    gather(SG, GN);                                          \
    reduce(GC, t);                                           \
    t += stride;                                             \
    if (t >= t_end) break;
    for (;;) {
        if (NS == 3) {
            PGH_STEP(s0, s1, g1, g0)
            PGH_STEP(s1, s2, g0, g1)
            PGH_STEP(s2, s0, g1, g0)
            PGH_STEP(s0, s1, g0, g1)
            PGH_STEP(s1, s2, g1, g0)
            PGH_STEP(s2, s0, g0, g1)
        } else if (NS == 4) {
            PGH_STEP(s0, s1, g1, g0)
            PGH_STEP(s1, s2, g0, g1)
            PGH_STEP(s2, s3, g1, g0)
            PGH_STEP(s3, s0, g0, g1)
        } else {
            PGH_STEP(s0, s1, g1, g0)
            PGH_STEP(s1, s2, g0, g1)
            PGH_STEP(s2, s3, g1, g0)
            PGH_STEP(s3, s4, g0, g1)
            PGH_STEP(s4, s5, g1, g0)
            PGH_STEP(s5, s0, g0, g1)
        }
    }
#undef PGH_STEP
}

template <int IPT, bool HAS_VAL, bool COLD, bool W16 = false, bool DROP = false>
__global__ __launch_bounds__(kBsfThreads) void k_bsf_partial(BsfView f, const float* __restrict__ xg,
                                                              const LoopState* __restrict__ state, PendingClose pc, DropView dv = DropView{}) {
    __shared__ __attribute__((aligned(16))) float s_lds[kBsfLdsFloats];
    __shared__ double s_close[16];
    PGH_STAMP_BEGIN(g_times_partial)
    bsf_partial_body<IPT, HAS_VAL, COLD, W16, DROP>(s_lds, f, xg, blockIdx.x, gridDim.x, dv, [&]() __attribute__((always_inline)) {
        if (state != nullptr && state->done) return true;
        // the previous step's close, if the loop driver left it to this kernel
        if (pc.active && run_pending_close(pc, s_close)) return true;
        if (pc.first_pred && blockIdx.x == 0 && threadIdx.x == 0) first_prediction(pc);      // read by the finish launch of this step
        return false;
    });
#if PGH_PROBE_TIMES
    // wavefronts leave one by one: the workgroup's end = the latest of them (the clock only grows, so the maximum over
    // launches is the last launch's)
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) atomicMax(&g_times_partial[2 * blockIdx.x + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// (Round 3 measured the block partial sums and phase A as ONE launch -- workgroups of both roles in one grid, the cross-tile
// fix-ups either inside k_pb_finish's work items or behind a device counter at the end of the phase A workgroups: 126-132 us
// against 58 + 73 + the boundary for the two launches, whatever the order of the roles.  A CU moves ~21 GB/s whatever it runs
// (the chip's 5.4 TB/s copy ceiling is 256 such CUs), a phase A workgroup alone on the chip takes as long as all of them
// together, so nothing overlaps by putting the VALU-bound role and the bandwidth-bound role on different CUs; and both need
// most of a CU's LDS, so they cannot share one.  profiles/r03/front_merge_rejected.log.)
// build time: where the fix-up of tile t goes (index into the partial vectors, -1 = nothing to fix), so that the
// per-iteration kernel below needs no dependent loads
__global__ void k_bsf_fixlist(const int4* __restrict__ tile, const int32_t* __restrict__ seg_row, int num_tiles,
                              int32_t* __restrict__ fix_seg) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < num_tiles; t += gridDim.x * blockDim.x) {
        const int4 ti = tile[t];
        // ti.z = the segment open when the tile starts; it closes here when the tile holds a flag (ti.w >= 0)
        fix_seg[t] = (ti.w >= 0 && ti.z >= 0 && seg_row[ti.z] >= 0) ? ti.z : -1;
    }
}

// segments that cross tiles (bsf_fixup_tiles, pgh_kernels.h) as a launch of their own: graphs without a cold image
__global__ __launch_bounds__(WG) void k_bsf_fixup(FixView f, const LoopState* __restrict__ state) {
    if (state != nullptr && state->done) return;
    bsf_fixup_tiles(f, blockIdx.x * WG, gridDim.x * WG);
}

// fold the B block partials, apply the filter epilogue, write the next gather vector
template <int MODE, int B>
__global__ __launch_bounds__(WG) void k_bsf_combine(RowSums rs, int64_t n_out,
                                                     const float* __restrict__ dst_scale, EpiParams ep,
                                                     const LoopState* __restrict__ state, double* __restrict__ partial_sum,
                                                     double* __restrict__ partial_delta) {
    __shared__ double s_red[4];
    double scale = 1.0;
    if (state != nullptr) {
        if (state->done) return;
        scale = state->scale;
    }
    const float a_eff = (float)(ep.a * scale);
    double sum_y = 0.0, delta = 0.0;
    const int64_t stride = (int64_t)gridDim.x * WG;
    // (no run-time branch around a load: epi_load_z, block_row_sum)
    const char* const zero = reinterpret_cast<const char*>(rs.psum + rs.zero_at);
    const bool has_ds = dst_scale != nullptr;
    const char* const ds_base = has_ds ? reinterpret_cast<const char*>(dst_scale) : zero;
    for (int64_t i = blockIdx.x * (int64_t)WG + threadIdx.x; i < n_out; i += stride) {
        EpiOps ops = epi_load_z<MODE>(ep, (int)i, zero);
        const float dsc = *reinterpret_cast<const float*>(ds_base + (has_ds ? (uint32_t)i << 2 : 0u));
        if (ep.xg_out != nullptr) ops.slot = xg_slot((int)i, ep.xg_blk, ep.xg_live, ep.xg_hot, ep.xg_cold);
        double s = block_row_sum<B>(rs, i);
        s = has_ds ? s * (double)dsc : s;
        epi_apply_z<MODE>(ep, ops, a_eff, (int)i, (float)s, true, sum_y, delta);
    }
    const double bs = block_reduce_256<0>(sum_y, s_red);
    if (threadIdx.x == 0) partial_sum[blockIdx.x] = bs;
    if (MODE == EPI_POLY) {
        const double bd = ep.err_linf ? block_reduce_256<1>(delta, s_red) : block_reduce_256<0>(delta, s_red);
        if (threadIdx.x == 0) partial_delta[blockIdx.x] = bd;
    }
}

BsfView view_of(const BsfFormat& f) {
    BsfView v;
    v.fix_seg = f.fix_seg;
    v.colf = f.colf;
    v.colf16 = f.colf16;
    v.flags8 = f.flags8;
    v.val = f.val;
    v.tile = f.tile;
    v.tail_carry = f.tail_carry;
    v.head_partial = f.head_partial;
    v.psum = f.psum;
    v.num_blocks = f.num_blocks;
    v.blk_size = f.blk_size;
    for (int i = 0; i < 8; ++i) v.xg_base[i] = f.xg_base[i];
    for (int i = 0; i < 9; ++i) v.tile_begin[i] = f.tile_begin[i];
    return v;
}

int env_int(const char* name, int fallback) {
    const char* s = getenv(name);
    return s ? atoi(s) : fallback;
}

}  // namespace

namespace pgh {

int bsf_combine_grid(int64_t n_out) {
    int64_t blocks = (n_out + (int64_t)WG * 4 - 1) / ((int64_t)WG * 4);
    const int64_t cap = (int64_t)rt().num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// Enqueue M^T-times-gather-vector in the blocked format followed by the MODE epilogue.
// xg: gather vector in the graph's internal id space, already multiplied by src_scale when the format has one.
// Block partials of sum(y) / delta land in rt().d_partials like the row-major path; *num_partials receives their count.
// stages: 1 = block partial sums only (hot-only streams read just the first `hot` slots of every block of the gather
// vector), 2 = the cold image's phase A + the cross-tile fix-ups, 0 = both.  A partitioned run overlaps the exchange of the
// cold part of the gather vector with stage 1.
int bsf_launch_partial(pgh_graph_s* g, const float* xg, const LoopState* state, int stage) {
    Runtime& r = rt();
    BsfFormat& f = g->bsf;
    const BsfView v = view_of(f);
    const int main_grid = r.num_cus;                  // one workgroup per CU; a multiple of 8 (XCD-affine blocks)
    // the previous step's close rides in this launch when the loop driver deferred it (PendingClose, pgh_kernels.h)
    PendingClose pc = pending_close_slot();
    if (stage == 2 || state == nullptr || pc.state != state) pc.active = 0, pc.first_pred = 0;
    else pending_close_slot().active = 0, pending_close_slot().first_pred = 0;      // consumed
    if (stage != 2) {
        ProfScope prof(PGH_K_SPMV);
        const DropView dv = bsf_dropout_view(f.drop_edge);
        if (dv.edge != nullptr) {                      // graph_dropout: the same kernels with the mask factor per entry
            if (f.colf16 != nullptr) {
                if (f.val) k_bsf_partial<kIPT, true, false, true, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc, dv);
                else k_bsf_partial<kIPT, false, false, true, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc, dv);
            } else if (f.pb.enabled && !f.pb.k1_cold) {
                if (f.val) k_bsf_partial<kIPT, true, false, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc, dv);
                else k_bsf_partial<kIPT, false, false, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc, dv);
            } else {
                if (f.val) k_bsf_partial<kIPT, true, true, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc, dv);
                else k_bsf_partial<kIPT, false, true, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc, dv);
            }
        } else if (f.colf16 != nullptr) {
            if (f.val) k_bsf_partial<kIPT, true, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc);
            else k_bsf_partial<kIPT, false, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc);
        } else if (f.pb.enabled && !f.pb.k1_cold) {
            if (f.val) k_bsf_partial<kIPT, true, false><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc);
            else k_bsf_partial<kIPT, false, false><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc);
        } else {
            if (f.val) k_bsf_partial<kIPT, true, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc);
            else k_bsf_partial<kIPT, false, true><<<main_grid, kBsfThreads, 0, r.stream>>>(v, xg, state, pc);
        }
    }
    if (stage != 2) {
        PGH_STAMP_DUMP(g_times_partial, main_grid, "k_bsf_partial")
    }
    if (stage == 1) {
        PGH_HIP(hipGetLastError());
        return 0;
    }
    FixView fix;
    fix.fix_seg = f.fix_seg;
    fix.tile = f.tile;
    fix.tail_carry = f.tail_carry;
    fix.head_partial = f.head_partial;
    fix.psum = f.psum;
    fix.num_tiles = f.num_tiles;
    if (f.pb.enabled && f.pb.num_tasks > 0) {
        // cold entries, phase A: gathered values -> bin order; its workgroups close the cross-tile segments first
        PGH_TRY(pb_launch_gather(g, xg, state, fix));
    } else {
        ProfScope prof(PGH_K_FIXUP);
        int fix_grid = (f.num_tiles + WG - 1) / WG;
        if (fix_grid < 1) fix_grid = 1;
        if (fix_grid > 1024) fix_grid = 1024;
        k_bsf_fixup<<<fix_grid, WG, 0, r.stream>>>(fix, state);
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

RowSums row_sums_of(const BsfFormat& f) {
    RowSums rs;
    rs.meta = f.meta;
    rs.psum = f.psum;
    rs.words = f.meta_words;
    rs.num_blocks = f.num_blocks;
    rs.zero_at = (unsigned int)f.num_segs;            // past the last segment: zeroed at build time, never written
    return rs;
}

template <int MODE>
int bsf_launch_combine(pgh_graph_s* g, const EpiParams& ep, const LoopState* state, int* num_partials) {
    Runtime& r = rt();
    BsfFormat& f = g->bsf;
    const RowSums rs = row_sums_of(f);
    // graphs with a cold image: phase B and the epilogue are one launch (pgh_pb.hip)
    if (f.pb.enabled) return pb_launch_finish<MODE>(g, rs, ep, state, num_partials);
    const int cgrid = bsf_combine_grid(f.n_out);
    double* psum = r.d_partials;
    double* pdel = r.d_partials + kMaxPartials;
    {
        ProfScope prof(PGH_K_COMBINE);
        switch (f.num_blocks) {
            case 1: k_bsf_combine<MODE, 1><<<cgrid, WG, 0, r.stream>>>(rs, f.n_out, f.dst_scale, ep, state, psum, pdel); break;
            case 2: k_bsf_combine<MODE, 2><<<cgrid, WG, 0, r.stream>>>(rs, f.n_out, f.dst_scale, ep, state, psum, pdel); break;
            case 4: k_bsf_combine<MODE, 4><<<cgrid, WG, 0, r.stream>>>(rs, f.n_out, f.dst_scale, ep, state, psum, pdel); break;
            default: k_bsf_combine<MODE, 8><<<cgrid, WG, 0, r.stream>>>(rs, f.n_out, f.dst_scale, ep, state, psum, pdel); break;
        }
    }
    PGH_HIP(hipGetLastError());
    if (num_partials) *num_partials = cgrid;
    return 0;
}

template <int MODE>
int bsf_launch(pgh_graph_s* g, const EpiParams& ep, const float* xg, const LoopState* state, int* num_partials,
               hipEvent_t before_combine) {
    PGH_TRY(bsf_launch_partial(g, xg, state, 0));
    // the epilogue is the first consumer of the previous step's scalars (quotient, done flag): the partial sums above
    // may run while the previous step's residual / close kernels are still in flight on the side stream
    if (before_combine != nullptr) PGH_HIP(hipStreamWaitEvent(rt().stream, before_combine, 0));
    return bsf_launch_combine<MODE>(g, ep, state, num_partials);
}

// ---- small graphs: everything of a recursive step after the block partial sums as ONE launch of ONE workgroup ------------------
// A graph of a few thousand rows is launch-bound: partial sums -> fix-ups -> epilogue -> residual are four dependent launches of
// 3-6 us each for well under a microsecond of work (profiles/r02/small_window_sweep.log).  Up to kSmallTailRows rows (one column
// block, no cold image) the last three run here: cross-tile fix-ups, the MODE epilogue of every row (y, next gather vector, sum(y)),
// the residual |y * inv - x * scale| (supervised.py:93-138) and the close (ConvergenceManager, convergence.py:77-101; the same
// close_outcome / close_commit as every other loop) -- one workgroup, __syncthreads between the stages (the wavefronts of a
// workgroup share the CU's L1, so what one stage wrote the next one reads).  Two launches per iteration instead of four.
constexpr int kSmallTailThreads = 1024;
constexpr int kSmallTailRows = 12288;

template <int KIND>      // 0 sum, 1 max; every thread returns the result; fixed order (wavefronts 0 .. 15): deterministic
__device__ __forceinline__ double small_block_reduce(double v, double* s16) {
    v = KIND == 0 ? wave_reduce_sum(v) : wave_reduce_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s16[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = s16[0];
#pragma unroll
    for (int w = 1; w < kSmallTailThreads / 64; ++w) r = KIND == 0 ? r + s16[w] : fmax(r, s16[w]);
    return r;
}

template <int MODE>
__global__ __launch_bounds__(kSmallTailThreads) void k_small_tail(FixView fix, RowSums rs, int n_out, const float* __restrict__ dst_scale,
                                                                  EpiParams ep, const float* __restrict__ x_prev, PendingClose pc) {
    __shared__ double s16[kSmallTailThreads / 64];
    LoopState* state = pc.state;
    // What bounds this kernel is its chain of dependent loads (~0.7 us each: everything was written by other CUs a moment ago), so
    // every load that does not depend on the fix-ups is issued BEFORE them: the state, the first round's map words and operands,
    // and the fix-up's own two index loads side by side.
    const int done = state->done;
    const double scale = state->scale;
    // a thread owns rows t, t + 1024, ...: U of them in flight per round (map word -> segment sum is a dependent pair of loads),
    // their y and x kept in registers for the residual, which needs 1 / sum(y) first
    constexpr int R = kSmallTailRows / kSmallTailThreads, U = 4;
    static_assert(R % U == 0, "rounds of U rows");
    float yk[R], xk[R];
    RowLookup<1> q[U];
    EpiOps ops[U];
    float ds[U];
    auto load_round = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = min((int)threadIdx.x + (c + u) * kSmallTailThreads, n_out - 1);      // clamped: loads stay unconditional
            row_lookup_meta<1>(rs, row, q[u]);
            ops[u] = epi_load<MODE>(ep, row);
            ds[u] = dst_scale != nullptr ? dst_scale[row] : 1.f;
            xk[c + u] = (MODE != EPI_POLY && pc.check) ? x_prev[row] : 0.f;
        }
    };
    load_round(0);
    {
        // cross-tile fix-ups (bsf_fixup_tiles' arithmetic and order; here the two index loads of a tile do not wait for each other)
        const int lane = threadIdx.x & 63;
        for (int t0 = 0; t0 < fix.num_tiles; t0 += kSmallTailThreads) {
            const int t = t0 + (int)threadIdx.x;
            const int tc = min(t, fix.num_tiles - 1);
            const int seg = fix.fix_seg[tc];
            const int dst = (t < fix.num_tiles && !done) ? seg : -1;          // a finished loop: the partial sums are left alone
            const int chain = fix.tile[tc].w;
            const double head = fix.head_partial[tc];
            const int first = dst >= 0 ? chain : 0;
            const int len = dst >= 0 ? t - first : 0;
            const bool is_long = len >= 32;
            if (dst >= 0 && !is_long) {
                double total = 0.0;
                for (int k = first; k < t; ++k) total += fix.tail_carry[k];
                total += head;
                fix.psum[dst] = (float)total;
            }
            unsigned long long todo = __ballot(is_long);
            while (todo != 0ULL) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1ULL;
                const int c_first = __shfl(first, src, 64), c_t = __shfl(t, src, 64);
                double part_sum = 0.0;
                for (int k = c_first + lane; k < c_t; k += 64) part_sum += fix.tail_carry[k];
                part_sum = wave_reduce_sum(part_sum);
                const double total = __shfl(part_sum, 0, 64) + __shfl(head, src, 64);
                if (lane == src) fix.psum[dst] = (float)total;
            }
        }
    }
    if (done) return;                                    // workgroup-uniform
    __syncthreads();
    const float a_eff = (float)(ep.a * scale);
    double sum_y = 0.0, delta = 0.0;
#pragma unroll
    for (int c = 0; c < R; c += U) {
        if (c * kSmallTailThreads >= n_out) {              // workgroup-uniform: no row left for this round
#pragma unroll
            for (int u = 0; u < U; ++u) yk[c + u] = xk[c + u] = 0.f;
            continue;
        }
        if (c > 0) load_round(c);
        float v[U][1];
#pragma unroll
        for (int u = 0; u < U; ++u) row_lookup_vals<1>(rs, min((int)threadIdx.x + (c + u) * kSmallTailThreads, n_out - 1), q[u], v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = (int)threadIdx.x + (c + u) * kSmallTailThreads;
            yk[c + u] = 0.f;
            if (row < n_out) {
                double sum = (double)v[u][0];
                if (dst_scale != nullptr) sum *= (double)ds[u];
                yk[c + u] = epi_apply<MODE>(ep, ops[u], a_eff, row, (float)sum, sum_y, delta);
            } else {
                xk[c + u] = 0.f;
            }
        }
    }
    const double S = small_block_reduce<0>(sum_y, s16);
    double err = 0.0;
    if (MODE == EPI_POLY) {
        // closed-form filters stop on the change of the accumulated result (abstract_filters.py:232-246), which the epilogue summed
        err = pc.err_kind == PGH_ERR_LINF ? small_block_reduce<1>(delta, s16) : small_block_reduce<0>(delta, s16);
    } else if (pc.check) {
        const int linf = pc.err_kind == PGH_ERR_LINF;
        const double inv = pc.use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;      // abstract_filters.py:133-134
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < R; ++c) {                                               // rows past the end hold 0 - 0
            const double d = fabs((double)yk[c] * inv - (double)xk[c] * scale);
            acc = linf ? fmax(acc, d) : acc + d;
        }
        err = linf ? small_block_reduce<1>(acc, s16) : small_block_reduce<0>(acc, s16);
    }
    if (threadIdx.x == 0) close_commit(pc, close_outcome(pc, S, err, 0.0, 0.0, 0.0));
}

bool bsf_small_tail_usable(const pgh_graph_s* g) {
    const char* sw = getenv("PGH_SMALL_TAIL");             // read per call: the parity tests run both sequences in one process
    const bool off = sw != nullptr && atoi(sw) == 0;
    const BsfFormat& f = g->bsf;
    return !off && f.enabled && !f.pb.enabled && f.num_blocks == 1 && f.meta != nullptr && f.n_out >= 1 && f.n_out <= kSmallTailRows &&
           f.n_out == f.n_src_pad;
}

// block partial sums + the tail above: one recursive step of a small graph, closed (pc) in the same launch
template <int MODE>
int bsf_launch_small(pgh_graph_s* g, const EpiParams& ep, const float* xg, const float* x_prev, const LoopState* state,
                     const PendingClose& pc) {
    PGH_TRY(bsf_launch_partial(g, xg, state, 1));
    BsfFormat& f = g->bsf;
    FixView fix;
    fix.fix_seg = f.fix_seg;
    fix.tile = f.tile;
    fix.tail_carry = f.tail_carry;
    fix.head_partial = f.head_partial;
    fix.psum = f.psum;
    fix.num_tiles = f.num_tiles;
    PendingClose rec = pc;
    rec.res_mode = 0;
    {
        ProfScope prof(PGH_K_COMBINE);
        k_small_tail<MODE><<<1, kSmallTailThreads, 0, rt().stream>>>(fix, row_sums_of(f), f.n_out, f.dst_scale, ep, x_prev, rec);
    }
    PGH_HIP(hipGetLastError());
    return 0;
}
template int bsf_launch_small<EPI_AXPBY>(pgh_graph_s*, const EpiParams&, const float*, const float*, const LoopState*, const PendingClose&);
template int bsf_launch_small<EPI_ABSORB>(pgh_graph_s*, const EpiParams&, const float*, const float*, const LoopState*, const PendingClose&);
template int bsf_launch_small<EPI_POLY>(pgh_graph_s*, const EpiParams&, const float*, const float*, const LoopState*, const PendingClose&);

template int bsf_launch_combine<EPI_AXPBY>(pgh_graph_s*, const EpiParams&, const LoopState*, int*);
template int bsf_launch_combine<EPI_ABSORB>(pgh_graph_s*, const EpiParams&, const LoopState*, int*);
template int bsf_launch_combine<EPI_POLY>(pgh_graph_s*, const EpiParams&, const LoopState*, int*);
template int bsf_launch<EPI_PLAIN>(pgh_graph_s*, const EpiParams&, const float*, const LoopState*, int*, hipEvent_t);
template int bsf_launch<EPI_AXPBY>(pgh_graph_s*, const EpiParams&, const float*, const LoopState*, int*, hipEvent_t);
template int bsf_launch<EPI_ABSORB>(pgh_graph_s*, const EpiParams&, const float*, const LoopState*, int*, hipEvent_t);
template int bsf_launch<EPI_POLY>(pgh_graph_s*, const EpiParams&, const float*, const LoopState*, int*, hipEvent_t);

// original-space source-side vector -> internal (relabelled, padded) space, optionally times src_scale
int bsf_to_internal(pgh_graph_s* g, const float* src, float* dst, bool prescale, float hole) {
    BsfFormat& f = g->bsf;
    const bool own = (dst == f.xg);          // the engine's own gather vector may be stored trimmed
    k_permute_in<<<blocks_for(f.n_src_pad), kBlock, 0, rt().stream>>>(src, f.perm, prescale ? f.src_scale : nullptr, f.n_src_pad,
                                                                    f.n_src, hole, dst, f.blk_size, own ? f.xg_live : 0);
    PGH_HIP(hipGetLastError());
    return 0;
}

// recursive loops on square relabelled graphs: both loop operands (and the scaled gather vector) in one launch
bool bsf_can_norm_on_device(const pgh_graph_s* g) { return bsf_can_bring_pair(g) && g->bsf.iperm != nullptr && rt().num_cus * 16 <= kMaxPartials; }
bool bsf_can_bring_pair(const pgh_graph_s* g) {
    const BsfFormat& f = g->bsf;
    return f.enabled && f.relabelled && f.perm != nullptr && f.n_out == f.n_src_pad;
}
int bsf_bring_pair(pgh_graph_s* g, const float* v, const float* ranks, float* v_int, float* y0, bool want_xg, float in_norm,
                   bool start_from_v, bool watch_iso, LoopState* init_state, LoopAux* init_aux, bool* state_inited, const float* pred_deg) {
    if (state_inited != nullptr) *state_inited = false;
    BsfFormat& f = g->bsf;
    IsoTail iso = IsoTail{};
    const bool watching = watch_iso && f.iso_flag != nullptr;      // the flag starts at 0; any non-zero operand on an isolated row raises it
    bool flag_cleared = false;
    if (watching) iso = iso_tail_of(f);
    const int* seed_count = nullptr;
    const bool norm_here = in_norm < 0.f;              // the caller left GraphFilter.rank's norm to this pass
    PGH_CHECK(!norm_here || (f.iperm != nullptr && init_aux != nullptr), "bsf_bring_pair: the norm cannot be computed on this layout");
    // (small graphs: the gather pass takes a few microseconds, two more launches would cost more -- unless the scan also saves the
    // caller a reduction and a host round trip for the norm)
    if (f.iperm != nullptr && (norm_here || (f.n_src_pad >= (1 << 21) && env_int("PGH_SEED_LIST", 1) != 0))) {
        if (f.seed_list == nullptr) {
            PGH_HIP(pooled_malloc(&f.seed_list, sizeof(int32_t) * (size_t)kSeedListCap));
            PGH_HIP(pooled_malloc(&f.seed_count, sizeof(int) * 2));          // two counters, used in turn: a run's closing launch clears the
            PGH_HIP(hipMemsetAsync(f.seed_count, 0, sizeof(int) * 2, rt().stream));      // other one for the run after it
            f.seed_turn = 0;
        }
        int* count_now = f.seed_count + f.seed_turn;
        int* count_next = f.seed_count + (1 - f.seed_turn);
        f.seed_turn = 1 - f.seed_turn;
        const int64_t xg_len = want_xg ? (f.xg_live > 0 ? (int64_t)f.num_blocks * f.xg_live : (int64_t)f.n_src_pad) + 1 : 0;
        const int64_t span = f.n_src_pad > xg_len ? f.n_src_pad : xg_len;
        const int scan_grid = blocks_for(span);         // <= 16 workgroups per CU: within kMaxPartials
        double* norm_partials = norm_here ? rt().d_partials : nullptr;     // (no step is in flight: the region is free)
        k_pair_scan<<<scan_grid, kBlock, 0, rt().stream>>>(v, start_from_v ? nullptr : ranks, f.n_out_orig, f.n_src_pad, xg_len, v_int, y0,
                                                           want_xg ? f.xg : nullptr, f.seed_list, count_now, norm_partials, init_aux);
        k_scan_close<<<1, kBlock, 0, rt().stream>>>(norm_partials, scan_grid, init_state, init_aux, watching ? f.iso_flag : nullptr, count_next);
        flag_cleared = watching;
        if (state_inited != nullptr && init_state != nullptr) *state_inited = true;
        seed_count = count_now;
    }
    if (watching && !flag_cleared) PGH_HIP(hipMemsetAsync(f.iso_flag, 0, sizeof(int), rt().stream));
    // (the sums of the first step's prediction need the closing launch of the scan to have cleared them)
    const bool predicting = pred_deg != nullptr && init_aux != nullptr && seed_count != nullptr && init_state != nullptr;
    k_permute_in_pair<<<blocks_for(f.n_src_pad), kBlock, 0, rt().stream>>>(v, ranks, f.perm, f.src_scale, f.n_src_pad, v_int, y0,
                                                                          want_xg ? f.xg : nullptr, f.blk_size, f.xg_live, in_norm,
                                                                          start_from_v ? 1 : 0, iso, seed_count, f.seed_list, f.iperm,
                                                                          norm_here ? init_aux : nullptr, predicting ? pred_deg : nullptr,
                                                                          predicting ? init_aux : nullptr);
    PGH_HIP(hipGetLastError());
    return 0;
}

IsoTail iso_tail_of(const BsfFormat& f) {
    IsoTail t = IsoTail{};
    if (!f.has_iso || f.iso_flag == nullptr) return t;
    t.flag = f.iso_flag;
    t.blk = f.blk_size;
    t.shift = -1;
    if (f.blk_size > 0 && (f.blk_size & (f.blk_size - 1)) == 0)
        for (t.shift = 0; (1 << t.shift) < f.blk_size; ++t.shift) {}
    t.num_blocks = f.iso_row_blocks;
    for (int b = 0; b < 8; ++b) t.begin[b] = b < f.iso_row_blocks ? f.iso_begin[b] : f.blk_size;
    return t;
}

// the flag back to "process every row" (what every launch outside a watched recursive loop assumes)
int iso_flag_release(pgh_graph_s* g) {
    BsfFormat& f = g->bsf;
    if (f.iso_flag != nullptr) PGH_HIP(hipMemsetAsync(f.iso_flag, 0xff, sizeof(int), rt().stream));
    return 0;
}

// gather vector of a square graph from an internal-space vector and an internal-space scale: xg = y * scale (stored in
// the engine's own, possibly trimmed, layout)
__global__ void k_make_gather(const float* __restrict__ y, const float* __restrict__ scale, int64_t n_pad, float* __restrict__ xg,
                              int xg_blk, int xg_live) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = xg_slot((int)i, xg_blk, xg_live);
        if (slot >= 0) xg[slot] = y[i] * scale[i];
    }
}
int bsf_make_gather(pgh_graph_s* g, const float* y_int, const float* scale_int, float* xg_out) {
    BsfFormat& f = g->bsf;
    k_make_gather<<<blocks_for(f.n_src_pad), kBlock, 0, rt().stream>>>(y_int, scale_int, f.n_src_pad, xg_out != nullptr ? xg_out : f.xg, f.blk_size,
                                                                       f.xg_live);
    PGH_HIP(hipGetLastError());
    return 0;
}

// ---- graph_dropout on the blocked layouts -------------------------------------------------------------------------------------
namespace {
double   g_drop_rate = 0.0;
uint64_t g_drop_seed = 0;
bool     g_drop_on = false;
__global__ void k_index_bits(float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = __int_as_float((int)i);
}
}  // namespace
void bsf_set_dropout(double rate, uint64_t seed) {
    g_drop_on = rate > 0.0;
    g_drop_rate = rate;
    g_drop_seed = seed;
}
void bsf_clear_dropout() { g_drop_on = false; }
DropView bsf_dropout_view(const int32_t* edge) {
    DropView dv{};
    if (g_drop_on && edge != nullptr) {
        dv.edge = edge;
        dv.seed = g_drop_seed;
        dv.threshold = (uint32_t)floor(g_drop_rate * 4294967296.0);
        dv.keep_scale = (float)(1.0 / (1.0 - g_drop_rate));
    }
    return dv;
}
bool bsf_dropout_usable(const pgh_graph_s* g) {
    static const bool off = getenv("PGH_DROPOUT_CSR") != nullptr && atoi(getenv("PGH_DROPOUT_CSR")) != 0;      // 1: always the row-major kernel
    return !off && g->bsf.enabled && g->bsf.pb_slices <= 1 && g->rowptr != nullptr && g->col != nullptr && g->nnz > 0 &&
           (g->bsf.val != nullptr || g->keep_mult != nullptr);
}
// The index in CSR(M^T) order of every entry of the stream and of the cold image (the words the dropout kernels read), built on the
// first dropout launch: the image is built a SECOND time with the entry's index in the place of its value -- same keys, same
// sort, same hot / cold split, same padding, so the value arrays of that shadow image ARE the index words in the layout of the
// kernels (a copy of a multigraph entry carries the index of its entry: one mask bit per entry of the matrix) -- and everything
// else of the shadow is released again.
int bsf_ensure_edge_ids(pgh_graph_s* g) {
    BsfFormat& f = g->bsf;
    if (f.drop_edge != nullptr) return 0;
    PGH_CHECK(bsf_dropout_usable(g), "graph_dropout: this graph has no blocked route");
    Runtime& r = rt();
    DevBuf<float> ids;
    PGH_TRY(ids.alloc(g->nnz));
    k_index_bits<<<blocks_for(g->nnz), kBlock, 0, r.stream>>>(ids.p, g->nnz);
    PGH_HIP(hipGetLastError());
    const BsfFormat saved = g->bsf;
    g->bsf = BsfFormat();
    const int rc = bsf_build(g, ids.p, saved.val == nullptr ? g->keep_mult : nullptr, nullptr, nullptr, saved.relabelled, saved.num_blocks);
    BsfFormat shadow = g->bsf;
    g->bsf = saved;
    if (rc != 0) {
        const std::string keep = pgh_last_error();
        bsf_destroy(shadow);
        return fail(keep);
    }
    const bool same = shadow.num_tiles == f.num_tiles && shadow.num_entries == f.num_entries && shadow.num_blocks == f.num_blocks &&
                      shadow.pb.enabled == f.pb.enabled && shadow.pb.num_entries == f.pb.num_entries && shadow.pb.num_tasks == f.pb.num_tasks &&
                      (shadow.colf16 != nullptr) == (f.colf16 != nullptr) && shadow.val != nullptr && (!f.pb.enabled || shadow.pb.val != nullptr);
    if (!same) {
        bsf_destroy(shadow);
        return fail("graph_dropout: the index image does not match the graph's blocked image");
    }
    f.drop_edge = reinterpret_cast<int32_t*>(shadow.val);
    shadow.val = nullptr;
    f.device_bytes += f.num_entries * 4;
    if (f.pb.enabled) {
        f.pb.drop_edge = reinterpret_cast<int32_t*>(shadow.pb.val);
        shadow.pb.val = nullptr;
        f.device_bytes += f.pb.num_entries * 4;
    }
    bsf_destroy(shadow);
    return 0;
}

// output-side vector: original -> internal
// row sums of M (pgh_graph_s::degrees, caller ids) in the internal id space of a square relabelled graph
int bsf_ensure_degrees(pgh_graph_s* g) {
    BsfFormat& f = g->bsf;
    if (f.deg_int != nullptr) return 0;
    PGH_CHECK(g->n_rows == g->n_cols && g->degrees != nullptr, "internal-space degrees need a square graph");
    float* d = nullptr;
    PGH_HIP(pooled_malloc(&d, sizeof(float) * (size_t)(f.n_out > 0 ? f.n_out : 1)));
    const int rc = bsf_out_to_internal(g, g->degrees, d, 0.f);
    if (rc != 0) {
        (void)pooled_free(d);
        return rc;
    }
    f.deg_int = d;
    f.device_bytes += (int64_t)f.n_out * 4;
    return 0;
}

int bsf_out_to_internal(pgh_graph_s* g, const float* src, float* dst, float hole) {
    BsfFormat& f = g->bsf;
    k_permute_in<<<blocks_for(f.n_out), kBlock, 0, rt().stream>>>(src, f.relabelled ? f.perm : nullptr, nullptr, f.n_out,
                                                                 f.n_out_orig, hole, dst);
    PGH_HIP(hipGetLastError());
    return 0;
}

// output-side vector: internal -> original
int bsf_to_original(pgh_graph_s* g, const float* src, float* dst, double factor) {
    BsfFormat& f = g->bsf;
    if (f.relabelled && f.iperm != nullptr) {          // gather through the inverse map: coalesced stores (the scatter form
        int64_t chunks = ((int64_t)f.n_out_orig + PGH_OUT_CHUNK - 1) / PGH_OUT_CHUNK;
        const int64_t cap = (int64_t)rt().num_cus * 8;
        if (chunks > cap) chunks = cap;
        if (chunks < 1) chunks = 1;
        k_permute_out_gather<<<(int)chunks, kBlock, 0, rt().stream>>>(src, f.iperm, f.n_out_orig, (float)factor, dst, iso_tail_of(f));   // (the scatter form
                                                                                                                                      // writes 4 bytes per line)
        PGH_HIP(hipGetLastError());
        return 0;
    }
    k_permute_out<<<blocks_for(f.n_out), kBlock, 0, rt().stream>>>(src, f.relabelled ? f.perm : nullptr, f.n_out, f.n_out_orig,
                                                                  (float)factor, dst);
    PGH_HIP(hipGetLastError());
    return 0;
}

void bsf_destroy(BsfFormat& f) {
    for (int sl = 1; sl < kPbMaxSlices; ++sl) pb_destroy(f.pb_more[sl - 1]);
    pb_destroy(f.pb);
    (void)pooled_free(f.colf);
    (void)pooled_free(f.colf16);
    (void)pooled_free(f.iperm);
    (void)pooled_free(f.flags8);
    (void)pooled_free(f.deg_int);
    (void)pooled_free(f.fix_seg);
    (void)pooled_free(f.psum);
    (void)pooled_free(f.psum64);
    (void)pooled_free(f.drop_edge);
    (void)pooled_free(f.meta);
    (void)pooled_free(f.live_dev);
    (void)pooled_free(f.val);
    (void)pooled_free(f.seg_row);
    (void)pooled_free(f.tile);
    (void)pooled_free(f.tail_carry);
    (void)pooled_free(f.head_partial);
    (void)pooled_free(f.part);
    (void)pooled_free(f.mm_close);
    (void)pooled_free(f.mm_edge);
    (void)pooled_free(f.iso_flag);
    (void)pooled_free(f.seed_list);
    (void)pooled_free(f.seed_count);
    (void)pooled_free(f.mm_row_has);
    (void)pooled_free(f.mm_rowop);
    (void)pooled_free(f.need_idx);
    (void)pooled_free(f.send_rows);
    (void)pooled_free(f.perm);
    (void)pooled_free(f.src_scale);
    (void)pooled_free(f.dst_scale);
    (void)pooled_free(f.xg);
    (void)pooled_free(f.tmp_out);
    f = BsfFormat();
}

// Relabelling shared by the single-GPU layout and the row-partitioned generator: ids sorted by descending reference
// count (stable: ties keep ascending id), rank r -> new id deal_new_id(r, B, blk, head): dealt to B contiguous hot-first blocks,
// the first `head` ranks one by one, the rest in runs of kDealRun.  perm[new] = old (pad slots -1, perm has B * blk entries), iperm[old] = new.
int build_count_perm(const unsigned int* cnt, int64_t n, int B, int blk, int32_t* perm, int32_t* iperm, int64_t head) {
    Runtime& r = rt();
    PGH_CHECK(head >= n || B == 1 || (head % ((int64_t)B * kDealRun) == 0 && blk % kDealRun == 0 &&
                                      (n + (int64_t)kDealRun * B - 1) / ((int64_t)kDealRun * B) * kDealRun <= blk),
              "relabelling: a column block must hold whole runs of the deal (kDealRun ids) and its share of the ranks");
    DevBuf<unsigned int> cnt_sorted;
    DevBuf<int32_t> ids, ids_sorted;
    PGH_TRY(cnt_sorted.alloc(n));
    PGH_TRY(ids.alloc(n));
    PGH_TRY(ids_sorted.alloc(n));
    k_iota<<<blocks_for(n), kBlock, 0, r.stream>>>(ids.p, n);
    size_t temp_bytes = 0;
    PGH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp_bytes, cnt, cnt_sorted.p, ids.p, ids_sorted.p, (int)n, 0, 32, r.stream));
    DevBuf<char> temp;
    PGH_TRY(temp.alloc(temp_bytes));
    PGH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(temp.p, temp_bytes, cnt, cnt_sorted.p, ids.p, ids_sorted.p, (int)n, 0, 32, r.stream));
    k_fill_perm_pad<<<blocks_for((int64_t)B * blk), kBlock, 0, r.stream>>>(perm, (int64_t)B * blk);
    k_make_perm<<<blocks_for(n), kBlock, 0, r.stream>>>(ids_sorted.p, n, B, blk, head, perm, iperm);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    return 0;
}

// number of column blocks: keep a block's slice of the gather vector around 8 MB (hot-first, so its reused prefix
// fits a 4 MB L2) up to 4 blocks; EIGHT from 16 MB of gather vector on (scale 22).  Every block is one more LDS hot cache
// (29 696 more sources out of the cold image) and one more row segment per row.  Rounds 1-2 measured 4 against 8 blocks with
// dense [B][n] partial vectors and separate fix-up / combine launches and kept 4 (profiles/r01/pb_large_graphs.log,
// profiles/r02/blocks_sweep_scale23.log); with compact partial sums and the fused finish kernel the balance has turned -- round 3,
// one PPR iteration over all launches at scale 22 / 23 / 24 / 25, 4 -> 8 blocks: 136 -> 132, 245 -> 237, 544 -> 523, 1513 -> 1165 us
// (phase A loses a quarter of its entries, the block partial sums gain 8 us; profiles/r03/blocks_sweep.log).
// Override with PGH_BLOCKS for experiments.
int bsf_auto_blocks(int64_t n_src) {
    int B = 1;
    while (B < 4 && n_src * 4 > (int64_t)B * (8 << 20)) B <<= 1;
    if (n_src * 4 > (int64_t)(16 << 20)) B = 8;
    const int forced = env_int("PGH_BLOCKS", 0);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) B = forced;
    return B;
}

// Build the blocked format from CSR(M^T) already on the device.
//   val  != null : generic weighted matrix (8 B/edge)
//   mult != null : value-free; M^T = diag(dst_old) * mult * diag(src_old) (either scale may be null = 1)
// relabel: permute ids by descending source count (square graphs only).
int bsf_build(pgh_graph_s* g, const float* val, const int32_t* mult, const float* src_old, const float* dst_old, bool relabel,
              int force_blocks, BsfFormat* target) {
    Runtime& r = rt();
    BsfFormat& f = target ? *target : g->bsf;
    const bool batch_layout = target != nullptr && target != &g->bsf;   // multi-seed layout: no single-vector work buffers
    const int64_t n_src = g->n_rows, n_out = g->n_cols, nnz = g->nnz;
    PGH_CHECK(n_src < (1LL << 28) && n_out < (1LL << 28), "blocked format needs fewer than 2^28 rows/columns");
    if (relabel && n_src != n_out) relabel = false;
    const int B = force_blocks > 0 ? force_blocks : bsf_auto_blocks(n_src);
    PGH_CHECK(B >= 1 && B <= kMaxBlocks && (B <= 8 || (target != nullptr && target != &g->bsf)), "blocked format: unsupported number of column blocks");
    // (whole runs of the deal per block: deal_new_id)
    const int blk = B > 1 ? (int)((n_src + (int64_t)kDealRun * B - 1) / ((int64_t)kDealRun * B)) * kDealRun : (int)n_src;
    const int n_src_pad = B * blk;
    f.num_blocks = B;
    f.blk_size = blk;
    // slices of a row partition: need lists (PGH_DIST_NEED_LISTS=0: the dense cold layout of rounds 1-4, exchanged by all-gather alone)
    f.want_compact = !batch_layout && g->part_perm != nullptr && env_int("PGH_DIST_NEED_LISTS", 1) != 0;
    f.whole_graph = g->part_perm == nullptr;
    f.n_src = (int)n_src;
    f.n_src_pad = n_src_pad;
    f.relabelled = relabel;
    f.n_out_orig = (int)n_out;
    // square graphs: outputs live in the padded internal id space [0, B * blk) (relabelled ids are spread over it,
    // and an un-relabelled output vector doubles as the next gather vector, whose hot prefix is read per block)
    f.n_out = (n_src == n_out) ? n_src_pad : (int)n_out;

    DevBuf<int32_t> iperm;
    if (relabel) {
        DevBuf<unsigned int> cnt;
        PGH_TRY(cnt.alloc(n_src, true));
        PGH_TRY(iperm.alloc(n_src));
        PGH_HIP(pooled_malloc(&f.perm, sizeof(int32_t) * (size_t)n_src_pad));
        // the reference count of every source: the caller's (the generator's out-degrees, an upload's row sums), the one an earlier image
        // of this graph counted, or counted now and kept
        const bool weighted = mult != nullptr;
        if (g->src_counts != nullptr && g->src_counts_weighted == weighted) {
            PGH_HIP(hipMemcpyAsync(cnt.p, g->src_counts, sizeof(unsigned int) * (size_t)n_src, hipMemcpyDeviceToDevice, r.stream));
        } else {
            if (nnz > 0) k_source_counts<<<blocks_for(nnz), kBlock, 0, r.stream>>>(g->col, mult, nnz, cnt.p);
            if (g->src_counts == nullptr && n_src > 0 && pooled_malloc(&g->src_counts, sizeof(unsigned int) * (size_t)n_src) == hipSuccess) {
                PGH_HIP(hipMemcpyAsync(g->src_counts, cnt.p, sizeof(unsigned int) * (size_t)n_src, hipMemcpyDeviceToDevice, r.stream));
                g->src_counts_weighted = weighted;
                g->device_bytes += (int64_t)n_src * 4;
            }
        }
        build_mark("image: source counts");
        // isolated nodes (never referenced, empty row) sort last: the rows [iso_begin[b], blk) of every block b hold nothing
        // and are referenced by nothing -- they change only through the personalization (k_pb_finish, k_step_residual)
        DevBuf<unsigned int> key, live_count;
        PGH_TRY(key.alloc(n_src));
        PGH_TRY(live_count.alloc(1, true));
        k_relabel_keys<<<blocks_for(n_src), kBlock, 0, r.stream>>>(cnt.p, g->rowptr, n_src, key.p, live_count.p);
        const bool iso_on = env_int("PGH_ISO", 1) != 0;          // 0: the round-1 order (ties by id), no isolated tail (diagnostic)
        // the hottest ranks of every block one by one (their entries decide how evenly the blocks / XCDs are loaded, and their ROWS are
        // the heavy ones: kept apart in the row order), the tail in runs.  PGH_DEAL_HEAD: slots per block dealt one by one (a multiple of
        // 32).  Same box, scale 23 (profiles/r04/default_rule.log): head 4096 / 29 696 / 131 072 / everything one by one -- the bench
        // 563 / 570 / 571 / 568 GTEPS (block partial sums 69.0 / 67.1 / 66.6 / 67.0 us), the 2-iteration default-rule run 364 / 360 /
        // 360 / 341 GTEPS: runs among the hot rows cost the step kernels what they save on the way out.
        const int head_slots = env_int("PGH_DEAL_HEAD", 131072) / kDealRun * kDealRun;
        f.deal_head = (B > 1 && env_int("PGH_DEAL_RUNS", 1) != 0) ? (int64_t)B * head_slots : kDealHeadAll;
        PGH_TRY(build_count_perm(iso_on ? key.p : cnt.p, n_src, B, blk, f.perm, iperm.p, f.deal_head));
        unsigned int live_nodes = 0;
        PGH_HIP(hipMemcpyAsync(&live_nodes, live_count.p, sizeof(unsigned int), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        f.live_nodes = iso_on ? (int64_t)live_nodes : -1;
        build_mark("image: relabelling (sort by count)");
        for (int b = 0; b < B && b < 8; ++b) {
            // the first slot of block b whose rank is isolated (ranks ascend with the slot: deal_rank_of), rounded up to whole float4s
            const int64_t first_iso = deal_first_slot((int64_t)live_nodes, b, B, blk, f.deal_head);
            f.iso_begin[b] = (int)std::min<int64_t>(blk, (std::max<int64_t>(first_iso, 0) + 3) & ~(int64_t)3);
        }
        f.has_iso = iso_on && B <= 8;
        f.iso_row_blocks = B;
    } else if (g->part_perm != nullptr && g->part_live_nodes >= 0 && n_out > 0 && n_out % blk == 0 && g->row_begin % blk == 0 &&
               (target == nullptr || target == &g->bsf)) {
        // a rank's slice of a generated partition: its rows are the global blocks row_begin / blk .. (the generator broke the
        // relabelling's ties the same way: pgh_graphgen.hip), so the tail of every one of its row blocks is isolated ids too
        const int first_block = (int)(g->row_begin / blk), row_blocks = (int)(n_out / blk);
        if (row_blocks >= 1 && row_blocks <= 8 && first_block + row_blocks <= B) {
            for (int j = 0; j < 8; ++j) f.iso_begin[j] = blk;
            for (int j = 0; j < row_blocks; ++j) {
                const int64_t first_iso = deal_first_slot(g->part_live_nodes, first_block + j, B, blk, kDealHeadAll);     // partitions: one by one
                f.iso_begin[j] = (int)std::min<int64_t>(blk, (std::max<int64_t>(first_iso, 0) + 3) & ~(int64_t)3);
            }
            f.has_iso = true;
            f.iso_row_blocks = row_blocks;
        }
    }
    // ---- entry expansion offsets (value-free: multiplicities become repeated entries)
    DevBuf<int64_t> offs;
    int64_t E = nnz;
    if (mult && nnz > 0) {
        PGH_TRY(offs.alloc(nnz));
        size_t temp_bytes = 0;
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, mult, offs.p, (int)nnz, r.stream));
        DevBuf<char> temp;
        PGH_TRY(temp.alloc(temp_bytes));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(temp.p, temp_bytes, mult, offs.p, (int)nnz, r.stream));
        int64_t last_off = 0;
        int32_t last_m = 0;
        PGH_HIP(hipMemcpyAsync(&last_off, offs.p + nnz - 1, sizeof(int64_t), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipMemcpyAsync(&last_m, mult + nnz - 1, sizeof(int32_t), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        E = last_off + last_m;
    }
    build_mark("image: expansion offsets");
    E += B;                                            // one sentinel per block
    PGH_CHECK(E < 2147483647LL, "blocked format: more than 2^31 entries in one graph; partition it (SURVEY.md 8e)");
    DevBuf<uint64_t> keys_a, keys_b;
    DevBuf<float> vals_a;
    DevBuf<int> flags, segid;
    PGH_TRY(keys_a.alloc(E));
    PGH_TRY(keys_b.alloc(E));
    if (val) {
        PGH_TRY(vals_a.alloc(E, true));
        PGH_HIP(pooled_malloc(&f.val, sizeof(float) * (size_t)E));
    }
    if (nnz > 0) {
        if (env_int("PGH_KEYS_BY_ROW", 0) != 0) {          // the round 1-5 expansion (A/B measurements)
            k_bsf_keys<<<blocks_for(n_out * 64), kBlock, 0, r.stream>>>(g->rowptr, g->col, val, mult, offs.p, n_out,
                                                                        relabel ? iperm.p : nullptr, relabel ? iperm.p : nullptr, blk,
                                                                        keys_a.p, val ? vals_a.p : nullptr);
        } else {
            // keys_b is free until the sort: its first nnz words hold the row of every entry
            int32_t* row_of = reinterpret_cast<int32_t*>(keys_b.p);
            int32_t* row_scan = row_of + nnz;              // (2 nnz words of 4 bytes fit the E >= nnz + 1 words of 8)
            PGH_HIP(hipMemsetAsync(row_of, 0, sizeof(int32_t) * (size_t)nnz, r.stream));
            k_row_heads<<<blocks_for(n_out), kBlock, 0, r.stream>>>(g->rowptr, n_out, row_of);
            size_t temp_bytes = 0;
            PGH_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, temp_bytes, row_of, row_scan, hipcub::Max(), (int)nnz, r.stream));
            DevBuf<char> temp;
            PGH_TRY(temp.alloc(temp_bytes));
            PGH_HIP(hipcub::DeviceScan::InclusiveScan(temp.p, temp_bytes, row_of, row_scan, hipcub::Max(), (int)nnz, r.stream));
            k_bsf_keys_flat<<<blocks_for(nnz), kBlock, 0, r.stream>>>(row_scan, g->col, val, mult, offs.p, nnz, relabel ? iperm.p : nullptr,
                                                                      relabel ? iperm.p : nullptr, blk, keys_a.p, val ? vals_a.p : nullptr);
            PGH_HIP(hipStreamSynchronize(r.stream));       // (temp goes out of scope)
        }
    }
    k_bsf_sentinels<<<1, 64, 0, r.stream>>>(keys_a.p, E - B, B, n_src_pad);
    PGH_HIP(hipGetLastError());
    build_mark("image: entry keys");
    {
        const int sort_bits = B > 8 ? 64 : 61;             // block id from bit 58 up
        size_t temp_bytes = 0;
        if (val) PGH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_a.p, keys_b.p, vals_a.p, f.val, (int)E, 0, sort_bits, r.stream));
        else PGH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys_a.p, keys_b.p, (int)E, 0, sort_bits, r.stream));
        DevBuf<char> temp;
        PGH_TRY(temp.alloc(temp_bytes));
        if (val) PGH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.p, keys_b.p, vals_a.p, f.val, (int)E, 0, sort_bits, r.stream));
        else PGH_HIP(hipcub::DeviceRadixSort::SortKeys(temp.p, temp_bytes, keys_a.p, keys_b.p, (int)E, 0, sort_bits, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    build_mark("image: entry sort");
    // ---- referenced prefix of every block (over ALL entries), then the cold tail moves to its own image (pgh_pb.hip)
    int live_all[kMaxBlocks] = {0};
    {
        DevBuf<int32_t> d_live;
        PGH_TRY(d_live.alloc(kMaxBlocks, true));
        k_bsf_live<<<blocks_for(E), kBlock, 0, r.stream>>>(keys_b.p, E, blk, d_live.p);
        PGH_HIP(hipMemcpyAsync(live_all, d_live.p, sizeof(live_all), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    build_mark("image: referenced prefixes");
    // (the f64 image -- a multi-seed-style layout -- asks for a cold image of its own with ITS hot cache's size: BsfFormat::pb64)
    const int hot_want = batch_layout && f.pb64 && f.pb_hot > 0 ? f.pb_hot : kBsfHot;
    const int hot_slots = hot_want < blk ? hot_want : blk;
    if ((!batch_layout || (f.pb64 && B <= 8)) && hot_slots < blk && E > B) {
        DevBuf<unsigned char> is_hot;
        DevBuf<int64_t> num_hot;
        PGH_TRY(is_hot.alloc(E));
        PGH_TRY(num_hot.alloc(1));
        k_bsf_is_hot<<<blocks_for(E), kBlock, 0, r.stream>>>(keys_b.p, E, blk, hot_slots, is_hot.p);
        PbPlan plan;
        bool use_pb = false;
        PGH_TRY(pb_plan(f, keys_b.p, E, live_all, hot_slots, is_hot.p, &plan, &use_pb));   // may keep heavy rows in the stream
        build_mark("cold image: plan");
        if (use_pb) {
            size_t temp_bytes = 0;
            PGH_HIP(hipcub::DevicePartition::Flagged(nullptr, temp_bytes, keys_b.p, is_hot.p, keys_a.p, num_hot.p, (int)E, r.stream));
            DevBuf<char> temp;
            PGH_TRY(temp.alloc(temp_bytes));
            // stream entries first, in order; the image's entries behind them (reversed; the image sorts them again)
            PGH_HIP(hipcub::DevicePartition::Flagged(temp.p, temp_bytes, keys_b.p, is_hot.p, keys_a.p, num_hot.p, (int)E, r.stream));
            int64_t E_hot = 0;
            PGH_HIP(hipMemcpyAsync(&E_hot, num_hot.p, sizeof(int64_t), hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
            if (val) {
                PGH_HIP(hipcub::DevicePartition::Flagged(temp.p, temp_bytes, f.val, is_hot.p, vals_a.p, num_hot.p, (int)E, r.stream));
                PGH_HIP(hipStreamSynchronize(r.stream));
            }
            build_mark("cold image: hot / cold split");
            int rc_pb = 0;
            {
                // the image is built slice by slice: the slice's entries are compacted out of the cold keys first
                DevBuf<uint64_t> slice_keys;
                DevBuf<float> slice_vals;
                const int64_t cold_count = E - E_hot;
                rc_pb = slice_keys.alloc(cold_count);
                if (rc_pb == 0 && val) rc_pb = slice_vals.alloc(cold_count);
                for (int sl = 0; rc_pb == 0 && sl < plan.slices; ++sl) {
                    int64_t got = 0;
                    rc_pb = pb_select_slice(&plan, sl, keys_a.p + E_hot, val ? vals_a.p + E_hot : nullptr, cold_count, slice_keys.p,
                                            val ? slice_vals.p : nullptr, &got);
                    if (rc_pb == 0) rc_pb = pb_build(f, &plan, sl, slice_keys.p, val ? slice_vals.p : nullptr, got, live_all, hot_slots);
                }
                if (rc_pb == 0) f.pb_slices = plan.slices;
            }
            pb_plan_release(&plan);
            PGH_TRY(rc_pb);
            build_mark("cold image: build (schedules)");
            std::swap(keys_a.p, keys_b.p);                 // keys_b: the stream's entries, still sorted by (block, row, col)
            if (val) std::swap(f.val, vals_a.p);
            E = E_hot;
        }
    }
    // ---- pad every block to whole tiles: every wavefront tile is full, 16-byte aligned and never straddles blocks.
    //      Pad entries repeat the block's sentinel key: no flag, they only feed the sentinel's never-closed segment.
    TileBuild tb;
    tb.B = B;
    int64_t h[kMaxBlocks + 1] = {0};
    {
        DevBuf<int64_t> starts;
        PGH_TRY(starts.alloc(kMaxBlocks + 1));
        k_bsf_block_starts<<<1, 128, 0, r.stream>>>(keys_b.p, E, B, starts.p);
        PGH_HIP(hipMemcpyAsync(h, starts.p, sizeof(int64_t) * (B + 1), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        int tiles = 0;
        for (int b = 0; b <= kMaxBlocks; ++b) {
            tb.tile_begin[b] = tiles;
            f.tile_begin[b] = tiles;
            tb.block_start[b] = (int64_t)tiles * kTile;
            if (b < B) tiles += (int)((h[b + 1] - h[b] + kTile - 1) / kTile);
        }
        f.num_tiles = tiles;
    }
    const int64_t EP = (int64_t)f.num_tiles * kTile;
    PGH_CHECK(EP < 2147483647LL, "blocked format: more than 2^31 padded entries in one graph; partition it (SURVEY.md 8e)");
    f.num_entries = EP;
    keys_a.~DevBuf<uint64_t>();
    new (&keys_a) DevBuf<uint64_t>();
    DevBuf<uint64_t> keys_p;
    PGH_TRY(keys_p.alloc(EP));
    float* val_sorted = f.val;
    if (val) PGH_HIP(pooled_malloc(&f.val, sizeof(float) * (size_t)EP));
    {
        PadLayout pl;
        pl.B = B;
        for (int b = 0; b <= kMaxBlocks; ++b) {
            pl.src_start[b] = b <= B ? h[b] : E;
            pl.dst_start[b] = tb.block_start[b];
        }
        k_bsf_pad<<<blocks_for(EP), kBlock, 0, r.stream>>>(pl, keys_b.p, val ? val_sorted : nullptr, EP, blk, keys_p.p, val ? f.val : nullptr);
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    if (val) (void)pooled_free(val_sorted);
    build_mark("stream: tiles and padding");
    keys_b.~DevBuf<uint64_t>();
    new (&keys_b) DevBuf<uint64_t>();
    PGH_HIP(pooled_malloc(&f.colf, sizeof(uint32_t) * (size_t)EP));
    PGH_TRY(flags.alloc(EP));
    PGH_TRY(segid.alloc(EP));
    k_bsf_flags<<<blocks_for(EP), kBlock, 0, r.stream>>>(keys_p.p, EP, f.colf, flags.p);
    {
        size_t temp_bytes = 0;
        PGH_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, temp_bytes, flags.p, segid.p, (int)EP, r.stream));
        DevBuf<char> temp;
        PGH_TRY(temp.alloc(temp_bytes));
        PGH_HIP(hipcub::DeviceScan::InclusiveSum(temp.p, temp_bytes, flags.p, segid.p, (int)EP, r.stream));
        int nseg = 0;
        PGH_HIP(hipMemcpyAsync(&nseg, segid.p + EP - 1, sizeof(int), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        f.num_segs = nseg;
    }
    // + 192: k_bsf_partial prefetches the rows of 128 segments per tile unconditionally (reads past a tile's own segments
    // are never used, but they must stay inside the allocation: found by tools/stress_gpu.py as a rare memory fault)
    PGH_HIP(pooled_malloc(&f.seg_row, sizeof(int32_t) * (size_t)(f.num_segs + 1)));
    PGH_HIP(hipMemsetAsync(f.seg_row, 0xff, sizeof(int32_t) * (size_t)(f.num_segs + 1), r.stream));
    if (!batch_layout || f.want_meta) {                // SpMV layouts: the row -> segment map of the compact partial sums
        f.meta_words = ((int64_t)(f.n_out > 0 ? f.n_out : 1) + 63) / 64;
        PGH_HIP(pooled_malloc(&f.meta, sizeof(SegMeta) * (size_t)(f.meta_words * B)));
        k_bsf_meta_init<<<blocks_for(f.meta_words * B), kBlock, 0, r.stream>>>(f.meta, f.meta_words * B);
    }
    k_bsf_seg_rows<<<blocks_for(EP), kBlock, 0, r.stream>>>(keys_p.p, segid.p, EP, f.seg_row, f.meta, f.meta_words);
    PGH_HIP(pooled_malloc(&f.tile, sizeof(int4) * (size_t)(f.num_tiles + 1)));
    PGH_HIP(pooled_malloc(&f.tail_carry, sizeof(double) * (size_t)(f.num_tiles + 1)));
    PGH_HIP(pooled_malloc(&f.head_partial, sizeof(double) * (size_t)(f.num_tiles + 1)));
    PGH_HIP(hipMemsetAsync(f.tail_carry, 0, sizeof(double) * (size_t)(f.num_tiles + 1), r.stream));
    PGH_HIP(hipMemsetAsync(f.head_partial, 0, sizeof(double) * (size_t)(f.num_tiles + 1), r.stream));
    k_bsf_tiles<<<blocks_for(f.num_tiles), kBlock, 0, r.stream>>>(tb, segid.p, f.tile);
    build_mark("stream: flags, segments, tile table");
    // ---- work buffers and scales
    const size_t n_int = (size_t)(f.n_out > 0 ? f.n_out : 1);
    if (!batch_layout) {
        // + one tile of slack: the predicated stores of k_bsf_partial address up to 512 slots past a tile's first segment
        PGH_HIP(pooled_malloc(&f.psum, sizeof(float) * (size_t)(f.num_segs + kTile + 64)));
        PGH_HIP(hipMemsetAsync(f.psum, 0, sizeof(float) * (size_t)(f.num_segs + kTile + 64), r.stream));
        PGH_HIP(pooled_malloc(&f.xg, sizeof(float) * (size_t)(n_src_pad + 1)));
        PGH_HIP(hipMemsetAsync(f.xg, 0, sizeof(float) * (size_t)(n_src_pad + 1), r.stream));
        PGH_HIP(pooled_malloc(&f.tmp_out, sizeof(float) * n_int));
    }
    if (src_old) {
        PGH_HIP(pooled_malloc(&f.src_scale, sizeof(float) * (size_t)(n_src_pad + 1)));
        PGH_HIP(hipMemsetAsync(f.src_scale, 0, sizeof(float) * (size_t)(n_src_pad + 1), r.stream));
        k_permute_in<<<blocks_for(n_src_pad), kBlock, 0, r.stream>>>(src_old, f.perm, nullptr, n_src_pad, n_src, 0.f, f.src_scale);
    }
    if (dst_old) {
        PGH_HIP(pooled_malloc(&f.dst_scale, sizeof(float) * n_int));
        // rows use the same relabelling as sources on square graphs; rectangular slices are not relabelled
        k_permute_in<<<blocks_for(f.n_out), kBlock, 0, r.stream>>>(dst_old, relabel ? f.perm : nullptr, nullptr, f.n_out, n_out, 0.f, f.dst_scale);
    }
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    f.device_bytes = (int64_t)E * (val ? 8 : 4) + f.num_segs * 4 + (int64_t)f.num_tiles * 32 + f.meta_words * B * (int64_t)sizeof(SegMeta) +
                     (int64_t)n_src_pad * (4 + (src_old ? 4 : 0) + (relabel ? 4 : 0)) + (int64_t)n_out * (dst_old ? 8 : 4);
    if (f.has_iso && !batch_layout) {                  // "process every row" until a recursive loop watches its operands
        PGH_HIP(pooled_malloc(&f.iso_flag, sizeof(int)));
        PGH_HIP(hipMemsetAsync(f.iso_flag, 0xff, sizeof(int), r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    if (relabel && !batch_layout) {                    // old id -> new id: results are brought back by a gather
        f.iperm = iperm.release();
        f.device_bytes += (int64_t)n_src * 4;
    }
    if (env_int("PGH_DEBUG", 0)) {
        DevBuf<unsigned long long> cnt;
        PGH_TRY(cnt.alloc(1, true));
        k_bsf_count_hot<<<blocks_for(EP), kBlock, 0, r.stream>>>(f.colf, EP, blk, kBsfHot < blk ? kBsfHot : blk, cnt.p);
        unsigned long long hcount = 0;
        PGH_HIP(hipMemcpyAsync(&hcount, cnt.p, sizeof(hcount), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        fprintf(stderr, "[pgh] bsf: tiles per block:");
        for (int b = 0; b < B; ++b) fprintf(stderr, " %d", f.tile_begin[b + 1] - f.tile_begin[b]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[pgh] bsf: B=%d blk=%d entries=%lld (padded %lld) segs=%lld tiles=%d hot=%d covers %.1f%% of entries, value-free=%d relabel=%d\n",
                B, blk, (long long)E, (long long)EP, (long long)f.num_segs, f.num_tiles, kBsfHot, 100.0 * (double)hcount / (double)EP,
                val ? 0 : 1, relabel ? 1 : 0);
    }
    if (!batch_layout) {
        static_assert(kIPT == 8, "k_bsf_pack digests 8 entries per lane");
        PGH_HIP(pooled_malloc(&f.flags8, (size_t)f.num_tiles * 64 + 64));
        PackLayout pl;
        pl.num_blocks = B;
        for (int b = 0; b <= 8; ++b) pl.tile_begin[b] = f.tile_begin[b];
        PGH_HIP(pooled_malloc(&f.live_dev, sizeof(int32_t) * 8));
        PGH_HIP(hipMemsetAsync(f.live_dev, 0, sizeof(int32_t) * 8, r.stream));
        k_bsf_pack<<<blocks_for((int64_t)f.num_tiles * 64), kBlock, 0, r.stream>>>(f.colf, reinterpret_cast<uint32_t*>(f.val), pl, f.num_tiles, blk, f.flags8,
                                                                                 f.live_dev);
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipMemcpyAsync(f.live, f.live_dev, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        for (int b = 0; b < 8; ++b) f.live[b] = live_all[b] > f.live[b] ? live_all[b] : f.live[b];   // the cold image's sources count too
        for (int b = 0; b < 8; ++b) f.xg_base[b] = f.xg_base_cold[b] = (int64_t)b * blk;
        PGH_HIP(pooled_malloc(&f.fix_seg, sizeof(int32_t) * (size_t)(f.num_tiles + 1)));
        k_bsf_fixlist<<<blocks_for(f.num_tiles), kBlock, 0, r.stream>>>(f.tile, f.seg_row, f.num_tiles, f.fix_seg);
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipStreamSynchronize(r.stream));
        (void)pooled_free(f.seg_row);                       // build-time only in this layout
        f.seg_row = nullptr;

        if (f.pb.enabled && !f.pb.k1_cold && env_int("PGH_STREAM16", 1)) {       // hot-only stream: 2 bytes per entry
            const uint32_t hot4 = (uint32_t)(kBsfHot < blk ? kBsfHot : blk) << 2;
            PGH_HIP(pooled_malloc(&f.colf16, sizeof(uint16_t) * (size_t)f.num_tiles * 512 + 64));
            k_bsf_narrow<<<blocks_for((int64_t)f.num_tiles * 64), kBlock, 0, r.stream>>>(f.colf, f.num_tiles, hot4, f.colf16);
            PGH_HIP(hipGetLastError());
            PGH_HIP(hipStreamSynchronize(r.stream));
            (void)pooled_free(f.colf);
            f.colf = nullptr;
            f.device_bytes -= (int64_t)f.num_tiles * 512 * 2;
        }
        // The engine's own gather vector (graphs with a source scale: xg = y * src_scale, written by the epilogue) keeps
        // only the referenced prefix of every block.  Partitioned graphs get their layout from the caller instead.
        if (src_old != nullptr && g->part_perm == nullptr && env_int("PGH_TRIM", 1)) {
            int top = 1;
            for (int b = 0; b < B; ++b) top = f.live[b] > top ? f.live[b] : top;
            top = (top + 63) / 64 * 64;
            if (top < blk) {
                f.xg_live = top;
                for (int b = 0; b < 8; ++b) f.xg_base[b] = f.xg_base_cold[b] = (int64_t)b * top;
            }
        }
        f.device_bytes += (int64_t)f.num_tiles * 64;
    }
    f.enabled = true;
    build_mark("stream: packing, buffers, scales");
    return 0;
}

}  // namespace pgh

PGH_WARM_KERNEL(k_row_heads)
