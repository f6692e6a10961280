// Propagation blocking for the cold tail of the blocked stream.
//
// k_bsf_partial serves the gathers whose source sits in the block's LDS hot cache; every other ("cold") gather is one
// L2 request for 4 useful bytes, and on the bench graph the L2 request rate -- not HBM -- bounds the step (DESIGN.md
// section 4).  This image removes them: the cold entries are taken out of the stream and processed by two streaming
// passes whose random accesses all land in LDS.
//
//   bins     runs of consecutive output rows: <= 4096 rows (their sums live in LDS during phase B; 16384 when the graph
//            has so many chunks that the runs below would get short) and <= 6 cold entries per row on average (balance);
//            a row with more than 16 K cold entries gets "hub" bins of its own, one per 64 K entries (summed in f64
//            registers instead of LDS atomics on one word; the pieces of a split row are folded in a fixed order)
//   cells    (source chunk, bin) pairs; the entries of a cell form a RUN, padded to a multiple of 8 entries.  The same
//            runs are laid out twice: A order = [chunk][bin] (what phase A reads), B order = [bin][chunk] (what phase B
//            reads); an 8-entry group of one order maps to one 8-entry group of the other.
//   phase A  k_pb_gather      every workgroup takes a share of the A-order stream: the chunk's slice of the gather vector
//                             goes to LDS (128 KB, coalesced), then per group of 8 entries: 16 B of 2-byte source indices
//                             + 4 B target group read, 8 LDS gathers, 32 B written to the group's place in B order.
//                             Writes land in runs of cells: ~2-3 KB each on the bench graph.
//   phase B  k_pb_finish      one work item per bin AND the filter's epilogue for the bin's rows in the same launch (round 1
//                             ran k_pb_accumulate -> dense cold vector -> k_bsf_combine: a 2 x 4 n byte round trip and a
//                             launch): the bin's part of the B-order stream is ONE contiguous range: 32 B
//                             of values + 16 B of 2-byte row indices per group.  The row sums of the bin live in LDS
//                             as 64-BIT FIXED-POINT numbers and every entry is one integer LDS atomic add (f32 LDS
//                             atomics run ~12x slower on gfx950: 386 us against 95 us for this kernel).  Integer sums
//                             do not depend on the order of the additions: the result is deterministic.  The scale is
//                             a power of two chosen per bin from max |value| of this launch (found by phase A) and the
//                             bin's largest row, so that no row sum can overflow: an entry keeps
//                             min(51, 62 - log2(rows' max entries)) bits below the launch's largest value -- f32-exact
//                             for every value above ~1e-8 of it, absolute error below 2^-51 of it otherwise.
//
// ~12.5 sequential bytes per cold entry instead of one L2 request.
//
// Used whenever the image holds >= ~10 M cold entries (pb_plan; PGH_PB=0 switches it off, PGH_PB_FORCE=1 lifts the size
// heuristics for tests); row-partitioned slices included, the multi-seed layout not yet.  Bench graph (RMAT scale 23):
// round 1: k_bsf_partial 83 us (hot entries only, 16-bit stream) + phase A 81 us + phase B 84 us (+ fix-up 11 + combine 49)
// against 375 us with the cold gathers left in the stream; round 2: 62 + 75 + 96 us, phase B including the epilogue.
// profiles/r01/pb_experiment_scale23.log holds the history (v1/v2: bins of <= 15 K ENTRIES staged in LDS and walked
// row-major, deterministic, but 6463 bins x 114 chunks made the runs 250 bytes long and phase B DRAM-inefficient).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "pgh_kernels.h"
#include "pgh_pb_gather.h"


namespace pgh {
namespace {

constexpr int kBlock = 256;
#ifndef PGH_PB_ROWS
#define PGH_PB_ROWS 4096
#endif
#ifndef PGH_PB_BTHREADS
#define PGH_PB_BTHREADS 256
#endif
// k_pb_finish build parameters (tools/build_variants.sh): groups per thread and stream round, rows per thread and epilogue
// round, whether the next item's first stream round is issued before the current item's epilogue
#ifndef PGH_FIN_P
#define PGH_FIN_P 1           // measured at scale 23 (profiles/r02/finish_pg_sweep.log): P x G = 4x4 108-113 us, 4x2 114, 2x4 104, 2x8 108, 6x4 118;
                              // round 5 (descriptors in scalar registers): 1 = 124 registers, nothing spilled, 91.9-93.3 us against 94.8-96.8 for 2
                              // (128 registers, 8 spilled) and 108-111 for 3; scale 22: 60.8-61.1 against 59.3-59.7
#endif
#ifndef PGH_FIN_STAGGER
#define PGH_FIN_STAGGER 0
#endif
#ifndef PGH_FIN_WPE
#define PGH_FIN_WPE 4         // wavefronts per SIMD the small shape is compiled for (register budget 512 / WPE)
#endif
#ifndef PGH_FIN_G
#define PGH_FIN_G 4
#endif
#ifndef PGH_FIN_PREFETCH
#define PGH_FIN_PREFETCH 0
#endif
#ifndef PGH_FIN_BF
#define PGH_FIN_BF 1          // the epilogue's operand loads without run-time branches (a missing operand reads psum's zero slot)
#endif
#ifndef PGH_FIN_G2
#define PGH_FIN_G2 2          // epilogue groups in flight per wavefront where the rows carry more operands (8 blocks, the in-kernel residual)
#endif
#ifndef PGH_FIN_PBIG
#define PGH_FIN_PBIG 1          // ... of the 512- and 1024-thread shapes (2 -> 1: 10 / 8 -> 2 registers spilled; finish -2 % at scale 25, -2 ... -6 % at scale 27 / ef 8,
                                // no different on the 8-way slice)
#endif
#ifndef PGH_FIN_UNI
#define PGH_FIN_UNI 1         // item descriptors as wavefront-uniform values (scalar registers, scalar branches on hub / rows)
#endif
// rows per bin = 64-bit row sums in LDS during phase B (<= 32768: 15-bit keys).  Two shapes: the small one (32 KB, four
// workgroups per CU) is the faster kernel; the large one (128 KB, one workgroup per CU) keeps the (chunk, bin) runs
// long enough on graphs with many chunks (row-partitioned slices gather from the whole source space).
constexpr int kPbBinRows = PGH_PB_ROWS, kPbBThreads = PGH_PB_BTHREADS;
constexpr int kPbBinRowsLarge = 16384, kPbBThreadsLarge = 1024;
// ... and a middle shape (round 5): 8192 rows, 512 threads, two workgroups per CU -- for graphs whose small bins would give a workgroup seven and
// more items of short runs but whose runs are still long enough at twice the rows (RMAT scales 25-26: finish + phase A -9 ... -12 % at scale 25,
// -4 % at scale 26; at scale 24 the small shape wins by 3 %, at scale 27 / ef 8 and on the 8-way slice the large one: profiles/r05/
// large_graphs_bin_shapes.log)
constexpr int kPbBinRowsMid = 8192, kPbBThreadsMid = 512;
constexpr int kPbBinFill = 6;                    // entries per bin <= kPbBinFill * rows (balance: the heavy rows come first);
                                                 // PGH_PB_BINFILL sweep at scale 23 (profiles/r02/binfill_sweep.log), k_pb_finish:
                                                 // 4 -> 108.9 us, 6 -> 101.6, 8 -> 108.2, 12 -> 107.7, 16 -> 133.9
constexpr int kPbHeavyRow = 16384;               // a row with more cold entries gets (hub) bins of its own:
constexpr int kPbHubMax = 65536;                 // one per this many entries ("pieces", entries dealt by source chunk),
constexpr int kPbMaxPieces = 1024;               // at most this many; beyond, its cold entries stay in the blocked stream
// (PGH_PB_HEAVY / PGH_PB_HUBMAX override the first two for tests: small graphs have no such rows)               // a row with more cold entries keeps them in the blocked stream
constexpr int kPbMaxChunks = 8192;               // 13-bit chunk field of the sort key
constexpr int kPbMaxBins = 32767;                // 15-bit bin field
constexpr uint64_t kLow29 = (1ULL << 29) - 1;

template <typename T>
struct PbBuf {
    T* p = nullptr;
    ~PbBuf() {
        if (p) (void)pooled_free(p);
    }
    int alloc(size_t count, bool zero = false) {
        PGH_HIP(pooled_malloc(&p, sizeof(T) * (count > 0 ? count : 1)));
        if (zero) PGH_HIP(hipMemsetAsync(p, 0, sizeof(T) * (count > 0 ? count : 1), rt().stream));
        return 0;
    }
};

inline int pb_blocks_for(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rt().num_cus * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ------------------------------------------------------------------------------------------------- build kernels
// The planner's input and output without two n-word transfers (round 6): the per-row cold counts go to the host as BYTES (255 = "look
// me up": the few rows with more are listed beside them), and the row -> bin map is made on the device from the bins' first rows.
__global__ void k_pb_counts_small(const uint32_t* __restrict__ counts, int n_out, unsigned char* __restrict__ small, uint32_t* __restrict__ big_rows,
                                  uint32_t* __restrict__ big_counts, uint32_t* __restrict__ big_n, uint32_t cap) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += gridDim.x * blockDim.x) {
        const uint32_t c = counts[i];
        small[i] = (unsigned char)(c < 255u ? c : 255u);
        if (c >= 255u) {
            const uint32_t at = atomicAdd(big_n, 1u);
            if (at < cap) {
                big_rows[at] = (uint32_t)i;
                big_counts[at] = c;
            }
        }
    }
}
// row_bin[i] = the bin that starts at row i (the first piece of a hub row), else the bin that started before it (bins ascend by first row)
__global__ void k_pb_row_bin(const int32_t* __restrict__ first_row, int num_bins, int n_out, int32_t* __restrict__ row_bin) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = num_bins;                     // first bin whose first row is >= i
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (first_row[mid] < i) lo = mid + 1;
            else hi = mid;
        }
        row_bin[i] = (lo < num_bins && first_row[lo] == i) ? lo : (lo > 0 ? lo - 1 : 0);
    }
}
__global__ void k_pb_rows_unbinned(const int32_t* __restrict__ rows, int count, int32_t* __restrict__ row_bin) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) row_bin[rows[i]] = -1;
}
// every stride_a-th word of a and every stride_b-th word of b, one after the other (the chunk / bin starts of the two orders)
__global__ void k_pb_pick_starts(const uint32_t* __restrict__ a, int64_t stride_a, int n_a, const uint32_t* __restrict__ b, int64_t stride_b, int n_b,
                                 uint32_t* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_a + n_b; i += gridDim.x * blockDim.x)
        out[i] = i < n_a ? a[(int64_t)i * stride_a] : b[(int64_t)(i - n_a) * stride_b];
}

// cold entries per output row (stream keys: block << 58 | row << 29 | col)
// (round 6: the entries of a row are consecutive in the sorted stream, so a wavefront adds the cold entries of every run of equal rows it
// holds with ONE atomic from the run's first lane -- 33 M atomics at scale 23 became ~8 M: 5.3 -> ~2 ms of the plan)
__global__ void k_pb_row_counts(const uint64_t* __restrict__ keys, const unsigned char* __restrict__ is_hot, int64_t E,
                                uint32_t* __restrict__ row_cold) {
    const int lane = threadIdx.x & 63;
    for (int64_t e0 = blockIdx.x * (int64_t)blockDim.x; e0 < E; e0 += (int64_t)gridDim.x * blockDim.x) {     // (uniform per workgroup)
        const int64_t e = e0 + threadIdx.x;
        const bool in = e < E;
        const uint64_t row = in ? (keys[e] >> 29) & kLow29 : kLow29;
        const bool cold = in && !is_hot[e] && row != kLow29;
        const uint64_t prev = (uint64_t)__shfl_up((long long)row, 1, 64);
        const bool head = lane == 0 || row != prev;
        const unsigned long long heads = __ballot(head), colds = __ballot(cold);
        if (head && row != kLow29) {
            const unsigned long long above = lane == 63 ? 0ULL : heads & ~((2ULL << lane) - 1ULL);       // heads in higher lanes
            const int next = above != 0ULL ? __builtin_ctzll(above) : 64;
            const unsigned long long upto = next == 64 ? ~0ULL : (1ULL << next) - 1ULL;
            const unsigned int count = (unsigned int)__popcll(colds & upto & ~((1ULL << lane) - 1ULL));
            if (count != 0u) atomicAdd(&row_cold[row], count);
        }
    }
}

// entries of rows without a bin (heavier than a bin) stay in the blocked stream
__global__ void k_pb_keep_heavy(const uint64_t* __restrict__ keys, int64_t E, const int32_t* __restrict__ row_bin,
                                unsigned char* __restrict__ is_hot) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        if (is_hot[e]) continue;
        const uint64_t row = (keys[e] >> 29) & kLow29;
        if (row == kLow29 || row_bin[row] < 0) is_hot[e] = 1;
    }
}

struct PbLayout {
    int64_t cold_prefix[9];            // dense numbering: first cold id of every block
    int     blk, hot, chunk;
    const uint32_t* rank;              // need lists: dense cold id -> compact cold id (null: the dense numbering is the image's)
};

// need lists: marks the dense cold ids that an entry of the image references
__global__ void k_pb_mark_cold(const uint64_t* __restrict__ keys, const unsigned char* __restrict__ is_hot, int64_t E, PbLayout L,
                               uint32_t* __restrict__ mark) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        if (is_hot[e]) continue;
        const uint64_t key = keys[e];
        if (((key >> 29) & kLow29) == kLow29) continue;           // sentinel / pad
        const int b = (int)(key >> 58);
        const int64_t loc = (int64_t)(key & kLow29) - (int64_t)b * L.blk;
        mark[L.cold_prefix[b] + (loc - L.hot)] = 1u;
    }
}
// ... and lists them: need_idx[compact id] = slot - hot inside its block
__global__ void k_pb_need_idx(const uint32_t* __restrict__ mark, const uint32_t* __restrict__ rank, int64_t dense_total, PbLayout L, int num_blocks,
                              uint32_t* __restrict__ need_idx) {
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < dense_total; d += (int64_t)gridDim.x * blockDim.x) {
        if (mark[d] == 0u) continue;
        int b = 0;
        for (int k = 1; k < num_blocks; ++k) b += d >= L.cold_prefix[k] ? 1 : 0;
        need_idx[rank[d]] = (uint32_t)(d - L.cold_prefix[b]);
    }
}

// stream key -> cell key (chunk << 45 | bin << 30 | row_in_bin << 15 | source_in_chunk)
__global__ void k_pb_keys(const uint64_t* __restrict__ keys, int64_t count, PbLayout L, const int32_t* __restrict__ row_bin,
                          int first_bin, const int4* __restrict__ bin /* of this slice */, uint64_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        const int b = (int)(key >> 58);
        const int64_t row = (int64_t)((key >> 29) & kLow29);
        const int64_t loc = (int64_t)(key & kLow29) - (int64_t)b * L.blk;
        const int64_t dense_id = L.cold_prefix[b] + (loc - L.hot);
        const int64_t cold_id = L.rank != nullptr ? (int64_t)L.rank[dense_id] : dense_id;
        const uint64_t c = (uint64_t)(cold_id / L.chunk), sl = (uint64_t)(cold_id % L.chunk);
        uint64_t w = (uint64_t)(row_bin[row] - first_bin);
        const uint64_t pieces = (uint64_t)(((unsigned)bin[w].y >> 22) & 0x3ffu) + 1u;      // a split hub row: piece by source chunk
        if (pieces > 1) w += c % pieces;
        const uint64_t dl = (uint64_t)(row - bin[w].x);
        out[i] = (c << 45) | (w << 30) | (dl << 15) | sl;
    }
}

// entries per cell, [chunk][bin]
// (the keys are sorted by (chunk, bin): a wavefront adds every run of one cell it holds with ONE atomic from the run's first lane)
__global__ void k_pb_cell_counts(const uint64_t* __restrict__ keys, int64_t count, int num_bins, uint32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < count; i0 += (int64_t)gridDim.x * blockDim.x) {     // (uniform per workgroup)
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < count;
        const uint64_t cell = in ? keys[i] >> 30 : ~0ULL;                                   // chunk << 15 | bin
        const uint64_t prev = (uint64_t)__shfl_up((long long)cell, 1, 64);
        const bool head = lane == 0 || cell != prev;
        const unsigned long long heads = __ballot(head), valid = __ballot(in);
        if (head && in) {
            const unsigned long long above = lane == 63 ? 0ULL : heads & ~((2ULL << lane) - 1ULL);
            const int next = above != 0ULL ? __builtin_ctzll(above) : 64;
            const unsigned long long upto = next == 64 ? ~0ULL : (1ULL << next) - 1ULL;
            const unsigned int n = (unsigned int)__popcll(valid & upto & ~((1ULL << lane) - 1ULL));
            atomicAdd(&counts[(cell >> 15) * (uint64_t)num_bins + (cell & 0x7fffu)], n);
        }
    }
}

// padded run lengths in both orders (the scans of these give the run starts)
__global__ void k_pb_padded(const uint32_t* __restrict__ counts, int num_chunks, int num_bins, uint32_t* __restrict__ pad_a /* [chunk][bin] */,
                            uint32_t* __restrict__ pad_b /* [bin][chunk] */) {
    const int64_t cells = (int64_t)num_chunks * num_bins;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / num_bins, w = i % num_bins;
        const uint32_t padded = (counts[i] + 7u) & ~7u;
        pad_a[i] = padded;
        pad_b[w * num_chunks + c] = padded;
    }
}

// sorted rank i -> its place in both orders; the first entry of every group of 8 records where the group goes
__global__ void k_pb_place(const uint64_t* __restrict__ keys, const float* __restrict__ vals, int64_t count, int num_chunks, int num_bins,
                           const uint32_t* __restrict__ first /* [chunk][bin] unpadded rank of the run's first entry */,
                           const uint32_t* __restrict__ start_a, const uint32_t* __restrict__ start_b, uint16_t* __restrict__ sloc,
                           float* __restrict__ val, uint32_t* __restrict__ dstg, uint16_t* __restrict__ drow) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        const uint64_t c = key >> 45, w = (key >> 30) & 0x7fffu, dl = (key >> 15) & 0x7fffu, sl = key & 0x7fffu;
        const int64_t cell = (int64_t)c * num_bins + (int64_t)w;
        const uint32_t off = (uint32_t)(i - (int64_t)first[cell]);
        const uint32_t pa = start_a[cell] + off, pb = start_b[(int64_t)w * num_chunks + (int64_t)c] + off;
        sloc[pa] = (uint16_t)sl;
        if (vals != nullptr) val[pa] = vals[i];
        drow[pb] = (uint16_t)dl;
        if ((off & 7u) == 0u) dstg[pa >> 3] = pb >> 3;
    }
}

// ------------------------------------------------------------------------------------------------- run-time kernels
PGH_STAMP_DECL(g_times_gather)
PGH_STAMP_DECL(g_times_finish)
#if PGH_PROBE_TIMES
__device__ unsigned int g_item_ticks[1 << 15];      // per work item of k_pb_finish: duration in 10 ns ticks
__device__ unsigned int g_item_begin[1 << 15];      // ... and its start relative to the workgroup's start
__device__ unsigned int g_phase_ticks[8 * 4096];   // per workgroup: ticks in the phases of its items (setup, stream, hand-over, epilogue, tail)
#endif

// ---- phase A (body: pgh_pb_gather.h, shared with the merged front kernel of a step in pgh_bsf.hip)
template <bool HAS_VAL, int PG, bool DROP = false>
__global__ __launch_bounds__(kPbThreads) void k_pb_gather(PbView f, const float* __restrict__ xg, const LoopState* __restrict__ state,
                                                          FixView fix, DropView dv = DropView{}) {
    __shared__ float s_x[kPbChunk + 1];
    if (state != nullptr && state->done) return;
    PGH_STAMP_BEGIN(g_times_gather)
    // the cross-tile fix-ups of the blocked stream ride along (one launch and one dependent boundary fewer per step):
    // they touch nothing this kernel reads, and the next kernel (k_pb_finish) is the first to read their results
    bsf_fixup_tiles(fix, blockIdx.x * kPbThreads, gridDim.x * kPbThreads);
    pb_gather_body<HAS_VAL, PG, DROP>(s_x, reinterpret_cast<uint32_t*>(s_x + kPbChunk), f, xg, blockIdx.x, dv);
    PGH_STAMP_END(g_times_gather)
}

// ---- phase B + epilogue
// Work items (PbFormat::item_a / item_b) tile the output rows: a regular bin (<= ROWS consecutive rows whose cold sums
// live in LDS), a stretch of rows without cold entries (epilogue only), or one piece of a hub row.  A persistent grid
// walks the list with a fixed stride, so the per-workgroup partial sums of sum(y) / delta are deterministic.
//
// Regular bin: the bin's entries are one contiguous range of whole 8-entry groups in B order (pad entries carry row
// 0xffff).  Row sums: 64-bit fixed point in LDS, integer atomics (see the head of this file).  Then, for every row of
// the bin: row sum = sum over the column blocks of the row's segment in psum (block_row_sum) + the cold sum, times the
// output scale, and the filter's epilogue (apply_epilogue) -- what k_bsf_combine does for graphs without a cold image.
//
// Hub row (more than kPbHeavyRow cold entries): thousands of atomics on one LDS word would serialise, so its values are
// summed in f64 registers and reduced in a fixed order.  A row split into pieces: every piece publishes its sum with a
// device-scope atomic and takes a ticket; the last arriver adds the pieces in index order (deterministic whatever the
// arrival order) and runs the row's epilogue.
// RES (PageRank epilogue only): the step's residual is evaluated here as well, against the predicted quotient (ResParams,
// pgh_kernels.h) -- two more operands per row (the previous iterate, the row sum of M), three more partial sums.
template <int MODE, int NB, int ROWS, int THREADS, bool RES = false>
__global__ __launch_bounds__(THREADS, THREADS > 512 ? 4 : PGH_FIN_WPE) void k_pb_finish(PbView f, RowSums rs, const float* __restrict__ dst_scale, EpiParams ep,
                                                        const LoopState* __restrict__ state, double* __restrict__ partial_sum,
                                                        double* __restrict__ partial_delta, ResParams rp) {
    static_assert(!RES || MODE == EPI_AXPBY, "the in-kernel residual is PageRank's");
    constexpr int WAVES = THREADS / 64;
    // groups per thread and stream round in flight (three 16-byte loads each), rows per thread and epilogue round.  Both
    // shapes keep 16 wavefronts per CU (128 registers): 4 workgroups of 256 threads (32 KB of row sums each) or one of
    // 1024; what covers the latencies an item exposes (stream, atomics, epilogue rounds) is the depth of each round plus
    // the other workgroups of the CU.
    constexpr int P = THREADS > 256 ? PGH_FIN_PBIG : PGH_FIN_P;
    // (with the in-kernel residual: two more operands per row -- 2 rows in flight measured 108 us, 3: 112-114, 4: 119)
    constexpr int G = (THREADS > 512 || NB > 4 || RES) ? PGH_FIN_G2 : PGH_FIN_G;
    constexpr int WORDS = ROWS / 64 + 1;                   // map words an item can touch per block (unaligned first row)
    __shared__ unsigned long long s_row[ROWS];
    __shared__ unsigned long long s_mask[NB * WORDS];      // the item's slice of the row -> segment map (BsfFormat::meta)
    __shared__ int s_base[NB * WORDS];
    __shared__ double s_red[RES ? 4 * WAVES : WAVES];
    __shared__ float s_hub;
    __shared__ int s_last;
    // The words a workgroup needs before its first stream round -- the launch's largest value, its slice of the schedule, the descriptors
    // of its first item, the isolated-row flag -- are ASKED FOR before the first of them is used (round 5: as a chain state -> amax ->
    // slice -> schedule -> item, each made uniform where it was loaded, they were five round trips at the head of every launch)
    const uint32_t amax_raw = __builtin_nontemporal_load(f.amax);
    const int sb0_raw = f.sched_begin[blockIdx.x], sb1_raw = f.sched_begin[blockIdx.x + 1];
    const int4 fa_raw = f.first_a[blockIdx.x], fb_raw = f.first_b[blockIdx.x];
    const int fi_raw = f.first_item[blockIdx.x];
    const int iso_raw = *(f.iso_flag != nullptr ? f.iso_flag : reinterpret_cast<const int*>(f.amax));
    double scale = 1.0;
    if (state != nullptr) {
        if (state->done) return;
        scale = state->scale;
    }
    const float a_eff = (float)(ep.a * scale);
    const int tid = threadIdx.x;
    const uint32_t amax = PGH_FIN_UNI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)amax_raw) : amax_raw;
    const bool finite = amax < 0x7f800000u;                // inf / NaN among the values: the sums are not representable
    // |value| <= amax < 2^e; an entry gets E = min(51, 62 - count_bits) bits: |value * S| < 2^E with S = 2^(E - e), and a
    // row of <= 2^count_bits entries stays below 2^62
    const int e = (int)(amax >> 23) - 126;                 // amax < 2^e (denormals: e = -126, still an upper bound)
    constexpr double kMagic = 6755399441055744.0;          // 1.5 * 2^52: fma(v, S, magic) holds round(v * S) in its low bits
    double sum_y = 0.0, delta = 0.0;
    // in-kernel residual: R' and D against the predicted quotient, T for the next prediction (first step: sum(p) in D)
    double res_r = 0.0, res_d = 0.0, res_t = 0.0;
    // (the close bounds what the predicted quotient can cost by 2 |inv - inv'| sum_i |y_i| and takes sum(y) for that sum: a
    // NEGATIVE y -- a signed personalization -- voids the bound, so such a workgroup reports R' = NaN and the close pauses the
    // fusion: the separate residual kernel decides, as for every other value it cannot vouch for)
    bool res_neg = false;
    const double inv_pred = RES && !rp.first ? rp.aux->pred_inv[rp.step & 1] : 1.0;
    const bool res_first = RES && rp.first != 0;
    // Branch-free operand loads (PGH_FIN_BF): which operands a run has is decided per LAUNCH, but a load under a run-time branch --
    // even a wavefront-uniform one -- gets a basic block and a drained wait of its own and splits an epilogue round into several
    // exposed latencies.  A missing operand therefore reads the zero slot of the partial sums and its use is a select.
    const bool has_xg = ep.xg_out != nullptr, has_ds = dst_scale != nullptr;
    const uint32_t zero_off = rs.zero_at << 2;
    const char* const zero_base = reinterpret_cast<const char*>(rs.psum);
    const char* const ds_base = has_ds ? reinterpret_cast<const char*>(dst_scale) : zero_base;
    const int xg_shift = (ep.xg_blk > 0 && (ep.xg_blk & (ep.xg_blk - 1)) == 0) ? __ffs(ep.xg_blk) - 1 : -1;
    auto slot_of = [&](int row) __attribute__((always_inline)) {
        int b;
        if (xg_shift >= 0) b = (int)((unsigned)row >> xg_shift);
        else {
            b = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) b += (row >= k * ep.xg_blk) ? 1 : 0;
        }
        const int loc = row - b * ep.xg_blk;
        const int s = loc < ep.xg_hot ? b * ep.xg_hot + loc : ep.xg_cold + b * (ep.xg_live - ep.xg_hot) + (loc - ep.xg_hot);
        const int stored = ep.xg_live == 0 ? row : (loc < ep.xg_live ? s : -1);
        return has_xg ? stored : -1;
    };
    auto residual_row = [&](float y, float x_prev, float deg, float pv) __attribute__((always_inline)) {
        res_t += (double)deg * (double)y;
        res_neg = res_neg || y < 0.f;
        if (res_first) {
            res_d += (double)pv;
        } else {
            const double d = (double)y * inv_pred - (double)x_prev * scale;
            res_r += fabs(d);
            res_d += d < 0.0 ? -(double)y : (double)y;
        }
    };

    struct Round {
        u16x8 r8[P];
        f32x4 lo[P], hi[P];
    };
    // groups tid + (round * P + q) * THREADS of the bin `bin` (pad rows 0xffff beyond the bin's range)
    auto fetch = [&](const int4& bin, int round, Round& R) __attribute__((always_inline)) {
        const bool hub = ((bin.y >> 21) & 1) != 0;
        const int groups = finite || hub ? bin.w : 0;
        const float* __restrict__ tmp = f.tmp;
        const uint16_t* __restrict__ drow = f.drow + (int64_t)bin.z * 8;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int g = tid + (round * P + q) * THREADS;
            const bool ok = g < groups && !(PGH_PROBE_PB & 8);
            R.r8[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(drow + (int64_t)g * 8))
                         : u16x8{0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff};
#if PGH_PB_TMP_PLAINLOAD
            R.lo[q] = ok ? *reinterpret_cast<const f32x4*>(tmp + pb_tmp_quad((uint32_t)(bin.z + g), 0, f.tmp_planes)) : f32x4{0.f, 0.f, 0.f, 0.f};
            R.hi[q] = ok ? *reinterpret_cast<const f32x4*>(tmp + pb_tmp_quad((uint32_t)(bin.z + g), 1, f.tmp_planes)) : f32x4{0.f, 0.f, 0.f, 0.f};
#else
            R.lo[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tmp + pb_tmp_quad((uint32_t)(bin.z + g), 0, f.tmp_planes))) : f32x4{0.f, 0.f, 0.f, 0.f};
            R.hi[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tmp + pb_tmp_quad((uint32_t)(bin.z + g), 1, f.tmp_planes))) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
        }
        // (the per-lane `ok` branches must meet again HERE, in a block of their own: when their join is also the join of the
        // workgroup-uniform branches around the call, every value merged there -- the item descriptors -- counts as divergent and
        // moves from scalar to vector registers, exec masks and all)
        if (PGH_FIN_UNI) asm volatile("; stream round issued");
    };

    // item descriptors are the same for every lane of the workgroup; said so, they live in scalar registers and the tests on them
    // (hub, rows, phase) are scalar branches instead of exec masks around every entry of the stream loop
    auto uni = [](int v) __attribute__((always_inline)) { return PGH_FIN_UNI ? __builtin_amdgcn_readfirstlane(v) : v; };
    auto uni4 = [&](const int4& v) __attribute__((always_inline)) { return make_int4(uni(v.x), uni(v.y), uni(v.z), uni(v.w)); };
    PGH_STAMP_BEGIN(g_times_finish)
#if PGH_FIN_STAGGER > 0
    // co-resident workgroups (the persistent grid is a multiple of the CU count: blockIdx / 256 = which of a CU's slots) start
    // their item loops a fraction of an item apart, so that their traffic-free phases do not coincide
    for (int k = 0; k < (int)(blockIdx.x >> 8) * PGH_FIN_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
#endif
    // partials of sum(y) (and delta) accumulated since the last flush -> slot `where`: wavefront shuffle, then a fixed-order
    // sum over the wavefronts; the accumulators start over
    auto flush = [&](int where) __attribute__((always_inline)) {
        if (RES) {
            // four sums at once: one barrier pair
            const double v0 = wave_reduce_sum(sum_y), v2 = wave_reduce_sum(res_d), v3 = wave_reduce_sum(res_t);
            double v1 = wave_reduce_sum(res_r);
            if (__any(res_neg)) v1 = __longlong_as_double(0x7ff8000000000000LL);
            if ((tid & 63) == 0) {
                s_red[tid >> 6] = v0;
                s_red[WAVES + (tid >> 6)] = v1;
                s_red[2 * WAVES + (tid >> 6)] = v2;
                s_red[3 * WAVES + (tid >> 6)] = v3;
            }
            __syncthreads();
            if (tid < 4) {
                double total = 0.0;
                for (int w = 0; w < WAVES; ++w) total += s_red[tid * WAVES + w];
                double* out = tid == 0 ? partial_sum : (tid == 1 ? rp.part_r : (tid == 2 ? rp.part_d : rp.part_t));
                out[where] = total;
            }
            __syncthreads();
            sum_y = 0.0, res_r = 0.0, res_d = 0.0, res_t = 0.0;
            res_neg = false;
            return;
        }
        double v = wave_reduce_sum(sum_y);
        if ((tid & 63) == 0) s_red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double total = 0.0;
            for (int w = 0; w < WAVES; ++w) total += s_red[w];
            partial_sum[where] = total;
        }
        if (MODE == EPI_POLY) {
            __syncthreads();
            v = ep.err_linf ? wave_reduce_max(delta) : wave_reduce_sum(delta);
            if ((tid & 63) == 0) s_red[tid >> 6] = v;
            __syncthreads();
            if (tid == 0) {
                double total = 0.0;
                for (int w = 0; w < WAVES; ++w) total = ep.err_linf ? fmax(total, s_red[w]) : total + s_red[w];
                partial_delta[where] = total;
            }
        }
        __syncthreads();
        sum_y = 0.0;
        delta = 0.0;
    };
    // Item schedule.  HEAD: static slices of f.sched per workgroup (round-robin in row order), their sum(y) / delta go to
    // the workgroup's partial.  TAIL: the last items of the list (the coldest rows: small items) are handed out by a device
    // counter to whoever has finished its slice; each tail item has a partial slot of its own, so the fold (workgroup
    // partials in workgroup order, then tail partials in item order) does not depend on who processed what.
    // Why: per-item times (PGH_PROBE_TIMES build, profiles/r02/finish_item_times.csv) are 10.6 us + 0.34 us per 1000
    // entries + 2.4 us per 1000 rows with a residual of +-4 us that no property of the item explains (it follows the
    // workgroup's place and time on the chip), so sums over a static slice spread by +-8 us and the slowest of 1024
    // workgroups ended at 98 us with the mean at 79.  Static deals by modelled cost (snake, longest-first) measured
    // 115-120 us against 106, a fully dynamic hand-out 135 us (every item pays the counter's round trip and a reduction),
    // one item per workgroup 117 us, second register sets for the stream / epilogue rounds 107-115 us.
    __shared__ int s_next;
    const int tail_count = f.tail_count;
    int at = uni(sb0_raw);
    const int at_end = uni(sb1_raw);
    // two-launch form (PbView::phase): phase 2 keeps its partials behind phase 1's; an item belongs to phase 1 when one of its rows has
    // an exchanged slot -- its first row's slot inside the block lies below phase_live, or its rows run into the next block
    const int slot0 = f.phase == 2 ? (int)gridDim.x + tail_count : 0;
    auto in_phase = [&](const int4& e) __attribute__((always_inline)) {
        if (f.phase == 0) return true;
        const int loc = (int)((unsigned)e.x % (unsigned)f.phase_blk);
        const bool exchanged = loc < f.phase_live || loc + max(e.y, 1) > f.phase_blk;
        return exchanged == (f.phase == 1);
    };
    int slot = slot0 + blockIdx.x, item = -1;       // partial-sum slot of the item in hand
    bool in_tail = at >= at_end;
    auto take_tail = [&]() __attribute__((always_inline)) {        // -> item index or -1; all threads call it together
        if (tail_count == 0) return -1;
        __syncthreads();
        if (tid == 0) s_next = (int)atomicAdd(f.work_counter, 1u);
        __syncthreads();
        const int k = uni(s_next);
        if (k >= tail_count) return -1;
        slot = slot0 + gridDim.x + k;
        return uni(f.sched[f.tail_begin + k]);
    };
    bool flushed_head = false;
    if (!in_tail) item = uni(fi_raw);
    else {
        flush(slot0 + blockIdx.x);          // no static items: the workgroup's partial is zero
        flushed_head = true;
        item = take_tail();
    }
    int4 bin = make_int4(0, 0, 0, 0), epi = make_int4(0, 0, -1, 0);
    Round R;
    // items marked -2 cover isolated rows (no entry, referenced by nobody): unless this run's operands are non-zero there
    // they hold zeros in both iterates and are passed over (the item shrinks to no rows: barriers only)
    const bool skip_iso = f.iso_flag != nullptr && uni(iso_raw) == 0;
    bool mine = false;                      // the item in hand belongs to this launch's phase
    if (item >= 0) {
        const bool first_static = !in_tail;   // (its descriptors came with the start-up words; a ticket's item is looked up)
        bin = first_static ? uni4(fa_raw) : uni4(f.item_a[item]);   // {first row, rows | log2ceil(largest row) << 16 | hub << 21 | (pieces - 1) << 22, first group, groups}
        epi = first_static ? uni4(fb_raw) : uni4(f.item_b[item]);   // {first row of the epilogue range, rows, split index or -1 (-2: isolated rows), first item of the split row}
        mine = in_phase(epi);
        if (mine) fetch(bin, 0, R);
    }
#if PGH_PROBE_TIMES
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned int ph[5] = {0u, 0u, 0u, 0u, 0u};
    unsigned int ph_items = 0u;
#define PGH_PHASE(K, SINCE) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); ph[K] += (unsigned int)(now_ - SINCE); SINCE = now_; }
#else
#define PGH_PHASE(K, SINCE)
#endif
    while (item >= 0) {
#if PGH_PROBE_TIMES
        const unsigned long long item_t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long ph_t = item_t0;
        ++ph_items;
#endif
        if (!mine) {                           // the other launch's item: only the walk over the schedule goes on (workgroup-uniform)
            const int cur_slot = slot;
            int next = -1;
            if (!in_tail && ++at < at_end) next = uni(f.sched[at]);
            else {
                in_tail = true;
                next = take_tail();
            }
            if (cur_slot != slot0 + (int)blockIdx.x) flush(cur_slot);
            else if (in_tail) {
                flush(slot0 + blockIdx.x);
                flushed_head = true;
            }
            item = next;
            if (item >= 0) {
                bin = uni4(f.item_a[item]);
                epi = uni4(f.item_b[item]);
                mine = in_phase(epi);
                if (mine) fetch(bin, 0, R);
            }
            continue;
        }
        if (epi.z == -2) {
            // isolated rows: no entry, no segment in any block.  Passed over while the run's operands are zero there; otherwise
            // the epilogue of an empty row sum, streamed (no row map, no partial sums), and the item is empty from here on.
            if (!skip_iso) {
                constexpr int UI = 4;                          // rows per thread in flight
                const int row_end = epi.x + epi.y;
                for (int row0 = epi.x + tid; row0 < row_end; row0 += THREADS * UI) {
                    EpiOps ops[UI];
                    float xp[UI], dg[UI];
#pragma unroll
                    for (int u = 0; u < UI; ++u) {
                        const int row = min(row0 + u * THREADS, row_end - 1);
                        ops[u] = epi_load<MODE>(ep, row);
                        if (RES) {
                            xp[u] = ld_off(rp.x_prev, (uint32_t)row << 2);
                            dg[u] = ld_off(rp.deg, (uint32_t)row << 2);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UI; ++u)
                        if (row0 + u * THREADS < row_end) {
                            const float y = epi_apply<MODE>(ep, ops[u], a_eff, row0 + u * THREADS, 0.f, sum_y, delta);
                            if (RES) residual_row(y, xp[u], dg[u], ops[u].v);
                        }
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
        // ---- the item's slice of the row -> segment map: issued now, parked in LDS after the stream (below)
        const int word0 = epi.x >> 6;
        const int words = hub ? 0 : ((epi.x + epi.y - 1) >> 6) - word0 + 1;
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
        // the id of the item after this one: asked for now, needed after the stream (static slice; the tail's ticket is taken there)
        const bool next_static = !in_tail && at + 1 < at_end;
        int next_id_raw = -1;
        if (PGH_FIN_UNI && next_static) next_id_raw = f.sched[at + 1];
        double hub_sum = 0.0;
        if (!hub)
            for (int i = tid; i < rows; i += THREADS) s_row[i] = 0ULL;
        __syncthreads();
        PGH_PHASE(0, ph_t)
        const int groups = finite || hub ? bin.w : 0;
        const int nrounds = (groups + THREADS * P - 1) / (THREADS * P);
        // one stream round: every entry of a regular bin is one integer LDS atomic; a hub piece sums its values in registers
        auto consume = [&](auto is_hub) __attribute__((always_inline)) {
            constexpr bool HUB = decltype(is_hub)::value;
            if (PGH_PROBE_PB & 16) {
                float z = 0.f;
#pragma unroll
                for (int q = 0; q < P; ++q) z += R.lo[q].x + R.hi[q].w + (float)R.r8[q][3];
                if (z == 123.456f) s_row[0] = 1ULL;
                return;
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int r = (int)R.r8[q][k];
                    if (PGH_PROBE_PB & 256) r = (r >> 15) + ((tid & 63) + 64 * k) % rows;   // diagnostic: atomics without bank / address conflicts (wrong sums)
                    const float v = k < 4 ? R.lo[q][k] : R.hi[q][k - 4];
                    if (HUB) {
                        if (r == 0) hub_sum += (double)v;
                    } else if (r < rows) {
                        const long long fixed = __double_as_longlong(__builtin_fma((double)v, S, kMagic)) - __double_as_longlong(kMagic);
                        atomicAdd(&s_row[r], (unsigned long long)fixed);
                    }
                }
            }
        };
        if (PGH_FIN_UNI && hub) {
            for (int round = 0; round < nrounds; ++round) {
                consume(std::true_type{});
                if (round + 1 < nrounds) fetch(bin, round + 1, R);
            }
        } else if (PGH_FIN_UNI) {
            for (int round = 0; round < nrounds; ++round) {
                consume(std::false_type{});
                if (round + 1 < nrounds) fetch(bin, round + 1, R);
            }
        } else {
            for (int round = 0; round < nrounds; ++round) {
                if (hub) consume(std::true_type{});
                else consume(std::false_type{});
                if (round + 1 < nrounds) fetch(bin, round + 1, R);
            }
        }
        PGH_PHASE(1, ph_t)
        // ---- the next item's descriptors are asked for before this item's epilogue and read after it
        const int cur_slot = slot;
        int next = -1;
        if (PGH_FIN_UNI) {
            if (next_static) {
                ++at;
                next = uni(next_id_raw);
            } else {
                in_tail = true;
                next = take_tail();
            }
        } else {
            if (!in_tail && ++at < at_end) next = f.sched[at];
            else {
                in_tail = true;
                next = take_tail();
            }
        }
        int4 next_bin = make_int4(0, 0, 0, 0), next_epi = make_int4(0, 0, -1, 0);
        bool next_mine = false;
        if (next >= 0) {
            next_bin = f.item_a[next];        // (not yet wavefront-uniform values: the wait for them belongs behind the epilogue)
            next_epi = f.item_b[next];
            if (!PGH_FIN_UNI || PGH_FIN_PREFETCH) {
                next_bin = uni4(next_bin), next_epi = uni4(next_epi);
                next_mine = in_phase(next_epi);
                if (PGH_FIN_PREFETCH && next_mine) fetch(next_bin, 0, R);
            }
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
            // one row: fixed-order reduction of the threads' f64 sums; of a split row only the last arriver continues
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) hub_sum += __shfl_xor(hub_sum, d, 64);
            if ((tid & 63) == 0) s_red[tid >> 6] = hub_sum;
            __syncthreads();
            if (tid == 0) {
                double total = 0.0;
                for (int w = 0; w < WAVES; ++w) total += s_red[w];
                int last = 1;
                if (pieces > 1) {
                    // publish the piece (device-scope exchange: performed at the memory side), then take a ticket
                    unsigned long long* slot = reinterpret_cast<unsigned long long*>(f.hub_part) + item;
                    (void)atomicExch(slot, (unsigned long long)__double_as_longlong(total));
                    __threadfence();
                    last = atomicAdd(f.hub_ticket + epi.z, 1u) == (unsigned)(pieces - 1) ? 1 : 0;
                }
                s_hub = (float)total;
                s_last = last;
            }
            __syncthreads();
            if (pieces > 1 && s_last) {
                // the pieces of the row are the items epi.w .. epi.w + pieces - 1: read them coherently (atomic add of 0),
                // add them in index order
                __threadfence();
                unsigned long long* part = reinterpret_cast<unsigned long long*>(f.hub_part) + epi.w;
                double* s_piece = reinterpret_cast<double*>(s_row);
                for (int k = tid; k < pieces; k += THREADS) s_piece[k] = __longlong_as_double((long long)atomicAdd(part + k, 0ULL));
                __syncthreads();
                if (tid == 0) {
                    double total = 0.0;
                    for (int k = 0; k < pieces; ++k) total += s_piece[k];
                    s_hub = (float)total;
                    (void)atomicExch(f.hub_ticket + epi.z, 0u);       // re-arm for the next launch
                }
                __syncthreads();
            }
            if (s_last && tid == 0) {                      // the row's epilogue (one row: the direct lookup)
                const int64_t row = epi.x;
                double sum = block_row_sum<NB>(rs, row) + (double)s_hub;
                if (dst_scale != nullptr) sum *= (double)dst_scale[row];
                const EpiOps ops = epi_load<MODE>(ep, (int)row);
                const float y = epi_apply<MODE>(ep, ops, a_eff, (int)row, (float)sum, sum_y, delta);
                if (RES) residual_row(y, rp.x_prev[row], rp.deg[row], ops.v);
            }
        } else {
            __syncthreads();
            PGH_PHASE(2, ph_t)
            // The epilogue walks the item's rows in ALIGNED groups of 64 (lane = row % 64), G groups per wavefront in
            // flight.  The map word of a (block, group) is then wavefront-uniform: one broadcast read of the LDS copy,
            // the lane's bit is a shift, its rank among the group's segments is v_mbcnt -- a handful of vector
            // instructions per block instead of 64-bit masks per lane (scalar loads of the words straight from
            // memory were tried: 260 us, every s_load a drained wait).  Branch-free loads: lanes outside the
            // item repeat a row of it, rows without a segment in a block read the zero slot, results are dropped later
            // (a load under a divergent branch gets its own wait and serialises everything around it).
            const int lane = tid & 63;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int row_lo = epi.x, row_hi = epi.x + epi.y - 1;
            const int g_hi = (PGH_PROBE_PB & 32) ? -1 : (row_hi >> 6);          // diagnostic: no epilogue
            const char* __restrict__ psum_b = reinterpret_cast<const char*>(rs.psum);
            constexpr bool STRAIGHT = PGH_FIN_BF != 0;
            for (int g0 = (row_lo >> 6) + wave; g0 <= g_hi; g0 += WAVES * G) {
                EpiOps ops[G];
                float dsc[G];
                float xp[RES ? G : 1], dg[RES ? G : 1];
                float vals[G][NB];
                // (the slots of the next gather vector first: integer work that must not sit between the loads and read a register
                // pair one half of which a load is still to fill -- the compiler's 64-bit multiply-add did, and waited for everything)
                if (STRAIGHT) {
#pragma unroll
                    for (int u = 0; u < G; ++u) ops[u].slot = slot_of(((g0 + u * WAVES) << 6) + lane);
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int g = min(g0 + u * WAVES, g_hi);                 // wavefront-uniform
                    const int row = min(max((g << 6) + lane, row_lo), row_hi);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        // blocks past num_blocks hold empty words (mask 0): the zero slot
                        const unsigned long long mask = s_mask[b * WORDS + (g - word0)];
                        const unsigned int first = (unsigned int)s_base[b * WORDS + (g - word0)];
                        const unsigned int lo = (unsigned int)mask, hi = (unsigned int)(mask >> 32);
                        const unsigned int rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                        // by LANE, also for lanes whose row lies outside the item: row 64 g + lane exists in the map
                        const bool has = (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) != 0u;
                        const unsigned int at = has ? first + rank : rs.zero_at;
                        vals[u][b] = *reinterpret_cast<const float*>(psum_b + (at << 2));
                    }
                    if (STRAIGHT) {
                        const int slot = ops[u].slot;
                        ops[u] = epi_load_z<MODE>(ep, row, zero_base + zero_off);
                        ops[u].slot = slot;
                        dsc[u] = *reinterpret_cast<const float*>(ds_base + (has_ds ? (uint32_t)row << 2 : zero_off));
                    } else {
                        ops[u] = epi_load<MODE>(ep, row);
                        dsc[u] = dst_scale != nullptr ? ld_off(dst_scale, (uint32_t)row << 2) : 1.f;
                    }
                    if (RES) {
                        xp[u] = ld_off(rp.x_prev, (uint32_t)row << 2);
                        dg[u] = ld_off(rp.deg, (uint32_t)row << 2);
                    }
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int g = g0 + u * WAVES;
                    const int row = (g << 6) + lane;
                    const bool live = g <= g_hi && row >= row_lo && row <= row_hi;
                    if (!STRAIGHT && !live) continue;
                    // STRAIGHT: every group is computed (dead lanes hold a repeated row of the item) and only the stores and the sums
                    // are predicated -- a path that skips the uses of a group leaves its loads pending where the paths meet, and the
                    // head of the next round then waits for everything in flight, the last stores included
                    const int i = live ? row - row_lo : 0;
                    float cold = 0.f;
                    if (STRAIGHT) {
                        const float c = finite ? (float)((double)(long long)s_row[i < rows ? i : 0] * inv_S) : __uint_as_float(0x7fc00000u);
                        cold = i < rows ? c : 0.f;
                    } else if (i < rows) cold = finite ? (float)((double)(long long)s_row[i] * inv_S) : __uint_as_float(0x7fc00000u);
                    double sum = 0.0;
#pragma unroll
                    for (int b = 0; b < NB; ++b) sum += (double)vals[u][b];      // blocks past num_blocks contribute 0
                    sum += (double)cold;
                    float y;
                    if (STRAIGHT) {
                        sum = has_ds ? sum * (double)dsc[u] : sum;
                        // (a dead lane adds +0 to every sum: sum(y), delta, T, R', D)
                        y = epi_apply_z<MODE>(ep, ops[u], a_eff, live ? row : row_lo, (float)sum, live, sum_y, delta);
                        if (RES) residual_row(y, live ? xp[u] : 0.f, dg[u], live ? ops[u].v : 0.f);
                    } else {
                        if (dst_scale != nullptr) sum *= (double)dsc[u];
                        y = epi_apply<MODE>(ep, ops[u], a_eff, row, (float)sum, sum_y, delta);
                        if (RES) residual_row(y, xp[u], dg[u], ops[u].v);
                    }
                }
            }
        }
        PGH_PHASE(3, ph_t)
        __syncthreads();                                   // s_row / s_hub are reused by the next item
#if PGH_PROBE_TIMES
        if (tid == 0 && item < (1 << 15)) {
            g_item_ticks[item] = (unsigned int)(__builtin_amdgcn_s_memrealtime() - item_t0);
            g_item_begin[item] = (unsigned int)(item_t0 - wg_t0);
        }
#endif
        // the head's partial leaves when the slice is done, a tail item's partial after the item
        if (cur_slot != slot0 + (int)blockIdx.x) flush(cur_slot);
        else if (in_tail) {
            flush(slot0 + blockIdx.x);
            flushed_head = true;
        }
        if (PGH_FIN_UNI && !PGH_FIN_PREFETCH && next >= 0) {
            next_bin = uni4(next_bin), next_epi = uni4(next_epi);
            next_mine = in_phase(next_epi);
        }
        if (!PGH_FIN_PREFETCH && next >= 0 && next_mine) fetch(next_bin, 0, R);
        item = next;
        bin = next_bin;
        epi = next_epi;
        mine = next_mine;
        PGH_PHASE(4, ph_t)
    }
#if PGH_PROBE_TIMES
    if (tid == 0 && blockIdx.x < 4096) {
        for (int k = 0; k < 5; ++k) g_phase_ticks[8 * blockIdx.x + k] = ph[k];
        g_phase_ticks[8 * blockIdx.x + 5] = ph_items;
        g_phase_ticks[8 * blockIdx.x + 6] = (unsigned int)(__builtin_amdgcn_s_memrealtime() - wg_t0);
    }
#endif
    if (!flushed_head) flush(slot0 + blockIdx.x);
    // the last workgroup to leave re-arms the words for the next launch (every workgroup has read amax long before its
    // ticket; the next phase A starts after this kernel).  Phase 1 of a two-launch finish leaves amax to phase 2.
    if (tid == 0 && atomicAdd(f.amax + 1, 1u) == gridDim.x - 1) {
        if (f.phase != 1) f.amax[0] = 0u;
        f.amax[1] = 0u;
        if (tail_count > 0) *f.work_counter = 0u;
    }
    PGH_STAMP_END(g_times_finish)
}

PbView pb_view(const BsfFormat& f, const PbFormat& p) {
    PbView v;
    v.sloc = p.sloc;
    v.val = p.val;
    v.dstg = p.dstg;
    v.task = p.task;
    v.task_range = p.task_range;
    v.first_task = p.first_task;
    v.tmp = p.tmp;
    v.tmp_planes = p.tmp_planes;
    static const int short_env = getenv("PGH_GATHER_SHORT") != nullptr ? atoi(getenv("PGH_GATHER_SHORT")) : -1;
    v.short_piece = short_env >= 0 ? short_env : p.short_piece;
    v.item_a = p.item_a;
    v.item_b = p.item_b;
    v.num_items = p.num_items;
    v.sched = p.sched;
    v.sched_begin = p.sched_begin;
    v.first_a = p.first_a;
    v.first_b = p.first_b;
    v.first_item = p.first_item;
    v.work_counter = p.work_counter;
    v.tail_begin = p.tail_begin;
    v.tail_count = p.tail_count;
    v.hub_ticket = p.hub_ticket;
    v.drow = p.drow;
    v.amax = p.amax;
    v.iso_flag = f.iso_flag;
    v.phase = 0;
    v.phase_blk = f.blk_size > 0 ? f.blk_size : 1;
    v.phase_live = 0;
    v.hub_part = p.hub_part;
    for (int i = 0; i < 9; ++i) v.cold_prefix[i] = p.cold_prefix[i];
    for (int i = 0; i < 8; ++i) v.xg_base[i] = f.xg_base_cold[i];      // phase A reads slots >= hot only
    v.num_blocks = f.num_blocks;
    v.hot = p.hot;
    v.chunk = p.chunk;
    v.num_chunks = p.num_chunks;
    v.num_bins = p.num_bins;
    v.num_cold = p.cold_prefix[f.num_blocks];
    return v;
}

}  // namespace

// Decides whether the cold tail gets its own image and, if so, lays out the bins.  keys: the sorted stream
// (block << 58 | row << 29 | col), is_hot: 1 = stays in the stream; on success entries of rows too heavy for a bin are
// re-flagged as staying.  plan->row_bin is a device array and plan->host_bins a host array, both freed by pb_plan_release.
int pb_plan(BsfFormat& f, const uint64_t* keys, int64_t E, const int* live, int hot, unsigned char* is_hot, PbPlan* plan, bool* use) {
    *use = false;
    // PGH_PB=0 switches the image off; PGH_PB_FORCE=1 lifts the size heuristics below (tests).
    const char* env = getenv("PGH_PB");
    if (env != nullptr && atoi(env) == 0) return 0;
    Runtime& r = rt();
    const int chunk = f.pb_chunk > 0 ? f.pb_chunk : kPbChunk;         // (the f64 image: 16 K doubles fill the LDS of phase A)
    int64_t cold_sources = 0;
    for (int b = 0; b < f.num_blocks; ++b) cold_sources += live[b] > hot ? live[b] - hot : 0;
    plan->dense_prefix[0] = 0;
    for (int b = 0; b < 8; ++b) plan->dense_prefix[b + 1] = plan->dense_prefix[b] + (b < f.num_blocks && live[b] > hot ? live[b] - hot : 0);
    for (int b = 0; b < 9; ++b) plan->compact_prefix[b] = plan->dense_prefix[b];
    // ---- need lists (partition slices): only the cold sources this slice REFERENCES get a cold id -- a rank of an 8-way partition
    // references ~43 % of the live slots: fewer chunks, longer (chunk, bin) runs, and an exchange of those slots alone
    PbBuf<uint32_t> mark;
    if (f.want_compact && cold_sources > 0 && cold_sources < (1LL << 31)) {
        PbLayout L{};
        for (int b = 0; b < 9; ++b) L.cold_prefix[b] = plan->dense_prefix[b];
        L.blk = f.blk_size;
        L.hot = hot;
        L.chunk = chunk;
        PGH_TRY(mark.alloc(cold_sources + 1, true));
        PGH_HIP(pooled_malloc(&plan->cold_rank, sizeof(uint32_t) * (size_t)(cold_sources + 1)));
        k_pb_mark_cold<<<pb_blocks_for(E), kBlock, 0, r.stream>>>(keys, is_hot, E, L, mark.p);
        PGH_HIP(hipGetLastError());
        size_t scan_bytes = 0;
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, mark.p, plan->cold_rank, (int)(cold_sources + 1), r.stream));
        PbBuf<char> scan_temp;
        PGH_TRY(scan_temp.alloc(scan_bytes));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, mark.p, plan->cold_rank, (int)(cold_sources + 1), r.stream));
        uint32_t at[9] = {0};
        for (int b = 0; b <= 8; ++b)
            PGH_HIP(hipMemcpyAsync(&at[b], plan->cold_rank + plan->dense_prefix[b], sizeof(uint32_t), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        for (int b = 0; b <= 8; ++b) plan->compact_prefix[b] = (int64_t)at[b];
    }
    const int64_t cold_ids = plan->compact_prefix[8];             // == cold_sources without need lists
    const int64_t chunks = (cold_ids + chunk - 1) / chunk;
    if (chunks < 1 || chunks > kPbMaxChunks || f.n_out >= (1 << 28)) {
        (void)pooled_free(plan->cold_rank);
        plan->cold_rank = nullptr;
        return 0;
    }
    PbBuf<uint32_t> d_counts;
    PGH_TRY(d_counts.alloc(f.n_out, true));
    k_pb_row_counts<<<pb_blocks_for(E), kBlock, 0, r.stream>>>(keys, is_hot, E, d_counts.p);
    // the counts as bytes + the list of rows with 255 or more (rounds 1-5 copied n words down and n words of row -> bin map up through
    // pageable memory: two thirds of the planner's 35 ms at scale 23)
    std::vector<unsigned char> small((size_t)f.n_out);
    std::vector<std::pair<uint32_t, uint32_t>> big;          // (row, count), ascending rows
    {
        const uint32_t cap = (uint32_t)std::max<int64_t>(1 << 16, f.n_out / 16);
        PbBuf<unsigned char> d_small;
        PbBuf<uint32_t> d_rows, d_cnts, d_n;
        PGH_TRY(d_small.alloc(f.n_out));
        PGH_TRY(d_rows.alloc(cap));
        PGH_TRY(d_cnts.alloc(cap));
        PGH_TRY(d_n.alloc(1, true));
        k_pb_counts_small<<<pb_blocks_for(f.n_out), kBlock, 0, r.stream>>>(d_counts.p, f.n_out, d_small.p, d_rows.p, d_cnts.p, d_n.p, cap);
        PGH_HIP(hipGetLastError());
        uint32_t big_n = 0;
        PGH_HIP(hipMemcpyAsync(small.data(), d_small.p, (size_t)f.n_out, hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipMemcpyAsync(&big_n, d_n.p, sizeof(uint32_t), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        PGH_CHECK(big_n <= cap, "propagation blocking: more rows with 255 or more cold entries than the planner lists");
        std::vector<uint32_t> rows(big_n), cnts(big_n);
        if (big_n > 0) {
            PGH_HIP(hipMemcpyAsync(rows.data(), d_rows.p, sizeof(uint32_t) * big_n, hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipMemcpyAsync(cnts.data(), d_cnts.p, sizeof(uint32_t) * big_n, hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
        }
        big.resize(big_n);
        for (uint32_t k = 0; k < big_n; ++k) big[k] = {rows[k], cnts[k]};
        std::sort(big.begin(), big.end());
    }
    std::vector<int32_t> unbinned;                            // rows whose cold entries stay in the blocked stream (row_bin = -1)
    // greedy bins: consecutive rows, <= bin_rows rows, <= kPbBinFill * bin_rows entries; a row above kPbHeavyRow entries
    // is a hub bin by itself (above kPbHubMax it gets none: its cold entries stay in the stream);
    // bins without entries are dropped (their rows never receive a cold contribution: `out` stays 0 there)
    std::vector<int4> bins;
    int64_t cold = 0, in_image = 0;
    bool heavy_rows = false;
    int heavy_row = kPbHeavyRow, hub_max = kPbHubMax;
    if (const char* v = getenv("PGH_PB_HEAVY")) heavy_row = std::max(1, std::min(atoi(v), kPbHeavyRow));
    if (const char* v = getenv("PGH_PB_HUBMAX")) hub_max = std::max(8, atoi(v));
    plan->heavy_row = heavy_row;
    std::vector<int4> split;               // {row, first bin, pieces, -} of every row with more than one hub bin
    auto lay_out = [&](int bin_rows, int bin_fill) {
        bins.clear();
        split.clear();
        cold = in_image = 0;
        heavy_rows = false;
        const int64_t bin_entries = (int64_t)bin_fill * bin_rows;
        unbinned.clear();
        size_t next_big = 0;
        int row0 = 0, rows = 0;
        int64_t fill = 0, largest = 0;     // cold entries of the open bin, and of its largest row
        auto close_bin = [&]() {
            if (rows > 0 && fill > 0) bins.push_back(make_int4(row0, rows, (int)largest, (int)fill));   // .y: rows (pieces - 1 = 0)
            rows = 0;
            fill = 0;
            largest = 0;
        };
        for (int i = 0; i < f.n_out; ++i) {
            int64_t c = small[(size_t)i];
            if (c == 255) c = big[next_big++].second;          // (the list ascends with the rows: every 255 is the next entry)
            cold += c;
            if (c > heavy_row) {                 // hub bins of its own
                const int64_t pieces = (c + hub_max - 1) / hub_max;
                close_bin();
                if (pieces > kPbMaxPieces || pieces > chunks) {       // its cold entries stay in the blocked stream
                    unbinned.push_back(i);
                    heavy_rows = true;
                    continue;
                }
                if (pieces > 1) split.push_back(make_int4(i, (int)bins.size(), (int)pieces, 0));
                for (int64_t k = 0; k < pieces; ++k)   // entries are dealt by source chunk: the fills are nominal (their sum is exact)
                    bins.push_back(make_int4(i, 1 | (int)((pieces - 1) << 22), (int)std::min<int64_t>(c, 1 << 30),
                                             (int)(c / pieces + (k == 0 ? c % pieces : 0))));
                in_image += c;
                continue;
            }
            if (rows > 0 && (fill + c > bin_entries || rows >= bin_rows)) close_bin();
            if (rows == 0) row0 = i;
            ++rows;                              // (row -> bin: k_pb_row_bin, from the bins' first rows)
            fill += c;
            largest = std::max<int64_t>(largest, c);
            in_image += c;
        }
        close_bin();
    };
    int bin_rows = kPbBinRows;
    if (const char* shape = getenv("PGH_PB_BINROWS"))     // diagnostic
        bin_rows = atoi(shape) > kPbBinRowsMid ? kPbBinRowsLarge : (atoi(shape) > kPbBinRows ? kPbBinRowsMid : kPbBinRows);
    const char* fill_env = getenv("PGH_PB_BINFILL");
    int bin_fill = fill_env != nullptr ? std::max(1, atoi(fill_env)) : kPbBinFill;
    auto mean_run = [&]() { return (double)in_image / ((double)chunks * (double)std::max<size_t>(bins.size(), 1)); };
    lay_out(bin_rows, bin_fill);
    // Fuller bins where the (chunk, bin) runs are short (round 4, profiles/r04/pb_large_graphs.log): twice the entries per bin = half
    // the bins = runs twice as long.  Same-box sweeps of the fill, GTEPS: scale 23 (146 entries per run at fill 6) 562 / 544 / 554 for
    // 6 / 9 / 12 -- stays 6; scale 24 (82) 510 / 512 / 517 / 517 for 6 / 9 / 12 / 16; scale 25 (46) 408 / 458 / 449 / 464 / 443 for
    // 6 / 9 / 12 / 16 / 24; the slices of the 2 / 4 / 8-GPU bench 265 -> 243, 292 -> 279, 421 -> 403 us per step for 6 -> 12.
    if (fill_env == nullptr && mean_run() < 100.0) {
        bin_fill = 2 * kPbBinFill;
        lay_out(bin_rows, bin_fill);
    }
    // measured (profiles/r01/partition_slices_pb.log, pb_large_graphs.log): at 55 entries per run the small shape is 3 %
    // faster (scale 24), at 31 it is 3 % faster on a partitioned slice but 7 % slower at scale 25, at 14 it does not pay
    // (the middle shape first: whole graphs only -- the slices of a partition were measured with the two old shapes)
    if (mean_run() < 80.0 && kPbBinRowsMid > bin_rows && !f.want_compact && f.whole_graph && getenv("PGH_PB_BINROWS") == nullptr) {
        bin_rows = kPbBinRowsMid;
        lay_out(bin_rows, bin_fill);
    }
    // (the f64 image, whose runs hold 8-byte values: the large shape from 80 entries per run down -- scale 25: 73 entries with 8192-row bins,
    // 134 with 16 384: 2243 against 2384 us per term, 2344 with the cold gathers left in the stream)
    if (mean_run() < (f.pb64 ? 80.0 : 40.0) && kPbBinRowsLarge > bin_rows && getenv("PGH_PB_BINROWS") == nullptr) {
        bin_rows = kPbBinRowsLarge;              // short runs: fewer, larger bins
        lay_out(bin_rows, bin_fill);
    }
    plan->bin_rows = bin_rows;
    const int64_t num_bins = (int64_t)bins.size();
    auto no_image = [&]() {
        (void)pooled_free(plan->cold_rank);
        plan->cold_rank = nullptr;
        return 0;
    };
    if (num_bins < 1 || num_bins > kPbMaxBins || in_image + 8 * chunks * num_bins >= 2147483647LL) return no_image();
    const double run = (double)in_image / ((double)chunks * (double)num_bins);
    const char* force = getenv("PGH_PB_FORCE");
    const bool forced = force != nullptr && atoi(force) != 0;
    // Worth it when the saved L2 requests (~4.2 ps per cold entry on MI355X) outweigh ~12.5 streamed bytes per entry
    // (~2.9 ps) plus two launches, the chunk fills and a fifth vector in the combine, minus what k_bsf_partial gains from
    // a hot-only 16-bit stream: from ~10 M cold entries.  Measured (profiles/r01/pb_skew_scales.log): RMAT a=.57
    // scale 20 (5 M cold) -36 %, scale 21 (11 M) +7 %, scale 22 +12 %, scale 23 +31 %; flat degree distributions: scale 20
    // (10-17 M cold) -2 .. +11 %, scale 21 and up x1.5 .. x2.4.
    // Runs are padded to whole groups of 8 (3.5 pad entries on average): below ~10 entries per run the padding eats the
    // gain.
    // Valued graphs stream 4 more bytes per cold entry in phase A: 13 M cold entries measure -3 % (scale 21 upload).
    const int64_t least = (f.val != nullptr ? 16 : 10) * (1LL << 20);
    if (!forced && (in_image < least || in_image * 20 < E || run < 10.0)) return no_image();
    // the f64 image hands DOUBLES from A to B: its (chunk, bin) runs must be longer to pay -- measured against the cold gathers left in
    // the stream (profiles/r06/cheb_f64_cold_image.log): scale 22 (132 entries per run) 215 -> 179 us per term, scale 23 (111) 470 -> 381,
    // scale 24 (133) 1037 -> 939, scale 25 (73 with 8192-row bins) 2314 -> 2373, (134 with 16 384-row bins) -> 1961-2243, scale 26 (81 with
    // 16 384-row bins) 5346 -> 4850
    if (!forced && f.pb64 && run < 60.0) return no_image();
    // rows that keep their cold entries in the blocked stream read the DENSE cold slots from there: such a slice cannot number its cold
    // sources compactly (its exchange stays the all-gather); the plan is laid out again for the dense numbering
    if (plan->cold_rank != nullptr && heavy_rows) {
        (void)pooled_free(plan->cold_rank);
        plan->cold_rank = nullptr;
        for (int b = 0; b < 9; ++b) plan->compact_prefix[b] = plan->dense_prefix[b];
        f.want_compact = false;
        return pb_plan(f, keys, E, live, hot, is_hot, plan, use);
    }
    if (plan->cold_rank != nullptr) {
        // the lists themselves: kept with the graph (pgh_dist_need_list), block-major
        PbLayout L{};
        for (int b = 0; b < 9; ++b) L.cold_prefix[b] = plan->dense_prefix[b];
        (void)pooled_free(f.need_idx);
        f.need_idx = nullptr;
        PGH_HIP(pooled_malloc(&f.need_idx, sizeof(uint32_t) * (size_t)(cold_ids > 0 ? cold_ids : 1)));
        k_pb_need_idx<<<pb_blocks_for(cold_sources), kBlock, 0, r.stream>>>(mark.p, plan->cold_rank, cold_sources, L, f.num_blocks, f.need_idx);
        PGH_HIP(hipGetLastError());
        for (int b = 0; b < 9; ++b) f.need_prefix[b] = plan->compact_prefix[b];
        f.device_bytes += cold_ids * 4;
    }
    PGH_HIP(pooled_malloc(&plan->row_bin, sizeof(int32_t) * (size_t)f.n_out));
    {
        std::vector<int32_t> first_row((size_t)num_bins);
        for (int64_t w = 0; w < num_bins; ++w) first_row[(size_t)w] = bins[(size_t)w].x;
        PbBuf<int32_t> d_first, d_unbinned;
        PGH_TRY(d_first.alloc((size_t)num_bins));
        PGH_HIP(hipMemcpyAsync(d_first.p, first_row.data(), sizeof(int32_t) * (size_t)num_bins, hipMemcpyHostToDevice, r.stream));
        k_pb_row_bin<<<pb_blocks_for(f.n_out), kBlock, 0, r.stream>>>(d_first.p, (int)num_bins, f.n_out, plan->row_bin);
        if (!unbinned.empty()) {
            PGH_TRY(d_unbinned.alloc(unbinned.size()));
            PGH_HIP(hipMemcpyAsync(d_unbinned.p, unbinned.data(), sizeof(int32_t) * unbinned.size(), hipMemcpyHostToDevice, r.stream));
            k_pb_rows_unbinned<<<pb_blocks_for((int64_t)unbinned.size()), kBlock, 0, r.stream>>>(d_unbinned.p, (int)unbinned.size(), plan->row_bin);
        }
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipStreamSynchronize(r.stream));           // (the host vectors and device temporaries go out of scope)
    }
    k_pb_keep_heavy<<<pb_blocks_for(E), kBlock, 0, r.stream>>>(keys, E, plan->row_bin, is_hot);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    plan->num_bins = (int)num_bins;
    plan->num_chunks = (int)chunks;
    plan->entries = in_image;
    // slices: consecutive bins, about equal entry counts, each small enough for its values to stay cached between the phases
    {
        const char* sl = getenv("PGH_PB_SLICES");
        int want = sl != nullptr ? atoi(sl) : 1;            // measured: 4 / 8 slices lose more to launches and partial rounds than
                                                            // the cached hand-over wins (pb_experiment_scale23.log)
        want = 1;                              // k_pb_finish walks ONE work list that tiles all output rows
        if (!split.empty()) want = 1;          // the pieces of a row must not straddle slices
        plan->slices = want;
        plan->host_bins = new int4[bins.size()];
        std::copy(bins.begin(), bins.end(), plan->host_bins);
        int64_t acc = 0;
        int s_at = 0;
        plan->slice_first[0] = 0;
        for (int w = 0; w < (int)num_bins; ++w) {
            if (s_at + 1 < want && acc >= (in_image * (s_at + 1)) / want) plan->slice_first[++s_at] = w;
            acc += bins[w].w;
        }
        while (s_at < want) plan->slice_first[++s_at] = (int)num_bins;
        for (int g = 0; g < want; ++g) {
            plan->slice_entries[g] = 0;
            for (int w = plan->slice_first[g]; w < plan->slice_first[g + 1]; ++w) plan->slice_entries[g] += bins[w].w;
        }
    }
    plan->heavy_rows = heavy_rows && in_image < cold;
    plan->num_split = (int)split.size();
    if (!split.empty()) {
        plan->host_split = new int4[split.size()];
        std::copy(split.begin(), split.end(), plan->host_split);
    }
    *use = true;
    return 0;
}

// cold_keys: the entries of the image as stream keys (block << 58 | row << 29 | col), any order; cold_vals: values or null.
// Shares of the A-order stream for the workgroups of phase A (at most one per CU), balanced by cost = entries + a fixed price
// for every chunk image a share has to load (the tail chunks hold few entries: a share there crosses many of them).
// (Round 5 measured the shares instead -- workgroup clocks of calibration launches, the price of a slow share's stretch raised, planned
// again -- and made the launch slower every round: at ANY price the workgroups end 12 us apart, p10 to p90, and which ones end late
// follows the CU, not the share's content.  profiles/r05/gather_shares_calibration_rejected.log)
static void pb_plan_shares(const PbFormat& p, const int64_t* chunk_start, int64_t padded, int64_t fill_cost, int num_cus,
                           std::vector<int4>& tasks, std::vector<int>& ranges) {
    // every share that starts inside a chunk pays one more fill, so the total grows with the number of shares: the
    // smallest per-share budget that needs no more shares than there are CUs is searched for (a fixed 1.16 x mean left 9
    // of 256 CUs without a share at scale 23)
    auto build = [&](int64_t target) {
        tasks.clear();
        ranges.assign(1, 0);
        int64_t left = target;
        for (int c = 0; c < p.num_chunks; ++c) {
            int64_t lo = chunk_start[c];
            const int64_t hi = chunk_start[c + 1];
            while (lo < hi) {
                if (left < fill_cost + 4096 && (int)tasks.size() > ranges.back()) {
                    ranges.push_back((int)tasks.size());   // next share
                    left = target;
                }
                int64_t take = std::min<int64_t>(hi - lo, std::max<int64_t>(left - fill_cost, 4096));
                if (lo + take < hi) take = std::max<int64_t>(8, take & ~(int64_t)7);
                take = std::min<int64_t>(take, hi - lo);
                tasks.push_back(make_int4(c, (int)lo, (int)(lo + take), 0));
                lo += take;
                left -= fill_cost + take;
            }
        }
        ranges.push_back((int)tasks.size());
        return (int)ranges.size() - 1;
    };
    const double mean = (double)(padded + (int64_t)p.num_chunks * fill_cost) / (double)num_cus;
    double lo_m = 1.0, hi_m = 1.5;
    while (build((int64_t)(hi_m * mean) + 8) > num_cus) hi_m *= 1.25;
    for (int it = 0; it < 24; ++it) {
        const double mid = 0.5 * (lo_m + hi_m);
        if (build((int64_t)(mid * mean) + 8) > num_cus) lo_m = mid;
        else hi_m = mid;
    }
    (void)build((int64_t)(hi_m * mean) + 8);
}

int pb_build(BsfFormat& f, PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, const int* live,
             int hot) {
    Runtime& r = rt();
    PbFormat& p = slice == 0 ? f.pb : f.pb_more[slice - 1];
    p = PbFormat();
    PGH_CHECK(count == plan->slice_entries[slice], "propagation blocking: entry count does not match the plan");
    p.num_entries = count;
    p.chunk = f.pb_chunk > 0 ? f.pb_chunk : kPbChunk;
    p.f64 = f.pb64;
    p.hot = hot;
    p.k1_cold = plan->heavy_rows;
    p.bin_rows = plan->bin_rows;
    const int first_bin = plan->slice_first[slice];
    p.num_bins = plan->slice_first[slice + 1] - first_bin;
    // this slice's bins (the group ranges are filled in below)
    std::vector<int4> mine(plan->host_bins + first_bin, plan->host_bins + first_bin + p.num_bins);
    PGH_HIP(pooled_malloc(&p.bin, sizeof(int4) * (size_t)(p.num_bins > 0 ? p.num_bins : 1)));
    PGH_HIP(hipMemcpyAsync(p.bin, mine.data(), sizeof(int4) * mine.size(), hipMemcpyHostToDevice, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    for (int b = 0; b < 9; ++b) p.cold_prefix[b] = plan->compact_prefix[b];       // (== the dense prefix without need lists)
    p.num_chunks = plan->num_chunks;
    PbLayout L;
    for (int b = 0; b < 9; ++b) L.cold_prefix[b] = plan->dense_prefix[b];
    L.rank = plan->cold_rank;
    L.blk = f.blk_size;
    L.hot = hot;
    L.chunk = p.chunk;
    PbBuf<uint64_t> keys_a, keys_b;
    PbBuf<float> sorted_vals;
    PGH_TRY(keys_a.alloc(count));
    PGH_TRY(keys_b.alloc(count));
    k_pb_keys<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(cold_keys, count, L, plan->row_bin, first_bin, p.bin, keys_a.p);
    PGH_HIP(hipGetLastError());
    PbBuf<char> temp;
    size_t temp_bytes = 0, need = 0;
    PGH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_a.p, keys_b.p, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)count, 0, 58,
                                               r.stream));
    PGH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, need, keys_a.p, keys_b.p, (int)count, 0, 58, r.stream));
    temp_bytes = std::max(temp_bytes, need);
    PGH_TRY(temp.alloc(temp_bytes));
    if (cold_vals) {
        PGH_TRY(sorted_vals.alloc(count));
        PGH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.p, keys_b.p, cold_vals, sorted_vals.p, (int)count, 0, 58, r.stream));
    } else {
        PGH_HIP(hipcub::DeviceRadixSort::SortKeys(temp.p, temp_bytes, keys_a.p, keys_b.p, (int)count, 0, 58, r.stream));
    }
    // ---- keys_b: (chunk, bin, row, source) order.  Runs = cells, padded to whole groups of 8, laid out in both orders.
    const int64_t cells = (int64_t)p.num_chunks * p.num_bins;
    build_mark("cold image: build (cell keys, sort)");
    PbBuf<uint32_t> counts, first, pad_a, pad_b, start_a, start_b;
    PGH_TRY(counts.alloc(cells + 1, true));
    PGH_TRY(first.alloc(cells + 1));
    PGH_TRY(pad_a.alloc(cells + 1, true));
    PGH_TRY(pad_b.alloc(cells + 1, true));
    PGH_TRY(start_a.alloc(cells + 1));
    PGH_TRY(start_b.alloc(cells + 1));
    k_pb_cell_counts<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(keys_b.p, count, p.num_bins, counts.p);
    PGH_HIP(hipGetLastError());
    k_pb_padded<<<pb_blocks_for(cells), kBlock, 0, r.stream>>>(counts.p, p.num_chunks, p.num_bins, pad_a.p, pad_b.p);
    PGH_HIP(hipGetLastError());
    {
        size_t scan_bytes = 0;
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, counts.p, first.p, (int)(cells + 1), r.stream));
        PbBuf<char> scan_temp;
        PGH_TRY(scan_temp.alloc(scan_bytes));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, counts.p, first.p, (int)(cells + 1), r.stream));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, pad_a.p, start_a.p, (int)(cells + 1), r.stream));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, pad_b.p, start_b.p, (int)(cells + 1), r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    // starts of the chunks in A order, of the bins in B order, and the padded total
    // (one strided gather + ONE copy: rounds 1-5 issued a 4-byte copy per chunk and per bin -- 2 150 of them at scale 23, most of the
    // 2 189 copy launches of a bench process)
    std::vector<uint32_t> chunk_start(p.num_chunks + 1), bin_start(p.num_bins + 1);
    {
        PbBuf<uint32_t> picked;
        const int n_a = p.num_chunks + 1, n_b = p.num_bins + 1;
        PGH_TRY(picked.alloc((size_t)(n_a + n_b)));
        k_pb_pick_starts<<<pb_blocks_for(n_a + n_b), kBlock, 0, r.stream>>>(start_a.p, p.num_bins, n_a, start_b.p, p.num_chunks, n_b, picked.p);
        PGH_HIP(hipGetLastError());
        std::vector<uint32_t> host((size_t)(n_a + n_b));
        PGH_HIP(hipMemcpyAsync(host.data(), picked.p, sizeof(uint32_t) * host.size(), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        std::copy(host.begin(), host.begin() + n_a, chunk_start.begin());
        std::copy(host.begin() + n_a, host.end(), bin_start.begin());
    }
    const int64_t padded = chunk_start[p.num_chunks];
    PGH_CHECK(padded == (int64_t)bin_start[p.num_bins] && padded < 2147483647LL, "propagation blocking: layout totals disagree");
    for (int w = 0; w < p.num_bins; ++w) {
        int count_bits = 0;                                // 2^count_bits >= entries of the bin's largest row
        while ((1 << count_bits) < mine[w].z) ++count_bits;
        mine[w].y |= count_bits << 16;
        if (mine[w].z > plan->heavy_row) mine[w].y |= 1 << 21;    // hub bin
        mine[w].z = (int)(bin_start[w] >> 3);
        mine[w].w = (int)((bin_start[w + 1] - bin_start[w]) >> 3);
    }
    PGH_HIP(hipMemcpyAsync(p.bin, mine.data(), sizeof(int4) * mine.size(), hipMemcpyHostToDevice, r.stream));
    PGH_HIP(pooled_malloc(&p.sloc, sizeof(uint16_t) * (size_t)(padded + 8)));
    PGH_HIP(pooled_malloc(&p.drow, sizeof(uint16_t) * (size_t)(padded + 8)));
    PGH_HIP(pooled_malloc(&p.dstg, sizeof(uint32_t) * (size_t)(padded / 8 + 1)));
    PGH_HIP(hipMemsetAsync(p.sloc, 0, sizeof(uint16_t) * (size_t)(padded + 8), r.stream));
    PGH_HIP(hipMemsetAsync(p.drow, 0xff, sizeof(uint16_t) * (size_t)(padded + 8), r.stream));
    PGH_HIP(hipMemsetAsync(p.dstg, 0, sizeof(uint32_t) * (size_t)(padded / 8 + 1), r.stream));
    if (cold_vals) {
        PGH_HIP(pooled_malloc(&p.val, sizeof(float) * (size_t)(padded + 8)));
        PGH_HIP(hipMemsetAsync(p.val, 0, sizeof(float) * (size_t)(padded + 8), r.stream));
    }
    k_pb_place<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(keys_b.p, cold_vals ? sorted_vals.p : nullptr, count, p.num_chunks, p.num_bins, first.p,
                                                               start_a.p, start_b.p, p.sloc, p.val, p.dstg, p.drow);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    build_mark("cold image: build (cells, placement)");
    // ---- phase A shares (positions in A order; chunks and pieces are whole groups of 8)
    std::vector<int4> tasks;
    std::vector<int> ranges(1, 0);
    {
        // (thin chunks -- fewer than 160 K entries each: the 4- and 8-way slices of the N-GPU bench, 127 K and 61 K -- keep 24 K: every
        // share crosses several chunks there whatever the price, and the higher one only unbalances the entries: phase A 106 against
        // 113 us and 153 against 166-189)
        const int64_t fill_cost = getenv("PGH_PB_FILLCOST") != nullptr ? atoll(getenv("PGH_PB_FILLCOST"))
                                                                        : (padded / std::max(p.num_chunks, 1) < 163840 ? 24576 : 65536);
        std::vector<int64_t> starts(chunk_start.begin(), chunk_start.end());
        pb_plan_shares(p, starts.data(), padded, fill_cost, r.num_cus, tasks, ranges);
    }
    const int shares = (int)ranges.size() - 1;
    p.num_tasks = shares;
    if (getenv("PGH_DEBUG_SHARES") != nullptr) {           // diagnostic: what every share of phase A holds (to set against its time)
        FILE* out = fopen(getenv("PGH_DEBUG_SHARES"), "w");
        if (out != nullptr) {
            fprintf(out, "share,pieces,entries,first_chunk,last_chunk\n");
            for (int b = 0; b < shares; ++b) {
                int64_t e = 0;
                for (int t = ranges[b]; t < ranges[b + 1]; ++t) e += tasks[t].z - tasks[t].y;
                fprintf(out, "%d,%d,%lld,%d,%d\n", b, ranges[b + 1] - ranges[b], (long long)e, tasks[ranges[b]].x, tasks[ranges[b + 1] - 1].x);
            }
            fclose(out);
        }
    }
    p.avg_piece = tasks.empty() ? 0 : padded / (int64_t)tasks.size();
    if (getenv("PGH_DEBUG") != nullptr && atoi(getenv("PGH_DEBUG")) != 0) {
        int64_t mn = padded, mx = 0;
        int most = 0;
        for (const int4& t : tasks) {
            mn = std::min<int64_t>(mn, t.z - t.y);
            mx = std::max<int64_t>(mx, t.z - t.y);
        }
        for (int b = 0; b < shares; ++b) most = std::max(most, ranges[b + 1] - ranges[b]);
        fprintf(stderr, "[pgh] pb: %lld entries (%lld padded), %d chunks, %d bins, phase A: %zu pieces over %d shares (piece %lld..%lld entries, <= %d per share)\n",
                (long long)count, (long long)padded, p.num_chunks, p.num_bins, tasks.size(), shares, (long long)mn, (long long)mx, most);
    }
    PGH_HIP(pooled_malloc(&p.task, sizeof(int4) * (size_t)(tasks.size() + 1)));
    PGH_HIP(pooled_malloc(&p.task_range, sizeof(int) * (size_t)(shares + 1)));
    if (!tasks.empty()) PGH_HIP(hipMemcpyAsync(p.task, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice, r.stream));
    PGH_HIP(hipMemcpyAsync(p.task_range, ranges.data(), sizeof(int) * (shares + 1), hipMemcpyHostToDevice, r.stream));
    {
        std::vector<int4> first(shares > 0 ? shares : 1, make_int4(0, 0, 0, 0));
        for (int w = 0; w < shares; ++w)
            if (ranges[w] < ranges[w + 1]) first[w] = tasks[ranges[w]];
        PGH_HIP(pooled_malloc(&p.first_task, sizeof(int4) * first.size()));
        PGH_HIP(hipMemcpyAsync(p.first_task, first.data(), sizeof(int4) * first.size(), hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    if (p.f64) {
        PGH_HIP(pooled_malloc(&p.tmp64, sizeof(double) * (size_t)(((padded / 8 + 63) / 64 + 1) * 512)));     // whole blocks of 64 groups (pb64_pair)
        PGH_HIP(pooled_malloc(&p.amax64, sizeof(unsigned long long) * 2));
        PGH_HIP(hipMemsetAsync(p.amax64, 0, sizeof(unsigned long long) * 2, r.stream));
    } else
    PGH_HIP(pooled_malloc(&p.tmp, sizeof(float) * (size_t)(((padded / 8 + 63) / 64 + 1) * 512)));      // whole blocks of 64 groups (pb_tmp_quad)
    {
        // (chunk, bin) runs of a dozen groups or more: the two-plane layout of tmp; shorter runs: quads side by side
        const double per_cell = (double)padded / ((double)std::max(p.num_chunks, 1) * (double)std::max(p.num_bins, 1));
        p.tmp_planes = per_cell >= 96.0 ? 1 : 0;
        if (getenv("PGH_PB_PLANES") != nullptr) p.tmp_planes = atoi(getenv("PGH_PB_PLANES")) != 0 ? 1 : 0;
    }
    PGH_HIP(pooled_malloc(&p.amax, sizeof(uint32_t) * 2));
    PGH_HIP(hipMemsetAsync(p.amax, 0, sizeof(uint32_t) * 2, r.stream));
    // ---- work list of k_pb_finish: the bins in row order (`mine` is sorted by first row; the pieces of a split hub row are
    // consecutive), the row stretches between them cut into epilogue-only items, hub items first (they are the longest).
    {
        std::vector<int4> hub_a, hub_b, reg_a, reg_b, iso_a, iso_b;
        // rows per epilogue-only item (an item costs ~6 us + 4.5 us per 1000 rows: 512 / 1024 / 2048 / 4096 rows measured 96.9 / 89.5 /
        // 87.6 / 91.8 us per finish launch at scale 23, 192.8 / 188.3 / 185.9 / 184.7 at scale 24)
        int stretch = p.bin_rows > 2048 ? 2048 : p.bin_rows;
        if (getenv("PGH_FIN_STRETCH") != nullptr) stretch = std::max(64, std::min(p.bin_rows, atoi(getenv("PGH_FIN_STRETCH"))));
        // rows [lo, hi) have no cold entries in the image.  The part that lies in the isolated tail of its block (BsfFormat::
        // iso_begin: rows without any entry that nobody references) goes into items of its own, marked -2: k_pb_finish passes
        // over them unless the run's operands are non-zero there (BsfFormat::iso_flag).
        auto cover_plain = [&](int64_t lo, int64_t hi) {
            for (int64_t at = lo; at < hi; at += stretch) {
                reg_a.push_back(make_int4((int)at, 0, 0, 0));
                reg_b.push_back(make_int4((int)at, (int)std::min<int64_t>(stretch, hi - at), -1, 0));
            }
        };
        auto cover = [&](int64_t lo, int64_t hi) {
            if (!f.has_iso || f.blk_size <= 0) {
                cover_plain(lo, hi);
                return;
            }
            while (lo < hi) {
                const int64_t blk = lo / f.blk_size;
                const int64_t blk_end = std::min<int64_t>(hi, (blk + 1) * (int64_t)f.blk_size);
                const int64_t iso_at = blk * (int64_t)f.blk_size + (blk < f.iso_row_blocks ? f.iso_begin[blk] : f.blk_size);
                const int64_t mid = std::min(std::max(lo, iso_at), blk_end);
                cover_plain(lo, mid);
                constexpr int64_t kIsoRows = 8192;            // streamed by the kernel: not bound by the LDS row map
                for (int64_t at = mid; at < blk_end; at += kIsoRows) {
                    iso_a.push_back(make_int4((int)at, 0, 0, 0));
                    iso_b.push_back(make_int4((int)at, (int)std::min<int64_t>(kIsoRows, blk_end - at), -2, 0));
                }
                lo = blk_end;
            }
        };
        int64_t cursor = 0;
        int split_at = 0;
        for (int w = 0; w < p.num_bins;) {
            const int4 b = mine[w];
            const int row0 = b.x, rows = b.y & 0xffff;
            const bool hub = ((b.y >> 21) & 1) != 0;
            const int pieces = (int)((unsigned)b.y >> 22) + 1;
            if (row0 > cursor) cover(cursor, row0);
            if (hub) {
                const int first_item = (int)hub_a.size();
                for (int k = 0; k < pieces; ++k) {
                    hub_a.push_back(mine[w + k]);
                    hub_b.push_back(make_int4(row0, 1, pieces > 1 ? split_at : -1, first_item));
                }
                if (pieces > 1) ++split_at;
                w += pieces;
                cursor = (int64_t)row0 + 1;
            } else {
                reg_a.push_back(b);
                reg_b.push_back(make_int4(row0, rows, -1, 0));
                ++w;
                cursor = (int64_t)row0 + rows;
            }
        }
        if (cursor < f.n_out) cover(cursor, f.n_out);
        PGH_CHECK(split_at == plan->num_split, "propagation blocking: split rows of the work list do not match the plan");
        p.num_split = split_at;
        // hub pieces first (the longest items), then the isolated stretches (usually passed over: dealt evenly, early), then
        // the bins and plain stretches in row order (the last of them form the dynamic tail)
        p.num_items = (int)(hub_a.size() + iso_a.size() + reg_a.size());
        std::vector<int4> all_a(hub_a), all_b(hub_b);
        all_a.insert(all_a.end(), iso_a.begin(), iso_a.end());
        all_b.insert(all_b.end(), iso_b.begin(), iso_b.end());
        all_a.insert(all_a.end(), reg_a.begin(), reg_a.end());
        all_b.insert(all_b.end(), reg_b.begin(), reg_b.end());
        PGH_HIP(pooled_malloc(&p.item_a, sizeof(int4) * (size_t)(p.num_items + 1)));
        PGH_HIP(pooled_malloc(&p.item_b, sizeof(int4) * (size_t)(p.num_items + 1)));
        PGH_HIP(hipMemcpyAsync(p.item_a, all_a.data(), sizeof(int4) * all_a.size(), hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipMemcpyAsync(p.item_b, all_b.data(), sizeof(int4) * all_b.size(), hipMemcpyHostToDevice, r.stream));
        if (p.num_split > 0) {                             // piece sums are indexed by (hub) item, tickets by split row
            PGH_HIP(pooled_malloc(&p.hub_part, sizeof(double) * hub_a.size()));
            PGH_HIP(hipMemsetAsync(p.hub_part, 0, sizeof(double) * hub_a.size(), r.stream));
            PGH_HIP(pooled_malloc(&p.hub_ticket, sizeof(uint32_t) * (size_t)p.num_split));
            PGH_HIP(hipMemsetAsync(p.hub_ticket, 0, sizeof(uint32_t) * (size_t)p.num_split, r.stream));
        }
        // ---- schedule: persistent grid = what the CUs hold at once.  The head of the item list is dealt round-robin in row
        // order (item i -> workgroup i % groups: neighbouring workgroups touch neighbouring rows), the tail -- PGH_FIN_TAIL
        // percent of the items, never a hub piece -- is handed out on the device (k_pb_finish).
        {
            const bool large = p.bin_rows > kPbBinRowsMid, mid = p.bin_rows == kPbBinRowsMid;
            const int by_regs = PGH_FIN_WPE * 4 / (kPbBThreads / 64);
            int groups = r.num_cus * (large ? 1 : (mid ? 2 : (by_regs < 4 ? (by_regs < 1 ? 1 : by_regs) : 4)));
            if (groups > p.num_items) groups = p.num_items;
            if (groups > kMaxPartials) groups = kMaxPartials;
            if (groups < 1) groups = 1;
            const char* tail_env = getenv("PGH_FIN_TAIL");
            // measured at scale 23 (profiles/r02/finish_tail_sweep.log): 0 % 105.7 us, 8 % 100.7, 16 % 103.3, 24 % 103.6, 32 % 105.7,
            // 48 % 110.1; workgroup end times with 16 %: max 97 us instead of 105 (finish_tail_times.log)
            const int tail_pct = tail_env != nullptr ? atoi(tail_env) : 12;
            const int plain_items = (int)reg_a.size();         // neither hub pieces nor isolated stretches
            int tail = (int)((int64_t)plain_items * (tail_pct < 0 ? 0 : (tail_pct > 90 ? 90 : tail_pct)) / 100);
            if (tail > plain_items) tail = plain_items;
            if (tail > kMaxPartials - groups) tail = kMaxPartials - groups;
            if (tail < 0) tail = 0;
            const int head = p.num_items - tail;
            std::vector<double> cost(p.num_items);
            std::vector<int> flat, begin(1, 0);
            flat.reserve(p.num_items);
            // The deal counts WORK, not list positions: an isolated stretch that is passed over costs ~3 us, every other item 10-35 us
            // (profiles/r03/finish_schedule_r03.log), so the stretches are dealt on their own: round-robin over the items that do work,
            // in row order (at any moment the workgroups sweep the same stretch of the image), then the isolated stretches from the
            // other end.  Dealing them as equals (rounds 2-3) left workgroups with 22 .. 95 us of static work around a mean of 53.
            // Longest first by the measured cost model (whole head, or inside windows of 1024 items; with the row-order tail or with the
            // cheapest items as the tail), a snake over the rounds: all measured, all slower on at least two of the scales 22-24 -- what
            // they give up is the common sweep over the image (same log).
            // measured cost of an item (us; PGH_PROBE_TIMES build at scale 23, profiles/r03/finish_schedule_r03.log): 6.1 + 0.30 per 1000
            // entries + 4.5 per 1000 rows + 3.9 when it has entries at all; an isolated stretch that is passed over 3.3
            for (int i = 0; i < p.num_items; ++i) {
                const double entries = 8.0 * (double)all_a[i].w, rows = (double)all_b[i].y;
                cost[i] = all_b[i].z == -2 ? 3300.0 : 6100.0 + 0.30 * entries + 4.5 * rows + (entries > 0 ? 3900.0 : 0.0);
            }
            {
                std::vector<std::vector<int>> mine_of(groups);
                int k = 0;
                for (int i = 0; i < head; ++i)
                    if (all_b[i].z != -2) mine_of[k++ % groups].push_back(i);
                int back = 0;
                for (int i = 0; i < head; ++i)
                    if (all_b[i].z == -2) mine_of[groups - 1 - (back++ % groups)].push_back(i);
                for (int w = 0; w < groups; ++w) {
                    flat.insert(flat.end(), mine_of[w].begin(), mine_of[w].end());
                    begin.push_back((int)flat.size());
                }
            }
            {
                for (int i = head; i < p.num_items; ++i) flat.push_back(i);
                // A SHORT tail (fewer items than half the workgroups) is handed out longest first: the launch then ends on its cheapest
                // items (scale 23: 87.7 -> 84.1 us); a long one stays in row order, where the common sweep matters more (scale 24:
                // 198.9 -> 201.2 us when sorted; scales 22 and 25 do not care).  PGH_FIN_TAILSORT=0/1 forces either.
                const char* ts = getenv("PGH_FIN_TAILSORT");
                if (ts != nullptr ? atoi(ts) != 0 : 2 * tail < groups)
                    std::stable_sort(flat.begin() + head, flat.end(), [&](int x, int y) { return cost[x] > cost[y]; });
            }
            p.tail_begin = head;
            p.tail_count = tail;
            if (tail > 0) {
                PGH_HIP(pooled_malloc(&p.work_counter, sizeof(uint32_t)));
                PGH_HIP(hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t), r.stream));
            }
            p.sched_groups = groups;
            PGH_HIP(pooled_malloc(&p.sched, sizeof(int) * (size_t)(flat.size() + 1)));
            PGH_HIP(pooled_malloc(&p.sched_begin, sizeof(int) * (size_t)(groups + 1)));
            PGH_HIP(hipMemcpyAsync(p.sched, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice, r.stream));
            PGH_HIP(hipMemcpyAsync(p.sched_begin, begin.data(), sizeof(int) * begin.size(), hipMemcpyHostToDevice, r.stream));
            std::vector<int4> first_a(groups, make_int4(0, 0, 0, 0)), first_b(groups, make_int4(0, 0, -1, 0));
            std::vector<int> first_item(groups, -1);
            for (int w = 0; w < groups; ++w)
                if (begin[w] < begin[w + 1]) first_item[w] = flat[begin[w]], first_a[w] = all_a[flat[begin[w]]], first_b[w] = all_b[flat[begin[w]]];
            PGH_HIP(pooled_malloc(&p.first_a, sizeof(int4) * (size_t)groups));
            PGH_HIP(pooled_malloc(&p.first_b, sizeof(int4) * (size_t)groups));
            PGH_HIP(pooled_malloc(&p.first_item, sizeof(int) * (size_t)groups));
            PGH_HIP(hipMemcpyAsync(p.first_a, first_a.data(), sizeof(int4) * (size_t)groups, hipMemcpyHostToDevice, r.stream));
            PGH_HIP(hipMemcpyAsync(p.first_b, first_b.data(), sizeof(int4) * (size_t)groups, hipMemcpyHostToDevice, r.stream));
            PGH_HIP(hipMemcpyAsync(p.first_item, first_item.data(), sizeof(int) * (size_t)groups, hipMemcpyHostToDevice, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
            if (getenv("PGH_DEBUG") != nullptr && atoi(getenv("PGH_DEBUG")) != 0) {
                double lo = 1e300, hi = 0, total = 0;
                for (int w = 0; w + 1 < (int)begin.size(); ++w) {
                    double load = 0;
                    for (int k = begin[w]; k < begin[w + 1]; ++k) load += cost[flat[k]];
                    lo = std::min(lo, load), hi = std::max(hi, load), total += load;
                }
                fprintf(stderr, "[pgh] pb finish schedule: %d items over %d workgroups, estimated load min %.1f mean %.1f max %.1f us\n",
                        p.num_items, groups, lo * 1e-3, total / groups * 1e-3, hi * 1e-3);
            }
        }
        PGH_HIP(hipStreamSynchronize(r.stream));           // the host vectors go out of scope
    }
    p.device_bytes = padded * ((p.f64 ? 8 : 4) + 2 + 2 + (cold_vals ? 4 : 0)) + padded / 2 + (int64_t)p.num_items * 32;
    p.enabled = true;
    return 0;
}

void pb_plan_release(PbPlan* plan) {
    (void)pooled_free(plan->row_bin);
    (void)pooled_free(plan->cold_rank);
    plan->cold_rank = nullptr;
    delete[] plan->host_bins;
    delete[] plan->host_split;
    plan->host_split = nullptr;
    plan->row_bin = nullptr;
    plan->host_bins = nullptr;
}

namespace {
struct InSlice {
    const int32_t* row_bin;
    int lo, hi;
    __device__ bool operator()(const uint64_t& key) const {
        const int w = row_bin[(key >> 29) & kLow29];
        return w >= lo && w < hi;
    }
};
__global__ void k_pb_slice_flags(const uint64_t* __restrict__ keys, int64_t count, InSlice pred, unsigned char* __restrict__ flag) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) flag[i] = pred(keys[i]) ? 1 : 0;
}
}  // namespace

int pb_select_slice(const PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, uint64_t* keys_out,
                    float* vals_out, int64_t* selected) {
    Runtime& r = rt();
    PbBuf<unsigned char> flag;
    PbBuf<int64_t> num;
    PGH_TRY(flag.alloc(count));
    PGH_TRY(num.alloc(1));
    const InSlice pred{plan->row_bin, plan->slice_first[slice], plan->slice_first[slice + 1]};
    k_pb_slice_flags<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(cold_keys, count, pred, flag.p);
    size_t temp_bytes = 0;
    PGH_HIP(hipcub::DeviceSelect::Flagged(nullptr, temp_bytes, cold_keys, flag.p, keys_out, num.p, (int)count, r.stream));
    PbBuf<char> temp;
    PGH_TRY(temp.alloc(temp_bytes));
    PGH_HIP(hipcub::DeviceSelect::Flagged(temp.p, temp_bytes, cold_keys, flag.p, keys_out, num.p, (int)count, r.stream));
    if (cold_vals != nullptr) PGH_HIP(hipcub::DeviceSelect::Flagged(temp.p, temp_bytes, cold_vals, flag.p, vals_out, num.p, (int)count, r.stream));
    PGH_HIP(hipMemcpyAsync(selected, num.p, sizeof(int64_t), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    return 0;
}

// phase A of the cold image (after the block partial sums of the same step; independent of them)
int pb_launch_gather(pgh_graph_s* g, const float* xg, const LoopState* state, const FixView& fix) {
    const BsfFormat& f = g->bsf;
    if (!f.pb.enabled) return 0;
    Runtime& r = rt();
    const PbFormat& p = f.pb;
    const PbView v = pb_view(f, p);
    {
        ProfScope prof(PGH_K_PB_GATHER);
        if (p.num_tasks > 0) {
            // (diagnostic of round 3: launched with a quarter of its workgroups this kernel takes 63 us instead of 76 -- a workgroup's
            // share costs the same alone on the chip as with all others running: the per-CU memory path bounds it, ~21 GB/s)
            const DropView dv = bsf_dropout_view(p.drop_edge);
            if (dv.edge != nullptr) {                  // graph_dropout: the mask factor per cold entry
                if (p.val) k_pb_gather<true, PGH_GATHER_P, true><<<p.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state, fix, dv);
                else k_pb_gather<false, PGH_GATHER_P, true><<<p.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state, fix, dv);
            } else if (p.val) k_pb_gather<true, PGH_GATHER_P><<<p.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state, fix);
            else k_pb_gather<false, PGH_GATHER_P><<<p.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state, fix);
        }
    }
    PGH_STAMP_DUMP(g_times_gather, p.num_tasks, "k_pb_gather")
    PGH_HIP(hipGetLastError());
    return 0;
}

// the in-kernel residual of the next PageRank finish launch (pb_set_residual; the loop driver sets it per step)
namespace {
ResParams g_residual = {};
bool      g_residual_set = false;
}  // namespace
void pb_set_residual(const ResParams* rp) {
    g_residual_set = rp != nullptr;
    if (rp != nullptr) g_residual = *rp;
}
namespace {
int g_finish_phase = 0, g_finish_live = 0;
}  // namespace
void pb_set_finish_phase(int phase, int live) {
    g_finish_phase = phase;
    g_finish_live = live;
}
int pb_pending_finish_phase() { return g_finish_phase; }

// phase B + the MODE epilogue for every output row; block partials of sum(y) / delta land in rt().d_partials
template <int MODE>
int pb_launch_finish(pgh_graph_s* g, const RowSums& rs, const EpiParams& ep, const LoopState* state, int* num_partials) {
    const BsfFormat& f = g->bsf;
    const PbFormat& p = f.pb;
    Runtime& r = rt();
    PbView v = pb_view(f, p);
    v.phase = g_finish_phase;
    v.phase_live = g_finish_live;
    const int phase = g_finish_phase;
    g_finish_phase = 0;                                // consumed
    double* psum = r.d_partials;
    double* pdel = r.d_partials + kMaxPartials;
    const bool large = p.bin_rows > kPbBinRowsMid, mid = p.bin_rows == kPbBinRowsMid;
    const int grid = p.sched_groups;                   // persistent: as many workgroups as the CUs hold at once
    PGH_CHECK(phase == 0 || 2 * (grid + p.tail_count) <= kMaxPartials, "finish kernel: too many partial sums for the two-launch form");
    ResParams rp = g_residual;
    const bool res = MODE == EPI_AXPBY && g_residual_set;
    if (phase != 1) g_residual_set = false;            // (phase 2 of a two-launch finish evaluates the residual of its rows as well)
    {
        ProfScope prof(PGH_K_PB_ACCUM);
        const bool wide = f.num_blocks > 4;               // 8 column blocks: 8-way row partitions
#define PGH_FINISH(NBLK, ROWS_, THREADS_, RES_) \
    k_pb_finish<MODE, NBLK, ROWS_, THREADS_, RES_><<<grid, THREADS_, 0, r.stream>>>(v, rs, f.dst_scale, ep, state, psum, pdel, rp)
        if constexpr (MODE == EPI_AXPBY) {
            if (res) {
                if (large && wide) PGH_FINISH(8, kPbBinRowsLarge, kPbBThreadsLarge, true);
                else if (large) PGH_FINISH(4, kPbBinRowsLarge, kPbBThreadsLarge, true);
                else if (mid && wide) PGH_FINISH(8, kPbBinRowsMid, kPbBThreadsMid, true);
                else if (mid) PGH_FINISH(4, kPbBinRowsMid, kPbBThreadsMid, true);
                else if (wide) PGH_FINISH(8, kPbBinRows, kPbBThreads, true);
                else PGH_FINISH(4, kPbBinRows, kPbBThreads, true);
            }
        }
        if (!res) {
            if (large && wide) PGH_FINISH(8, kPbBinRowsLarge, kPbBThreadsLarge, false);
            else if (large) PGH_FINISH(4, kPbBinRowsLarge, kPbBThreadsLarge, false);
            else if (mid && wide) PGH_FINISH(8, kPbBinRowsMid, kPbBThreadsMid, false);
            else if (mid) PGH_FINISH(4, kPbBinRowsMid, kPbBThreadsMid, false);
            else if (wide) PGH_FINISH(8, kPbBinRows, kPbBThreads, false);
            else PGH_FINISH(4, kPbBinRows, kPbBThreads, false);
        }
#undef PGH_FINISH
    }
    PGH_HIP(hipGetLastError());
    PGH_STAMP_DUMP(g_times_finish, grid, "k_pb_finish")
#if PGH_PROBE_TIMES
    if (getenv("PGH_DUMP_PHASES") != nullptr) {            // where the workgroups of the 10th launch spent their time
        static int dumped_ph = 0;
        if (dumped_ph++ == 9) {
            (void)hipStreamSynchronize(r.stream);
            std::vector<unsigned int> ph(8 * 4096);
            (void)hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_phase_ticks), sizeof(unsigned int) * 8 * 4096);
            const int n_ = grid < 4096 ? grid : 4096;
            double tot[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int w = 0; w < n_; ++w)
                for (int k = 0; k < 7; ++k) tot[k] += ph[8 * w + k] * 0.01 / n_;
            fprintf(stderr, "[pgh phases] k_pb_finish: per workgroup (mean us): setup %.1f  stream %.1f  hand-over %.1f  epilogue %.1f  tail %.1f | items %.2f  item loop %.1f\n",
                    tot[0], tot[1], tot[2], tot[3], tot[4], tot[5] * 100.0, tot[6]);
        }
    }
    if (getenv("PGH_DUMP_ITEMS") != nullptr) {             // per-item durations of the 10th launch, with the item descriptors
        static int dumped = 0;
        if (dumped++ == 9) {
            (void)hipStreamSynchronize(r.stream);
            const int n_ = p.num_items < (1 << 15) ? p.num_items : (1 << 15);
            std::vector<unsigned int> ticks(1 << 15), begin(1 << 15);
            std::vector<int4> ia(p.num_items), ib(p.num_items);
            (void)hipMemcpyFromSymbol(ticks.data(), HIP_SYMBOL(g_item_ticks), sizeof(unsigned int) * (1 << 15));
            (void)hipMemcpyFromSymbol(begin.data(), HIP_SYMBOL(g_item_begin), sizeof(unsigned int) * (1 << 15));
            (void)hipMemcpy(ia.data(), p.item_a, sizeof(int4) * p.num_items, hipMemcpyDeviceToHost);
            (void)hipMemcpy(ib.data(), p.item_b, sizeof(int4) * p.num_items, hipMemcpyDeviceToHost);
            FILE* out = fopen(getenv("PGH_DUMP_ITEMS"), "w");
            if (out != nullptr) {
                fprintf(out, "item,rows,count_bits,hub,pieces,groups,epi_rows,begin_us,us\n");
                for (int i = 0; i < n_; ++i)
                    fprintf(out, "%d,%d,%d,%d,%d,%d,%d,%.2f,%.2f\n", i, ia[i].y & 0xffff, (ia[i].y >> 16) & 0x1f, (ia[i].y >> 21) & 1,
                            (int)((unsigned)ia[i].y >> 22) + 1, ia[i].w, ib[i].y, begin[i] * 0.01, ticks[i] * 0.01);
                fclose(out);
            }
        }
    }
#endif
    // one per workgroup, then one per tail item; the second launch of the two-launch form reports both launches' slots
    if (num_partials) *num_partials = (phase == 2 ? 2 : 1) * (grid + p.tail_count);
    return 0;
}
template int pb_launch_finish<EPI_PLAIN>(pgh_graph_s*, const RowSums&, const EpiParams&, const LoopState*, int*);
template int pb_launch_finish<EPI_AXPBY>(pgh_graph_s*, const RowSums&, const EpiParams&, const LoopState*, int*);
template int pb_launch_finish<EPI_ABSORB>(pgh_graph_s*, const RowSums&, const EpiParams&, const LoopState*, int*);
template int pb_launch_finish<EPI_POLY>(pgh_graph_s*, const RowSums&, const EpiParams&, const LoopState*, int*);

void pb_destroy(PbFormat& p) {
    (void)pooled_free(p.sloc);
    (void)pooled_free(p.val);
    (void)pooled_free(p.drop_edge);
    (void)pooled_free(p.task);
    (void)pooled_free(p.task_range);
    (void)pooled_free(p.first_task);
    (void)pooled_free(p.tmp);
    (void)pooled_free(p.tmp64);
    (void)pooled_free(p.amax64);
    (void)pooled_free(p.dstg);
    (void)pooled_free(p.amax);
    (void)pooled_free(p.split);
    (void)pooled_free(p.hub_part);
    (void)pooled_free(p.hub_ticket);
    (void)pooled_free(p.item_a);
    (void)pooled_free(p.item_b);
    (void)pooled_free(p.sched);
    (void)pooled_free(p.sched_begin);
    (void)pooled_free(p.first_a);
    (void)pooled_free(p.first_b);
    (void)pooled_free(p.first_item);
    (void)pooled_free(p.work_counter);
    (void)pooled_free(p.bin);
    (void)pooled_free(p.drow);
    p = PbFormat();
}

}  // namespace pgh

PGH_WARM_KERNEL(pgh::k_pb_pick_starts)
