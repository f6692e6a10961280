// Propagation blocking for the cold tail of the blocked stream.
//
// k_bsf_partial serves the gathers whose source sits in the block's LDS hot cache; every other ("cold") gather costs a
// 128-byte L1 line fill for 4 useful bytes, and on the bench graph those fills -- not HBM -- bound the step (DESIGN.md
// section 4).  This image removes them: the cold entries are taken out of the stream and processed by two streaming
// passes whose random accesses all land in LDS.
//
//   bins     runs of consecutive output rows holding <= kPbBinEntries cold entries (greedy, built on the host from the
//            per-row counts); a row heavier than a whole bin keeps its cold entries in the blocked stream
//   phase A  k_pb_gather      entries in (source chunk, bin, row, source) order.  Every workgroup takes a share of that
//                             stream: the chunk's slice of the gather vector goes to LDS (128 KB, coalesced), then
//                             tmp[e] = x_chunk[sloc[e]] (* val[e]): 2 B read + 4 B written per entry, sequential.
//   phase B  k_pb_accumulate  one workgroup per bin: the bin's runs (one per chunk, each contiguous in tmp) are staged
//                             in LDS; the bin's entries are then walked in ROW-MAJOR order through a 2-byte
//                             permutation -- every row is one segment -- with a lane-local f32 segmented sum, a DPP
//                             stitch and f64 carries like k_bsf_partial; the rows of the bin are written once,
//                             coalesced.  4 B + 2 B + 2 B read per entry, sequential.  Deterministic, no atomics.
//
// ~14 sequential bytes per cold entry instead of one line fill.  Runs must stay long enough to be worth a copy, which
// limits the image to graphs where cold_entries / (chunks * bins) >= ~24 (scale <= 24 on RMAT); beyond that the cold
// entries stay in k_bsf_partial.
//
// STATUS: experimental, opt-in with PGH_PB=1 (PGH_PB_FORCE=1 lifts the size heuristics for tests).  Results are identical
// to the default path to 1e-7; at scale 23 the three kernels take 136 + 68 + 181 us against 372 us for k_bsf_partial
// with the cold gathers left in: phase B is bound by DRAM-inefficient reads of ~250-byte runs (LDS capacity fixes the
// chunk and bin sizes, hence the run length).  profiles/r01/pb_experiment_scale23.log.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "pgh_kernels.h"

// diagnostic builds only (tools/build_variants.sh): 1 no chunk fill, 2 fills only, 4 no stores (phase A); 8 no staging,
// 16 no walk (phase B)
#ifndef PGH_PROBE_PB
#define PGH_PROBE_PB 0
#endif

namespace pgh {
namespace {

constexpr int kBlock = 256;
constexpr int kPbChunk = 32768;          // sources per chunk: 128 KB of LDS in phase A
#ifndef PGH_PB_BIN
#define PGH_PB_BIN 15360
#endif
#ifndef PGH_PB_BTHREADS
#define PGH_PB_BTHREADS 1024
#endif
constexpr int kPbBinEntries = PGH_PB_BIN;        // entries per bin staged in LDS by phase B
constexpr int kPbBThreads = PGH_PB_BTHREADS;     // phase B workgroup
constexpr int kPbBWaves = kPbBThreads / 64;
constexpr int kPbMaxChunks = 64 * kPbBWaves;     // run table of a bin in LDS: one descriptor pass
constexpr int kPbBinRows = kPbBinEntries >= 8192 ? 2048 : 1024;   // rows per bin (f32 row sums in LDS)
constexpr int kPbThreads = 1024;
constexpr uint64_t kLow29 = (1ULL << 29) - 1;

template <typename T>
struct PbBuf {
    T* p = nullptr;
    ~PbBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t count, bool zero = false) {
        PGH_HIP(hipMalloc(&p, sizeof(T) * (count > 0 ? count : 1)));
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
// cold entries per output row (stream keys: block << 58 | row << 29 | col)
__global__ void k_pb_row_counts(const uint64_t* __restrict__ keys, const unsigned char* __restrict__ is_hot, int64_t E,
                                uint32_t* __restrict__ row_cold) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        if (is_hot[e]) continue;
        const uint64_t row = (keys[e] >> 29) & kLow29;
        if (row != kLow29) atomicAdd(&row_cold[row], 1u);
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
    int64_t cold_prefix[9];
    int     blk, hot, chunk;
};

// stream key -> phase A key (chunk << 45 | bin << 27 | row_in_bin << 15 | source_in_chunk)
__global__ void k_pb_keys(const uint64_t* __restrict__ keys, int64_t count, PbLayout L, const int32_t* __restrict__ row_bin,
                          int first_bin, const int4* __restrict__ bin /* of this slice */, uint64_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        const int b = (int)(key >> 58);
        const int64_t row = (int64_t)((key >> 29) & kLow29);
        const int64_t loc = (int64_t)(key & kLow29) - (int64_t)b * L.blk;
        const int64_t cold_id = L.cold_prefix[b] + (loc - L.hot);
        const uint64_t c = (uint64_t)(cold_id / L.chunk), sl = (uint64_t)(cold_id % L.chunk);
        const uint64_t w = (uint64_t)(row_bin[row] - first_bin);
        const uint64_t dl = (uint64_t)(row - bin[w].x);
        out[i] = (c << 45) | (w << 27) | (dl << 15) | sl;
    }
}

// phase A order: source indices and the per-(chunk, bin) run lengths
__global__ void k_pb_split(const uint64_t* __restrict__ keys, int64_t count, int num_bins, uint16_t* __restrict__ sloc,
                           uint32_t* __restrict__ counts /* [chunk][bin] */) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        sloc[i] = (uint16_t)(key & 0x7fffu);
        const uint64_t c = key >> 45, w = (key >> 27) & 0x3ffffu;
        atomicAdd(&counts[c * (uint64_t)num_bins + w], 1u);
    }
}

// [chunk][bin] starts / counts -> [bin][chunk] tables; stage[bin][chunk] = offset of the run inside the bin's staged region
__global__ void k_pb_tables(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts, int num_chunks, int num_bins,
                            uint32_t* __restrict__ run_start, uint32_t* __restrict__ run_len, uint32_t* __restrict__ stage) {
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < num_bins; w += gridDim.x * blockDim.x) {
        uint32_t off = 0;
        for (int c = 0; c < num_chunks; ++c) {
            const uint32_t len = counts[(int64_t)c * num_bins + w];
            run_start[(int64_t)w * num_chunks + c] = starts[(int64_t)c * num_bins + w];
            run_len[(int64_t)w * num_chunks + c] = len;
            stage[(int64_t)w * num_chunks + c] = off;
            off += len;
        }
    }
}

// row-major key of every entry (bin << 40 | row_in_bin << 28 | chunk << 15 | source) and its position in the staged region
__global__ void k_pb_rowmajor_keys(const uint64_t* __restrict__ keys, int64_t count, int num_chunks, const uint32_t* __restrict__ run_start,
                                   const uint32_t* __restrict__ stage, uint64_t* __restrict__ keys2, uint32_t* __restrict__ pos) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        const uint64_t c = key >> 45, w = (key >> 27) & 0x3ffffu, dl = (key >> 15) & 0xfffu, sl = key & 0x7fffu;
        keys2[i] = (w << 40) | (dl << 28) | (c << 15) | sl;
        const int64_t cell = (int64_t)w * num_chunks + (int64_t)c;
        pos[i] = stage[cell] + (uint32_t)(i - (int64_t)run_start[cell]);
    }
}

// rank i of the row-major sort -> padded slot of its bin (bins start at multiples of 8 entries)
__global__ void k_pb_rowmajor_split(const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ pos, int64_t count,
                                    const int4* __restrict__ bin, const uint32_t* __restrict__ bin_rank0, uint16_t* __restrict__ perm,
                                    uint16_t* __restrict__ drow) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t w = keys2[i] >> 40;
        const int64_t slot = (int64_t)bin[w].z + (i - (int64_t)bin_rank0[w]);
        perm[slot] = (uint16_t)pos[i];
        drow[slot] = (uint16_t)((keys2[i] >> 28) & 0xfffu);
    }
}

// ------------------------------------------------------------------------------------------------- run-time kernels
struct PbView {
    const uint16_t* sloc;
    const float*    val;
    const int4*     task;
    const int*      task_range;
    float*          tmp;
    const uint32_t* run_start;
    const uint32_t* run_len;
    const int4*     bin;
    const uint16_t* perm;
    const uint16_t* drow;
    float*          out;
    int64_t         cold_prefix[9];
    int64_t         xg_base[8];
    int             num_blocks, hot, chunk, num_chunks, num_bins;
    int64_t         num_cold;          // referenced cold sources in total
};

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- phase A
template <bool HAS_VAL>
__global__ __launch_bounds__(kPbThreads) void k_pb_gather(PbView f, const float* __restrict__ xg, const LoopState* __restrict__ state) {
    __shared__ float s_x[kPbChunk];
    if (state != nullptr && state->done) return;
    // this workgroup's share of the entry stream: consecutive pieces, each inside one chunk; the LDS image of the chunk
    // is refilled only when the chunk changes
    const int piece_begin = f.task_range[blockIdx.x], piece_end = f.task_range[blockIdx.x + 1];
    int loaded = -1;
    for (int piece = piece_begin; piece < piece_end; ++piece) {
        const int4 task = f.task[piece];
        if (task.x != loaded && !(PGH_PROBE_PB & 1)) {
            __syncthreads();
            // cold ids [first_id, first_id + chunk) -> positions in the gather vector, block by block: the block loop is
            // unrolled so that the layout tables are read with constant indices (scalar loads), and a thread keeps 8
            // independent loads in flight
            const int64_t first_id = (int64_t)task.x * f.chunk;
            const int64_t last_id = min(first_id + f.chunk, f.num_cold);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b >= f.num_blocks) continue;
                const int64_t lo = max(first_id, f.cold_prefix[b]), hi = min(last_id, f.cold_prefix[b + 1]);
                if (lo >= hi) continue;                     // wavefront-uniform
                const float* __restrict__ src = xg + f.xg_base[b] + f.hot - f.cold_prefix[b];      // src[id] = value of cold id
                for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += kPbThreads * 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t id = i0 + (int64_t)u * kPbThreads;
                        v[u] = src[min(id, hi - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t id = i0 + (int64_t)u * kPbThreads;
                        if (id < hi) s_x[id - first_id] = v[u];
                    }
                }
            }
            __syncthreads();
            loaded = task.x;
        }
        // every lane takes 8 consecutive entries: one 16-byte load of source indices, two 16-byte stores of values; the
        // unaligned head / tail of the range (the arrays are 16-byte aligned at entry 0) goes entry by entry
        const int64_t begin = task.y, end = task.z;
        const int64_t body_begin = min((begin + 7) & ~(int64_t)7, end), body_end = max(end & ~(int64_t)7, body_begin);
        for (int64_t e = begin + threadIdx.x; e < body_begin; e += kPbThreads) f.tmp[e] = HAS_VAL ? s_x[f.sloc[e]] * f.val[e] : s_x[f.sloc[e]];
        for (int64_t e = body_end + threadIdx.x; e < end; e += kPbThreads) f.tmp[e] = HAS_VAL ? s_x[f.sloc[e]] * f.val[e] : s_x[f.sloc[e]];
        // four 16-byte index loads per lane in flight: one load per round trip would leave the CU latency-bound
        constexpr int P = 4;
        if (PGH_PROBE_PB & 2) continue;
        for (int64_t e0 = body_begin + (int64_t)threadIdx.x * 8; e0 < body_end; e0 += (int64_t)kPbThreads * 8 * P) {
            u16x8 s8[P];
            f32x4 w0[P], w1[P];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int64_t e = e0 + (int64_t)q * kPbThreads * 8;
                const bool ok = e < body_end;
                s8[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(f.sloc + e)) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (HAS_VAL) {
                    w0[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.val + e)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    w1[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.val + e + 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int64_t e = e0 + (int64_t)q * kPbThreads * 8;
                if (e >= body_end) continue;
                f32x4 lo, hi;
                lo.x = s_x[s8[q][0]];
                lo.y = s_x[s8[q][1]];
                lo.z = s_x[s8[q][2]];
                lo.w = s_x[s8[q][3]];
                hi.x = s_x[s8[q][4]];
                hi.y = s_x[s8[q][5]];
                hi.z = s_x[s8[q][6]];
                hi.w = s_x[s8[q][7]];
                if (HAS_VAL) {
                    lo *= w0[q];
                    hi *= w1[q];
                }
                if (PGH_PROBE_PB & 4) {
                    if (lo.x + hi.w == 123.456f) f.tmp[e] = lo.y;
                    continue;
                }
                *reinterpret_cast<f32x4*>(f.tmp + e) = lo;
                *reinterpret_cast<f32x4*>(f.tmp + e + 4) = hi;
            }
        }
    }
}

// ---- phase B
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pb_dpp_f32(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int pb_dpp_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int pb_wave_inclusive_sum(int v) {
    v += pb_dpp_i32<0x111, 0xf>(0, v);
    v += pb_dpp_i32<0x112, 0xf>(0, v);
    v += pb_dpp_i32<0x114, 0xf>(0, v);
    v += pb_dpp_i32<0x118, 0xf>(0, v);
    v += pb_dpp_i32<0x142, 0xa>(0, v);
    v += pb_dpp_i32<0x143, 0xc>(0, v);
    return v;
}
// inclusive segmented sum, head flags as keep = 0 (starts a segment) / 1 (continues the previous lane's)
__device__ __forceinline__ float pb_segmented_sum(float keep, float val) {
#define PGH_PB_STEP(CTRL, MASK)                                   \
    {                                                             \
        const float v2 = pb_dpp_f32<CTRL, MASK>(0.f, val);        \
        const float k2 = pb_dpp_f32<CTRL, MASK>(1.f, keep);       \
        val = __builtin_fmaf(v2, keep, val);                      \
        keep *= k2;                                               \
    }
    PGH_PB_STEP(0x111, 0xf)
    PGH_PB_STEP(0x112, 0xf)
    PGH_PB_STEP(0x114, 0xf)
    PGH_PB_STEP(0x118, 0xf)
    PGH_PB_STEP(0x142, 0xa)
    PGH_PB_STEP(0x143, 0xc)
#undef PGH_PB_STEP
    return val;
}

// One workgroup per bin.  Wavefront w walks the w-th contiguous part of the bin's row-major entry list in tiles of 512
// (8 consecutive entries per lane).  A row is one segment of that list.  Inside a part: the segment that contains the
// part's first entry is its HEAD piece, the one that contains its last entry its TAIL piece (a part without a row change
// is a single piece); every other segment is complete and goes straight to the bin's row array in LDS.  The pieces are
// handed over in f64 and stitched in part order by one thread after the barrier: fixed order, no atomics.
__global__ __launch_bounds__(kPbBThreads) void k_pb_accumulate(PbView f, const LoopState* __restrict__ state) {
    __shared__ float s_val[kPbBinEntries];                 // the bin's entries, staged: runs in chunk order
    __shared__ float s_row[kPbBinRows];                    // row sums of the bin
    __shared__ double s_head[kPbBWaves], s_tail[kPbBWaves];
    __shared__ int s_head_row[kPbBWaves], s_tail_row[kPbBWaves], s_pieces[kPbBWaves];   // pieces: 0 none, 1 single, 2 head + tail
    __shared__ int s_pref[kPbMaxChunks + 1];               // staged offset of every run
    __shared__ uint32_t s_start[kPbMaxChunks];             // its first entry in tmp
    __shared__ int s_group[kPbBWaves], s_group_base[kPbBWaves];
    if (state != nullptr && state->done) return;
    const int4 bin = f.bin[blockIdx.x];                    // {first row, rows, first row-major slot, entries}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t* __restrict__ starts = f.run_start + (int64_t)blockIdx.x * f.num_chunks;
    const uint32_t* __restrict__ lens = f.run_len + (int64_t)blockIdx.x * f.num_chunks;
    for (int i = tid; i < bin.y; i += kPbBThreads) s_row[i] = 0.f;
    if (PGH_PROBE_PB & 8) s_val[tid] = 1.f;
    // ---- this wavefront's part of the row-major walk: the index loads of its first tile do not depend on the staging,
    //      so they are issued now and land while the runs are being staged
    constexpr int T = 512;
    const int tiles = (bin.w + T - 1) / T;
    const int per_wave = (tiles + kPbBWaves - 1) / kPbBWaves;
    const int t_begin = min(wave * per_wave, tiles), t_end = (PGH_PROBE_PB & 16) ? 0 : min(t_begin + per_wave, tiles);
    const uint16_t* __restrict__ perm = f.perm + bin.z;
    const uint16_t* __restrict__ drow = f.drow + bin.z;
    u16x8 pk_next = {0, 0, 0, 0, 0, 0, 0, 0}, dk_next = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t_begin < t_end) {
        const int e0 = t_begin * T + lane * 8;
        if (bin.w - e0 > 0) {
            pk_next = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(perm + e0));
            dk_next = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(drow + e0));
        }
    }
    // ---- stage the runs (each contiguous in tmp) one behind the other.  The run table goes to LDS (first entry in tmp,
    //      exclusive prefix of the lengths = staged offset); then the staged region is filled FLAT: every thread owns 16
    //      consecutive staged slots, finds their run by binary search and walks forward -- sixteen independent loads per
    //      thread, i.e. one round trip for the whole bin however its entries split into runs.
    if (!(PGH_PROBE_PB & 8)) {
        for (int c0 = 0; c0 < f.num_chunks; c0 += 64 * kPbBWaves) {          // wavefront w takes descriptors c0 + 64 w ..
            const int cc = c0 + wave * 64 + lane;
            const int len = cc < f.num_chunks ? (int)lens[cc] : 0;
            const uint32_t start = cc < f.num_chunks ? starts[cc] : 0u;
            const int incl = pb_wave_inclusive_sum(len);
            if (cc < f.num_chunks) {
                s_pref[cc] = incl - len;                                    // prefix inside the wavefront's group of 64
                s_start[cc] = start;
            }
            if (lane == 63) s_group[wave] = incl;                           // total of the group
        }
        __syncthreads();
        // groups of 64 runs -> bin-wide exclusive prefix (num_chunks <= kPbMaxChunks = 64 * kPbBWaves: one pass)
        if (tid < kPbBWaves) {
            int acc = 0;
            for (int w = 0; w < tid; ++w) acc += s_group[w];
            s_group_base[tid] = acc;
        }
        __syncthreads();
        for (int c = tid; c < f.num_chunks; c += kPbBThreads) s_pref[c] += s_group_base[c >> 6];
        if (tid == 0) s_pref[f.num_chunks] = bin.w;
        __syncthreads();
        // wavefront w fills staged slots [w * S, (w + 1) * S) in steps of 64 consecutive slots (coalesced reads of tmp:
        // neighbouring slots are neighbouring entries of a run); a lane finds its run by binary search at the first
        // step and walks forward from there (a step advances by about one run)
        constexpr int S = kPbBinEntries / kPbBWaves;                         // staged slots per wavefront
        constexpr int STEPS = S / 64;
        static_assert(S % 64 == 0, "whole steps");
        const int w0 = wave * S;
        if (w0 < bin.w) {
            int lo = 0, hi = f.num_chunks;                                  // last run with pref <= w0 + lane
            const int p_first = min(w0 + lane, bin.w - 1);
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_pref[mid] <= p_first) lo = mid; else hi = mid;
            }
            int run = lo;
            float v[STEPS];
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                const int pos = w0 + k * 64 + lane;
                const int q = min(pos, bin.w - 1);
                while (s_pref[run + 1] <= q) ++run;                         // s_pref[num_chunks] = bin.w > q: terminates
                v[k] = pos < bin.w ? __builtin_nontemporal_load(f.tmp + s_start[run] + (q - s_pref[run])) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < STEPS; ++k) s_val[w0 + k * 64 + lane] = v[k];
        }
    }
    __syncthreads();
    // ---- row-major walk of this wavefront's part
    double carry = 0.0;                                     // sum so far of the segment open at the start of the tile
    int open_row = t_begin < t_end ? __shfl((int)dk_next[0], 0, 64) : -1;   // row of the part's first entry
    bool have_head = false;                                 // a row change has been seen in this part
    double head = 0.0;
    int head_row = -1;
    for (int t = t_begin; t < t_end; ++t) {
        const int e0 = t * T + lane * 8;
        const int left = bin.w - e0;                        // this lane's valid entries: min(max(left, 0), 8)
        int my_last = open_row;
        const u16x8 pk = pk_next, dk = dk_next;             // fetched one tile ahead
        if (t + 1 < t_end) {                                // the bin's range starts at a multiple of 8 slots and is padded
            const int n0 = e0 + T;
            pk_next = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            dk_next = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (bin.w - n0 > 0) {
                pk_next = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(perm + n0));
                dk_next = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(drow + n0));
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < left) my_last = (int)dk[k];
        // previous entry's row for the lane's first entry: last row of the previous lane, `open_row` for lane 0
        int prev = pb_dpp_i32<0x138, 0xf>(-1, my_last);    // wave_shr:1
        if (lane == 0) prev = open_row;
        float acc = 0.f, first_val = 0.f;
        int first_row = -1, cur = prev;
        bool any = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool valid = k < left;
            const int r = (int)dk[k];
            if (valid && r != cur) {                        // the row changes: the segment of `cur` ends here
                if (!any) {
                    first_val = acc;                        // may have started in earlier lanes / tiles: finished below
                    first_row = cur;
                    any = true;
                } else {
                    s_row[cur] = acc;                       // a row entirely inside this lane
                }
                acc = 0.f;
                cur = r;
            }
            if (valid) acc += s_val[pk[k]];
        }
        // the lane's tail (acc) continues into the following lanes until one of them sees a row change
        const float run = pb_segmented_sum(any ? 0.f : 1.f, acc);       // tails chained over lanes without a change
        const float before = pb_dpp_f32<0x138, 0xf>(0.f, run);           // what the lanes before hold of my first row
        const unsigned long long changes = __ballot(any);
        const bool first_in_tile = any && (changes & ((1ULL << lane) - 1ULL)) == 0ULL;
        if (any) {
            const double total = (double)first_val + (lane > 0 ? (double)before : 0.0) + (first_in_tile ? carry : 0.0);
            if (first_in_tile && !have_head) {
                head = total;                               // the segment that contains the part's first entry
                head_row = first_row;
            } else {
                s_row[first_row] = (float)total;
            }
        }
        const float tile_tail = __shfl(run, 63, 64);
        const int last_lane = min(63, max(0, (bin.w - t * T + 7) / 8 - 1));
        const int tile_last_row = __shfl(left > 0 ? cur : open_row, last_lane, 64);
        if (changes != 0ULL) {
            if (!have_head) {
                const int closer = __builtin_ctzll(changes);
                head = __shfl(head, closer, 64);
                head_row = __shfl(head_row, closer, 64);
                have_head = true;
            }
            carry = (double)tile_tail;
            open_row = tile_last_row;
        } else {
            carry += (double)tile_tail;
        }
    }
    if (lane == 0) {
        const int pieces = t_begin >= t_end ? 0 : (have_head ? 2 : 1);
        s_pieces[wave] = pieces;
        s_head[wave] = pieces == 2 ? head : carry;
        s_head_row[wave] = pieces == 2 ? head_row : open_row;
        s_tail[wave] = carry;
        s_tail_row[wave] = open_row;
    }
    __syncthreads();
    // ---- stitch the pieces in part order (one thread)
    if (tid == 0) {
        double open = 0.0;
        int row = -1;
        for (int w = 0; w < kPbBWaves; ++w) {
            if (s_pieces[w] == 0) continue;
            if (s_head_row[w] == row) {                     // head (or single) piece continues the open segment
                open += s_head[w];
            } else {
                if (row >= 0) s_row[row] = (float)open;
                open = s_head[w];
                row = s_head_row[w];
            }
            if (s_pieces[w] == 2) {                         // the head segment ended inside the part; the tail one is open now
                if (row >= 0) s_row[row] = (float)open;
                open = s_tail[w];
                row = s_tail_row[w];
            }
        }
        if (row >= 0) s_row[row] = (float)open;
    }
    __syncthreads();
    for (int i = tid; i < bin.y; i += kPbBThreads) f.out[bin.x + i] = s_row[i];
}

PbView pb_view(const BsfFormat& f, const PbFormat& p) {
    PbView v;
    v.sloc = p.sloc;
    v.val = p.val;
    v.task = p.task;
    v.task_range = p.task_range;
    v.tmp = p.tmp;
    v.run_start = p.run_start;
    v.run_len = p.run_len;
    v.bin = p.bin;
    v.perm = p.perm;
    v.drow = p.drow;
    v.out = p.out;
    for (int i = 0; i < 9; ++i) v.cold_prefix[i] = p.cold_prefix[i];
    for (int i = 0; i < 8; ++i) v.xg_base[i] = f.xg_base[i];
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
    // Opt-in (PGH_PB=1): measured on MI355X at scale 23 the image is correct but not yet faster than leaving the cold
    // gathers in k_bsf_partial (136 + 68 + 181 us against 372 us) -- see the analysis there.
    const char* env = getenv("PGH_PB");
    if (env == nullptr || atoi(env) == 0) return 0;
    Runtime& r = rt();
    int64_t cold_sources = 0;
    for (int b = 0; b < f.num_blocks; ++b) cold_sources += live[b] > hot ? live[b] - hot : 0;
    const int64_t chunks = (cold_sources + kPbChunk - 1) / kPbChunk;
    if (chunks < 1 || chunks > kPbMaxChunks || f.n_out >= (1 << 28)) return 0;
    PbBuf<uint32_t> d_counts;
    PGH_TRY(d_counts.alloc(f.n_out, true));
    k_pb_row_counts<<<pb_blocks_for(E), kBlock, 0, r.stream>>>(keys, is_hot, E, d_counts.p);
    std::vector<uint32_t> counts(f.n_out);
    PGH_HIP(hipMemcpyAsync(counts.data(), d_counts.p, sizeof(uint32_t) * f.n_out, hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    // greedy bins: consecutive rows, <= kPbBinEntries entries, <= kPbBinRows rows; rows heavier than a bin get none
    std::vector<int4> bins;
    std::vector<int32_t> row_bin(f.n_out);
    int64_t cold = 0, in_image = 0;
    int row0 = 0, rows = 0;
    int64_t fill = 0;
    bool heavy_rows = false;
    auto close_bin = [&]() {
        if (rows > 0) bins.push_back(make_int4(row0, rows, 0, (int)fill));
        rows = 0;
        fill = 0;
    };
    for (int i = 0; i < f.n_out; ++i) {
        const int64_t c = counts[i];
        cold += c;
        if (c > kPbBinEntries) {             // heavier than a bin: its cold entries stay in the blocked stream
            close_bin();
            row_bin[i] = -1;
            heavy_rows = true;
            continue;
        }
        if (rows > 0 && (fill + c > kPbBinEntries || rows >= kPbBinRows)) close_bin();
        if (rows == 0) row0 = i;
        row_bin[i] = (int32_t)bins.size();
        ++rows;
        fill += c;
        in_image += c;
    }
    close_bin();
    // drop bins without entries: their rows never receive a cold contribution (`out` stays 0 there)
    {
        std::vector<int4> kept;
        std::vector<int32_t> remap(bins.size(), 0);
        for (size_t w = 0; w < bins.size(); ++w)
            if (bins[w].w > 0) {
                remap[w] = (int32_t)kept.size();
                kept.push_back(bins[w]);
            }
        for (int i = 0; i < f.n_out; ++i)
            if (row_bin[i] >= 0) row_bin[i] = remap[row_bin[i]];      // rows of dropped bins have no entries: never looked up
        bins.swap(kept);
    }
    const int64_t num_bins = (int64_t)bins.size();
    if (num_bins < 1 || num_bins >= (1 << 18) || in_image >= 2147483647LL) return 0;
    const double run = (double)in_image / ((double)chunks * (double)num_bins);
    const char* force = getenv("PGH_PB_FORCE");
    const bool forced = force != nullptr && atoi(force) != 0;
    if (!forced && (in_image < (1 << 22) || in_image * 20 < E || run < 24.0)) return 0;
    PGH_HIP(hipMalloc(&plan->row_bin, sizeof(int32_t) * (size_t)f.n_out));
    PGH_HIP(hipMemcpyAsync(plan->row_bin, row_bin.data(), sizeof(int32_t) * f.n_out, hipMemcpyHostToDevice, r.stream));
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
        want = std::max(1, std::min(want, kPbMaxSlices));
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
    *use = true;
    return 0;
}

// cold_keys: the entries of the image as stream keys (block << 58 | row << 29 | col), any order; cold_vals: values or null.
int pb_build(BsfFormat& f, PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, const int* live,
             int hot) {
    Runtime& r = rt();
    PbFormat& p = slice == 0 ? f.pb : f.pb_more[slice - 1];
    p = PbFormat();
    PGH_CHECK(count == plan->slice_entries[slice], "propagation blocking: entry count does not match the plan");
    p.num_entries = count;
    p.chunk = kPbChunk;
    p.hot = hot;
    p.k1_cold = plan->heavy_rows;
    const int first_bin = plan->slice_first[slice];
    p.num_bins = plan->slice_first[slice + 1] - first_bin;
    int64_t padded_slots = 0;
    {   // this slice's bins, row-major slots re-laid from 0
        std::vector<int4> mine(plan->host_bins + first_bin, plan->host_bins + first_bin + p.num_bins);
        for (int4& b : mine) {
            b.z = (int)padded_slots;
            padded_slots += ((int64_t)b.w + 7) & ~(int64_t)7;
        }
        PGH_HIP(hipMalloc(&p.bin, sizeof(int4) * (size_t)(p.num_bins > 0 ? p.num_bins : 1)));
        PGH_HIP(hipMemcpyAsync(p.bin, mine.data(), sizeof(int4) * mine.size(), hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    p.cold_prefix[0] = 0;
    for (int b = 0; b < 8; ++b) p.cold_prefix[b + 1] = p.cold_prefix[b] + (b < f.num_blocks && live[b] > hot ? live[b] - hot : 0);
    p.num_chunks = plan->num_chunks;
    PbLayout L;
    for (int b = 0; b < 9; ++b) L.cold_prefix[b] = p.cold_prefix[b];
    L.blk = f.blk_size;
    L.hot = hot;
    L.chunk = kPbChunk;
    PbBuf<uint64_t> keys_a, keys_b, keys_c;
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
        PGH_HIP(hipMalloc(&p.val, sizeof(float) * (size_t)count));
        PGH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.p, keys_b.p, cold_vals, p.val, (int)count, 0, 58, r.stream));
    } else {
        PGH_HIP(hipcub::DeviceRadixSort::SortKeys(temp.p, temp_bytes, keys_a.p, keys_b.p, (int)count, 0, 58, r.stream));
    }
    // ---- phase A order: keys_b
    const int64_t cells = (int64_t)p.num_chunks * p.num_bins;
    PbBuf<uint32_t> counts, starts, stage, pos_a, pos_b;
    PGH_TRY(counts.alloc(cells + 1, true));
    PGH_TRY(starts.alloc(cells + 1));
    PGH_TRY(stage.alloc(cells));
    PGH_HIP(hipMalloc(&p.sloc, sizeof(uint16_t) * (size_t)(count + 8)));
    k_pb_split<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(keys_b.p, count, p.num_bins, p.sloc, counts.p);
    PGH_HIP(hipGetLastError());
    {
        size_t scan_bytes = 0;
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, counts.p, starts.p, (int)(cells + 1), r.stream));
        PbBuf<char> scan_temp;
        PGH_TRY(scan_temp.alloc(scan_bytes));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, counts.p, starts.p, (int)(cells + 1), r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    PGH_HIP(hipMalloc(&p.run_start, sizeof(uint32_t) * (size_t)cells));
    PGH_HIP(hipMalloc(&p.run_len, sizeof(uint32_t) * (size_t)cells));
    k_pb_tables<<<pb_blocks_for(p.num_bins), kBlock, 0, r.stream>>>(starts.p, counts.p, p.num_chunks, p.num_bins, p.run_start, p.run_len, stage.p);
    PGH_HIP(hipGetLastError());
    // ---- row-major order of every bin: second sort, payload = position inside the bin's staged region
    PGH_TRY(pos_a.alloc(count));
    PGH_TRY(pos_b.alloc(count));
    PGH_TRY(keys_c.alloc(count));
    k_pb_rowmajor_keys<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(keys_b.p, count, p.num_chunks, p.run_start, stage.p, keys_a.p, pos_a.p);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.p, keys_c.p, pos_a.p, pos_b.p, (int)count, 0, 58, r.stream));
    // first row-major rank of every bin = exclusive prefix of the bin sizes
    std::vector<int4> hbins(p.num_bins);
    PGH_HIP(hipMemcpyAsync(hbins.data(), p.bin, sizeof(int4) * p.num_bins, hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    PbBuf<uint32_t> bin_rank0;
    {
        std::vector<uint32_t> rank0(p.num_bins);
        uint32_t at = 0;
        for (int w = 0; w < p.num_bins; ++w) {
            rank0[w] = at;
            at += (uint32_t)hbins[w].w;
        }
        PGH_TRY(bin_rank0.alloc(p.num_bins));
        PGH_HIP(hipMemcpyAsync(bin_rank0.p, rank0.data(), sizeof(uint32_t) * p.num_bins, hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    const int64_t padded = padded_slots;
    PGH_HIP(hipMalloc(&p.perm, sizeof(uint16_t) * (size_t)(padded + 8)));
    PGH_HIP(hipMalloc(&p.drow, sizeof(uint16_t) * (size_t)(padded + 8)));
    PGH_HIP(hipMemsetAsync(p.perm, 0, sizeof(uint16_t) * (size_t)(padded + 8), r.stream));
    PGH_HIP(hipMemsetAsync(p.drow, 0xff, sizeof(uint16_t) * (size_t)(padded + 8), r.stream));
    k_pb_rowmajor_split<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(keys_c.p, pos_b.p, count, p.bin, bin_rank0.p, p.perm, p.drow);
    PGH_HIP(hipGetLastError());
    // ---- phase A shares
    std::vector<uint32_t> chunk_start(p.num_chunks + 1);
    for (int c = 0; c <= p.num_chunks; ++c)
        PGH_HIP(hipMemcpyAsync(&chunk_start[c], starts.p + (int64_t)c * p.num_bins, sizeof(uint32_t), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    // shares of the stream for the workgroups of phase A (at most one per CU), balanced by cost = entries + a fixed price
    // for every chunk image a share has to load (the tail chunks hold few entries: a share there crosses many of them)
    std::vector<int4> tasks;
    std::vector<int> ranges(1, 0);
    {
        const int64_t fill_cost = 24576;                   // a 128 KB fill ~ this many entries of streaming
        const int64_t target = (int64_t)(1.16 * (double)(count + (int64_t)p.num_chunks * fill_cost) / (double)r.num_cus) + 8;
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
    }
    const int shares = (int)ranges.size() - 1;
    p.num_tasks = shares;
    if (getenv("PGH_DEBUG") != nullptr && atoi(getenv("PGH_DEBUG")) != 0) {
        int64_t mn = count, mx = 0;
        int most = 0;
        for (const int4& t : tasks) {
            mn = std::min<int64_t>(mn, t.z - t.y);
            mx = std::max<int64_t>(mx, t.z - t.y);
        }
        for (int b = 0; b < shares; ++b) most = std::max(most, ranges[b + 1] - ranges[b]);
        fprintf(stderr, "[pgh] pb: %lld entries, %d chunks, %d bins, phase A: %zu pieces over %d shares (piece %lld..%lld entries, <= %d per share)\n",
                (long long)count, p.num_chunks, p.num_bins, tasks.size(), shares, (long long)mn, (long long)mx, most);
    }
    PGH_HIP(hipMalloc(&p.task, sizeof(int4) * (size_t)(tasks.size() + 1)));
    PGH_HIP(hipMalloc(&p.task_range, sizeof(int) * (size_t)(shares + 1)));
    if (!tasks.empty()) PGH_HIP(hipMemcpyAsync(p.task, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice, r.stream));
    PGH_HIP(hipMemcpyAsync(p.task_range, ranges.data(), sizeof(int) * (shares + 1), hipMemcpyHostToDevice, r.stream));
    PGH_HIP(hipMalloc(&p.tmp, sizeof(float) * (size_t)(count + 8)));
    if (slice == 0) {
        PGH_HIP(hipMalloc(&p.out, sizeof(float) * (size_t)(f.n_out > 0 ? f.n_out : 1)));
        PGH_HIP(hipMemsetAsync(p.out, 0, sizeof(float) * (size_t)(f.n_out > 0 ? f.n_out : 1), r.stream));
    } else {
        p.out = f.pb.out;                                  // disjoint rows of the same vector
        p.owns_out = false;
    }
    PGH_HIP(hipStreamSynchronize(r.stream));
    p.device_bytes = count * (4 + 2 + (cold_vals ? 4 : 0)) + padded * 4 + cells * 8 + (int64_t)f.n_out * 4;
    p.enabled = true;
    return 0;
}

void pb_plan_release(PbPlan* plan) {
    (void)hipFree(plan->row_bin);
    delete[] plan->host_bins;
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

int pb_launch(pgh_graph_s* g, const float* xg, const LoopState* state) {
    const BsfFormat& f = g->bsf;
    if (!f.pb.enabled) return 0;
    Runtime& r = rt();
    for (int slice = 0; slice < f.pb_slices; ++slice) {
    const PbFormat& p = slice == 0 ? f.pb : f.pb_more[slice - 1];
    const PbView v = pb_view(f, p);
    {
        ProfScope prof(PGH_K_PB_GATHER);
        if (p.num_tasks > 0) {
            if (p.val) k_pb_gather<true><<<p.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state);
            else k_pb_gather<false><<<p.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state);
        }
    }
    {
        ProfScope prof(PGH_K_PB_ACCUM);
        if (p.num_bins > 0) k_pb_accumulate<<<p.num_bins, kPbBThreads, 0, r.stream>>>(v, state);
    }
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

void pb_destroy(PbFormat& p) {
    (void)hipFree(p.sloc);
    (void)hipFree(p.val);
    (void)hipFree(p.task);
    (void)hipFree(p.task_range);
    (void)hipFree(p.tmp);
    (void)hipFree(p.run_start);
    (void)hipFree(p.run_len);
    (void)hipFree(p.bin);
    (void)hipFree(p.perm);
    (void)hipFree(p.drow);
    if (p.owns_out) (void)hipFree(p.out);
    p = PbFormat();
}

}  // namespace pgh
