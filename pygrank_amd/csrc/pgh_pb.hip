// Propagation blocking for the cold tail of the blocked stream.
//
// k_bsf_partial serves the gathers whose source sits in the block's LDS hot cache; every other ("cold") gather costs a
// 128-byte L1 line fill for 4 useful bytes, and on the bench graph those fills -- not HBM -- bound the step (DESIGN.md
// section 4).  This image removes them: the cold entries are taken out of the stream and processed by two streaming
// passes whose random accesses all land in LDS.
//
//   order of the cold entries: (source chunk c, output bin w, output row, source)
//   phase A  k_pb_gather     one workgroup per (chunk, entry range): the chunk's slice of the gather vector goes to LDS
//                            (128 KB, coalesced), then tmp[e] = x_chunk[sloc[e]] (* val[e])  -- 2 B read + 4 B written
//                            per entry, sequential.
//   phase B  k_pb_accumulate one wavefront per bin of 1024 output rows, f64 sums in LDS: for every chunk the bin's run
//                            of entries (contiguous in tmp, sorted by row) is read sequentially, equal rows are folded
//                            with a DPP segmented scan and added to the bin's sums; the bin is written once.
//                            4 B + 2 B read per entry.  Fixed order: deterministic, no atomics.
//
// 12 sequential bytes per cold entry instead of one line fill.  Runs must stay long enough to feed a wavefront, which
// limits the image to graphs where cold_entries / (chunks * bins) >= ~24 (scale <= 24 on RMAT); beyond that the cold
// entries stay in k_bsf_partial.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "pgh_kernels.h"

namespace pgh {
namespace {

constexpr int kBlock = 256;
constexpr int kPbChunk = 32768;          // sources per chunk: 128 KB of LDS
constexpr int kPbRows = 1024;            // rows per wavefront bin: 8 KB of f64 sums, 16 bins per workgroup
constexpr int kPbThreads = 1024;
constexpr int kPbTask = 196608;          // entries per phase A workgroup (the 128 KB chunk fill amortises over them)
constexpr int kPbUnit = 8192;            // entries per phase B wavefront unit
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

struct PbLayout {
    int64_t cold_prefix[9];
    int     blk, hot, chunk, rows, num_bins;
};

// stream key (block << 58 | row << 29 | col) -> propagation-blocking key (chunk << 43 | bin << 25 | row_in_bin << 15 | source_in_chunk)
__global__ void k_pb_keys(const uint64_t* __restrict__ keys, int64_t count, PbLayout L, uint64_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        const int b = (int)(key >> 58);
        const int64_t row = (int64_t)((key >> 29) & kLow29);
        const int64_t col = (int64_t)(key & kLow29);
        const int64_t loc = col - (int64_t)b * L.blk;
        const int64_t cold_id = L.cold_prefix[b] + (loc - L.hot);
        const uint64_t c = (uint64_t)(cold_id / L.chunk), sl = (uint64_t)(cold_id % L.chunk);
        const uint64_t w = (uint64_t)(row / L.rows), dl = (uint64_t)(row % L.rows);
        out[i] = (c << 43) | (w << 25) | (dl << 15) | sl;
    }
}

__global__ void k_pb_split(const uint64_t* __restrict__ keys, int64_t count, int num_bins, uint16_t* __restrict__ sloc,
                           uint16_t* __restrict__ dloc, uint32_t* __restrict__ counts /* [chunk][bin] */) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        sloc[i] = (uint16_t)(key & 0x7fffu);
        dloc[i] = (uint16_t)((key >> 15) & 0x3ffu);
        const uint64_t c = key >> 43, w = (key >> 25) & 0x3ffffu;
        atomicAdd(&counts[c * (uint64_t)num_bins + w], 1u);
    }
}

// [chunk][bin] starts / counts -> [bin][chunk] tables
__global__ void k_pb_transpose(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts, int num_chunks, int num_bins,
                               uint32_t* __restrict__ run_start, uint32_t* __restrict__ run_len) {
    const int64_t total = (int64_t)num_chunks * num_bins;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = i / num_chunks, c = i % num_chunks;
        run_start[i] = starts[c * num_bins + w];
        run_len[i] = counts[c * num_bins + w];
    }
}

struct PbView {
    const uint16_t* sloc;
    const uint16_t* dloc;
    const float*    val;
    const uint32_t* run_start;
    const uint32_t* run_len;
    const int4*     task;
    const int4*     unit;
    const int4*     merge;
    double*         extra;
    int             num_units, num_merges;
    float*          tmp;
    float*          out;
    int64_t         cold_prefix[9];
    int64_t         xg_base[8];
    int             num_blocks, hot, chunk, num_chunks, num_bins, n_out;
    int64_t         num_cold;          // referenced cold sources in total
};

// ---- phase A
template <bool HAS_VAL>
__global__ __launch_bounds__(kPbThreads) void k_pb_gather(PbView f, const float* __restrict__ xg, const LoopState* __restrict__ state) {
    __shared__ float s_x[kPbChunk];
    if (state != nullptr && state->done) return;
    const int4 task = f.task[blockIdx.x];
    const int64_t first_id = (int64_t)task.x * f.chunk;
    for (int i = threadIdx.x; i < f.chunk; i += kPbThreads) {
        const int64_t id = first_id + i;
        float v = 0.f;
        if (id < f.num_cold) {
            int b = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) b += (k < f.num_blocks && id >= f.cold_prefix[k]) ? 1 : 0;
            v = xg[f.xg_base[b] + f.hot + (id - f.cold_prefix[b])];
        }
        s_x[i] = v;
    }
    __syncthreads();
    // every lane takes 8 consecutive entries: one 16-byte load of source indices, two 16-byte stores of values; the
    // unaligned head / tail of the range (the arrays are 16-byte aligned at entry 0) goes entry by entry
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int64_t begin = task.y, end = task.z;
    const int64_t body_begin = min((begin + 7) & ~(int64_t)7, end), body_end = max(end & ~(int64_t)7, body_begin);
    for (int64_t e = begin + threadIdx.x; e < body_begin; e += kPbThreads) f.tmp[e] = HAS_VAL ? s_x[f.sloc[e]] * f.val[e] : s_x[f.sloc[e]];
    for (int64_t e = body_end + threadIdx.x; e < end; e += kPbThreads) f.tmp[e] = HAS_VAL ? s_x[f.sloc[e]] * f.val[e] : s_x[f.sloc[e]];
    for (int64_t e0 = body_begin + (int64_t)threadIdx.x * 8; e0 < body_end; e0 += (int64_t)kPbThreads * 8) {
        const u16x8 s8 = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(f.sloc + e0));
        f32x4 lo, hi;
        lo.x = s_x[s8[0]];
        lo.y = s_x[s8[1]];
        lo.z = s_x[s8[2]];
        lo.w = s_x[s8[3]];
        hi.x = s_x[s8[4]];
        hi.y = s_x[s8[5]];
        hi.z = s_x[s8[6]];
        hi.w = s_x[s8[7]];
        if (HAS_VAL) {
            const f32x4 w0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.val + e0));
            const f32x4 w1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.val + e0 + 4));
            lo *= w0;
            hi *= w1;
        }
        *reinterpret_cast<f32x4*>(f.tmp + e0) = lo;
        *reinterpret_cast<f32x4*>(f.tmp + e0 + 4) = hi;
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

__device__ __forceinline__ int pb_wave_inclusive_sum(int v) {
    v += pb_dpp_i32<0x111, 0xf>(0, v);
    v += pb_dpp_i32<0x112, 0xf>(0, v);
    v += pb_dpp_i32<0x114, 0xf>(0, v);
    v += pb_dpp_i32<0x118, 0xf>(0, v);
    v += pb_dpp_i32<0x142, 0xa>(0, v);
    v += pb_dpp_i32<0x143, 0xc>(0, v);
    return v;
}

// one wavefront per unit = a slice [first, last) of the entries of one bin, its runs concatenated in chunk order
__global__ __launch_bounds__(kPbThreads) void k_pb_accumulate(PbView f, const LoopState* __restrict__ state) {
    __shared__ double s_acc[kPbThreads / 64][kPbRows];
    if (state != nullptr && state->done) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int u = blockIdx.x * (kPbThreads / 64) + wave;
    if (u >= f.num_units) return;
    const int4 unit = f.unit[u];
    const int bin = unit.x;
    double* __restrict__ acc = s_acc[wave];
    for (int i = lane; i < kPbRows; i += 64) acc[i] = 0.0;
    const uint32_t* __restrict__ starts = f.run_start + (int64_t)bin * f.num_chunks;
    const uint32_t* __restrict__ lens = f.run_len + (int64_t)bin * f.num_chunks;
    // one step: up to 64 entries of one run (sorted by row): fold equal rows, add the folded sums to the bin
    auto fold = [&](float v, int d, bool valid) {
        const int dp = pb_dpp_i32<0x138, 0xf>(-1, valid ? d : -2);          // wave_shr:1: row of the previous lane
        const int dn = pb_dpp_i32<0x130, 0xf>(-3, valid ? d : -2);          // wave_shl:1: row of the next lane
        const float keep = (valid && dp == d) ? 1.f : 0.f;
        const float sum = pb_segmented_sum(keep, valid ? v : 0.f);
        if (valid && dn != d) acc[d] += (double)sum;
    };
    constexpr int G = 8;                       // runs in flight per wavefront
    int vbase = 0;                             // entries of the bin in the chunks before c0
    for (int c0 = 0; c0 < f.num_chunks && vbase < unit.z; c0 += 64) {
        // the next 64 runs: lane-parallel load of the descriptors, clipped to this unit's slice of the bin
        const int cc = c0 + lane;
        const int len = cc < f.num_chunks ? (int)lens[cc] : 0;
        const int vend = vbase + pb_wave_inclusive_sum(len), vstart = vend - len;
        const int lo = max(unit.y, vstart), hi = min(unit.z, vend);
        const uint32_t my_start = (cc < f.num_chunks ? starts[cc] : 0u) + (uint32_t)max(lo - vstart, 0);
        const uint32_t my_len = hi > lo ? (uint32_t)(hi - lo) : 0u;
        vbase = __shfl(vend, 63, 64);
        const unsigned long long live = __ballot(my_len != 0u);
        if (live == 0ULL) continue;
        const int g_first = (__builtin_ctzll(live) / G) * G, g_last = 63 - __builtin_clzll(live);
        for (int g0 = g_first; g0 <= g_last; g0 += G) {
            float v[G];
            int d[G];
            uint32_t st[G], ln[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                st[k] = __shfl(my_start, min(g0 + k, 63), 64);
                ln[k] = (g0 + k < 64) ? __shfl(my_len, min(g0 + k, 63), 64) : 0u;
                const bool valid = (uint32_t)lane < ln[k];
                v[k] = valid ? __builtin_nontemporal_load(f.tmp + st[k] + lane) : 0.f;
                d[k] = valid ? (int)__builtin_nontemporal_load(f.dloc + st[k] + lane) : 0;
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (ln[k] == 0u) continue;                                  // wavefront-uniform
                fold(v[k], d[k], (uint32_t)lane < ln[k]);
                // long runs (hub rows; at most kPbUnit entries inside a unit): next step's loads issued before this one folds
                if (ln[k] > 64u) {
                    bool nvalid = 64u + lane < ln[k];
                    float nv = nvalid ? f.tmp[st[k] + 64 + lane] : 0.f;
                    int nd = nvalid ? (int)f.dloc[st[k] + 64 + lane] : 0;
                    for (uint32_t i = 64; i < ln[k]; i += 64) {
                        const bool cvalid = nvalid;
                        const float cv = nv;
                        const int cd = nd;
                        nvalid = i + 64 + lane < ln[k];
                        nv = nvalid ? f.tmp[st[k] + i + 64 + lane] : 0.f;
                        nd = nvalid ? (int)f.dloc[st[k] + i + 64 + lane] : 0;
                        fold(cv, cd, cvalid);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (unit.w < 0) {                          // the only unit of its bin
        const int64_t row0 = (int64_t)bin * kPbRows;
        for (int i = lane; i < kPbRows; i += 64)
            if (row0 + i < f.n_out) f.out[row0 + i] = (float)acc[i];
    } else {
        double* __restrict__ dst = f.extra + (int64_t)unit.w * kPbRows;
        for (int i = lane; i < kPbRows; i += 64) dst[i] = acc[i];
    }
}

// bins cut into several units: fixed-order sum of the unit partials
__global__ void k_pb_merge(PbView f, const LoopState* __restrict__ state) {
    if (state != nullptr && state->done) return;
    const int4 m = f.merge[blockIdx.x];
    const int64_t row0 = (int64_t)m.x * kPbRows;
    for (int i = threadIdx.x; i < kPbRows; i += blockDim.x) {
        double total = 0.0;
        for (int s = 0; s < m.z; ++s) total += f.extra[(int64_t)(m.y + s) * kPbRows + i];
        if (row0 + i < f.n_out) f.out[row0 + i] = (float)total;
    }
}

__global__ void k_pb_bin_totals(const uint32_t* __restrict__ run_len, int num_bins, int num_chunks, uint32_t* __restrict__ totals) {
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < num_bins; w += gridDim.x * blockDim.x) {
        uint32_t t = 0;
        for (int c = 0; c < num_chunks; ++c) t += run_len[(int64_t)w * num_chunks + c];
        totals[w] = t;
    }
}

PbView pb_view(const BsfFormat& f) {
    const PbFormat& p = f.pb;
    PbView v;
    v.sloc = p.sloc;
    v.dloc = p.dloc;
    v.val = p.val;
    v.run_start = p.run_start;
    v.run_len = p.run_len;
    v.task = p.task;
    v.unit = p.unit;
    v.merge = p.merge;
    v.extra = p.extra;
    v.num_units = p.num_units;
    v.num_merges = p.num_merges;
    v.tmp = p.tmp;
    v.out = p.out;
    for (int i = 0; i < 9; ++i) v.cold_prefix[i] = p.cold_prefix[i];
    for (int i = 0; i < 8; ++i) v.xg_base[i] = f.xg_base[i];
    v.num_blocks = f.num_blocks;
    v.hot = p.hot;
    v.chunk = p.chunk;
    v.num_chunks = p.num_chunks;
    v.num_bins = p.num_bins;
    v.n_out = f.n_out;
    v.num_cold = p.cold_prefix[f.num_blocks];
    return v;
}

}  // namespace

// Is the propagation-blocking image worth building?  cold: number of cold entries, live[b]: referenced prefix of block b.
bool pb_wanted(const BsfFormat& f, int64_t cold_entries, int64_t all_entries, const int* live, int hot) {
    const char* e = getenv("PGH_PB");
    if (e != nullptr && atoi(e) == 0) return false;
    if (cold_entries < (1 << 22) || cold_entries * 20 < all_entries) return false;      // small graph, or hardly any cold gathers
    int64_t cold_sources = 0;
    for (int b = 0; b < f.num_blocks; ++b) cold_sources += live[b] > hot ? live[b] - hot : 0;
    const int64_t chunks = (cold_sources + kPbChunk - 1) / kPbChunk, bins = (f.n_out + kPbRows - 1) / kPbRows;
    if (chunks < 1 || chunks >= (1 << 13) || bins >= (1 << 18)) return false;
    const double run = (double)cold_entries / ((double)chunks * (double)bins);
    const char* force = getenv("PGH_PB_FORCE");
    return run >= 24.0 || (force != nullptr && atoi(force) != 0);
}

// cold_keys: the cold entries of the stream as (block << 58 | row << 29 | col) keys, any order; cold_vals: their values or null.
int pb_build(BsfFormat& f, const uint64_t* cold_keys, const float* cold_vals, int64_t count, const int* live, int hot) {
    Runtime& r = rt();
    PbFormat& p = f.pb;
    p = PbFormat();
    p.num_entries = count;
    p.chunk = kPbChunk;
    p.rows_per_bin = kPbRows;
    p.hot = hot;
    p.num_bins = (f.n_out + kPbRows - 1) / kPbRows;
    p.cold_prefix[0] = 0;
    for (int b = 0; b < 8; ++b) p.cold_prefix[b + 1] = p.cold_prefix[b] + (b < f.num_blocks && live[b] > hot ? live[b] - hot : 0);
    p.num_chunks = (int)((p.cold_prefix[f.num_blocks] + kPbChunk - 1) / kPbChunk);
    PGH_CHECK(count < 4294967295LL && p.num_chunks >= 1, "propagation blocking: bad size");
    PbLayout L;
    for (int b = 0; b < 9; ++b) L.cold_prefix[b] = p.cold_prefix[b];
    L.blk = f.blk_size;
    L.hot = hot;
    L.chunk = kPbChunk;
    L.rows = kPbRows;
    L.num_bins = p.num_bins;
    PbBuf<uint64_t> keys_a, keys_b;
    PbBuf<float> vals_b;
    PGH_TRY(keys_a.alloc(count));
    PGH_TRY(keys_b.alloc(count));
    k_pb_keys<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(cold_keys, count, L, keys_a.p);
    PGH_HIP(hipGetLastError());
    {
        size_t temp_bytes = 0;
        if (cold_vals) {
            PGH_HIP(hipMalloc(&p.val, sizeof(float) * (size_t)count));
            PGH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_a.p, keys_b.p, cold_vals, p.val, (int)count, 0, 56, r.stream));
        } else {
            PGH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys_a.p, keys_b.p, (int)count, 0, 56, r.stream));
        }
        PbBuf<char> temp;
        PGH_TRY(temp.alloc(temp_bytes));
        if (cold_vals) PGH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.p, keys_b.p, cold_vals, p.val, (int)count, 0, 56, r.stream));
        else PGH_HIP(hipcub::DeviceRadixSort::SortKeys(temp.p, temp_bytes, keys_a.p, keys_b.p, (int)count, 0, 56, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    const int64_t cells = (int64_t)p.num_chunks * p.num_bins;
    PbBuf<uint32_t> counts, starts;
    PGH_TRY(counts.alloc(cells + 1, true));
    PGH_TRY(starts.alloc(cells + 1));
    PGH_HIP(hipMalloc(&p.sloc, sizeof(uint16_t) * (size_t)count));
    PGH_HIP(hipMalloc(&p.dloc, sizeof(uint16_t) * (size_t)count));
    k_pb_split<<<pb_blocks_for(count), kBlock, 0, r.stream>>>(keys_b.p, count, p.num_bins, p.sloc, p.dloc, counts.p);
    PGH_HIP(hipGetLastError());
    {
        size_t temp_bytes = 0;
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, counts.p, starts.p, (int)(cells + 1), r.stream));
        PbBuf<char> temp;
        PGH_TRY(temp.alloc(temp_bytes));
        PGH_HIP(hipcub::DeviceScan::ExclusiveSum(temp.p, temp_bytes, counts.p, starts.p, (int)(cells + 1), r.stream));
    }
    PGH_HIP(hipMalloc(&p.run_start, sizeof(uint32_t) * (size_t)cells));
    PGH_HIP(hipMalloc(&p.run_len, sizeof(uint32_t) * (size_t)cells));
    k_pb_transpose<<<pb_blocks_for(cells), kBlock, 0, r.stream>>>(starts.p, counts.p, p.num_chunks, p.num_bins, p.run_start, p.run_len);
    PGH_HIP(hipGetLastError());
    // phase A work list: every chunk's entry range cut into pieces of kPbTask entries
    std::vector<uint32_t> chunk_start(p.num_chunks + 1);
    for (int c = 0; c <= p.num_chunks; ++c)
        PGH_HIP(hipMemcpyAsync(&chunk_start[c], starts.p + (int64_t)c * p.num_bins, sizeof(uint32_t), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    std::vector<int4> tasks;
    for (int c = 0; c < p.num_chunks; ++c)
        for (int64_t b0 = chunk_start[c]; b0 < chunk_start[c + 1]; b0 += kPbTask)
            tasks.push_back(make_int4(c, (int)b0, (int)std::min<int64_t>(b0 + kPbTask, chunk_start[c + 1]), 0));
    // phase B work list
    {
        PbBuf<uint32_t> d_totals;
        PGH_TRY(d_totals.alloc(p.num_bins));
        k_pb_bin_totals<<<pb_blocks_for(p.num_bins), kBlock, 0, r.stream>>>(p.run_len, p.num_bins, p.num_chunks, d_totals.p);
        std::vector<uint32_t> totals(p.num_bins);
        PGH_HIP(hipMemcpyAsync(totals.data(), d_totals.p, sizeof(uint32_t) * p.num_bins, hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        std::vector<int4> units, merges;
        int slots = 0;
        for (int w = 0; w < p.num_bins; ++w) {            // heavy bins first would balance better; rounds are short anyway
            const int64_t t = totals[w];
            if (t == 0) continue;                          // structurally empty bin: `out` keeps its zeros
            if (t <= kPbUnit) {
                units.push_back(make_int4(w, 0, (int)t, -1));
            } else {
                const int k = (int)((t + kPbUnit - 1) / kPbUnit);
                merges.push_back(make_int4(w, slots, k, 0));
                for (int j = 0; j < k; ++j)
                    units.push_back(make_int4(w, j * kPbUnit, (int)std::min<int64_t>((int64_t)(j + 1) * kPbUnit, t), slots + j));
                slots += k;
            }
        }
        p.num_units = (int)units.size();
        p.num_merges = (int)merges.size();
        PGH_HIP(hipMalloc(&p.unit, sizeof(int4) * (size_t)(units.size() + 1)));
        PGH_HIP(hipMalloc(&p.merge, sizeof(int4) * (size_t)(merges.size() + 1)));
        PGH_HIP(hipMalloc(&p.extra, sizeof(double) * (size_t)(slots + 1) * kPbRows));
        if (!units.empty()) PGH_HIP(hipMemcpyAsync(p.unit, units.data(), sizeof(int4) * units.size(), hipMemcpyHostToDevice, r.stream));
        if (!merges.empty()) PGH_HIP(hipMemcpyAsync(p.merge, merges.data(), sizeof(int4) * merges.size(), hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
    }
    p.num_tasks = (int)tasks.size();
    PGH_HIP(hipMalloc(&p.task, sizeof(int4) * (size_t)(p.num_tasks > 0 ? p.num_tasks : 1)));
    if (p.num_tasks > 0) PGH_HIP(hipMemcpyAsync(p.task, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice, r.stream));
    PGH_HIP(hipMalloc(&p.tmp, sizeof(float) * (size_t)count));
    PGH_HIP(hipMalloc(&p.out, sizeof(float) * (size_t)(f.n_out > 0 ? f.n_out : 1)));
    PGH_HIP(hipMemsetAsync(p.out, 0, sizeof(float) * (size_t)(f.n_out > 0 ? f.n_out : 1), r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    p.device_bytes = count * (4 + 2 + 2 + (cold_vals ? 4 : 0)) + cells * 8 + (int64_t)f.n_out * 4;
    p.enabled = true;
    return 0;
}

int pb_launch(pgh_graph_s* g, const float* xg, const LoopState* state) {
    const BsfFormat& f = g->bsf;
    if (!f.pb.enabled) return 0;
    Runtime& r = rt();
    const PbView v = pb_view(f);
    {
        ProfScope prof(PGH_K_PB_GATHER);
        if (f.pb.num_tasks > 0) {
            if (f.pb.val) k_pb_gather<true><<<f.pb.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state);
            else k_pb_gather<false><<<f.pb.num_tasks, kPbThreads, 0, r.stream>>>(v, xg, state);
        }
    }
    {
        ProfScope prof(PGH_K_PB_ACCUM);
        const int grid = (f.pb.num_units + kPbThreads / 64 - 1) / (kPbThreads / 64);
        if (grid > 0) k_pb_accumulate<<<grid, kPbThreads, 0, r.stream>>>(v, state);
        if (f.pb.num_merges > 0) k_pb_merge<<<f.pb.num_merges, 256, 0, r.stream>>>(v, state);
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

void pb_destroy(PbFormat& p) {
    (void)hipFree(p.sloc);
    (void)hipFree(p.dloc);
    (void)hipFree(p.val);
    (void)hipFree(p.run_start);
    (void)hipFree(p.run_len);
    (void)hipFree(p.task);
    (void)hipFree(p.unit);
    (void)hipFree(p.merge);
    (void)hipFree(p.extra);
    (void)hipFree(p.tmp);
    (void)hipFree(p.out);
    p = PbFormat();
}

}  // namespace pgh
