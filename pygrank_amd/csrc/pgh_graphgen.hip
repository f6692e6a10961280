// Synthetic power-law workload on the device: Graph500-style RMAT edges -> normalised CSR(M^T) in HBM.
//
// The reference has no generator ("the build writes its own"); this is the build-side input
// path for BASELINE.json configs[1..4] (RMAT scale 23) and configs[4] (scale 27, row-partitioned).  It also
// performs, on the GPU, the normalisation the reference's preprocessor does on the host:
//   "col"        M = D^-1 A                      pygrank/core/utils/preprocessing.py:109-113
//   "symmetric"  M = Dl^-1/2 A Dr^-1/2           pygrank/core/utils/preprocessing.py:131-138
// with the same fp64 evaluation order ((1/deg) * w, then * right) before the single rounding to f32, zero-degree
// rows left zero (preprocessing.py:111), duplicate edges summed into weights and self-loops kept
// (coo -> csr semantics of pygrank/fastgraph/fastgraph.py:77-78).
//
// Edges are a pure function of (seed, edge index): splitmix64 hashes and 32-bit integer thresholds, bit-for-bit
// the same as oracle/rmat_np.py, so host tests see exactly the graph the GPU builds.  Sorting / run-length
// encoding use rocPRIM through hipCUB: one-time format construction, not the per-iteration hot path.
#include "pgh_kernels.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

using namespace pgh;


namespace {

constexpr int kBlock = 256;

struct RmatParams {
    int      scale;
    int64_t  num_edges;       // generated (directed) edges
    uint32_t ta, tb, tc;      // quadrant thresholds on a 32-bit uniform
    uint64_t seed;
    int      symmetrize;      // also emit (dst, src)
    int64_t  row_begin, row_end;
    const int32_t* iperm;   // row-partitioned mode: old id -> globally relabelled id (null otherwise)
};

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ void rmat_edge(const RmatParams& P, uint64_t e, uint32_t& src, uint32_t& dst) {
    const uint64_t emix = e * 0xD6E8FEB86659FD93ULL;
    uint32_t s = 0, d = 0;
    const int pairs = (P.scale + 1) / 2;
    for (int pair = 0; pair < pairs; ++pair) {
        const uint64_t key = splitmix64(P.seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(pair + 1));
        const uint64_t h = splitmix64(key ^ emix);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int level = 2 * pair + half;
            if (level < P.scale) {
                const uint32_t u = half == 0 ? (uint32_t)(h >> 32) : (uint32_t)(h & 0xffffffffu);
                const uint32_t rbit = u >= P.tb;                                    // quadrants c, d
                const uint32_t cbit = ((u >= P.ta) && (u < P.tb)) || (u >= P.tc);   // quadrants b, d
                const int shift = P.scale - 1 - level;
                s |= rbit << shift;
                d |= cbit << shift;
            }
        }
    }
    src = s;
    dst = d;
}

// pass 1: global out-degree weights (edge multiplicities per source) and number of edges kept by this row range
// (has_in != null: marks the ids that receive an edge -- the relabelling of a partitioned graph breaks ties with it)
__global__ void k_rmat_count(RmatParams P, unsigned int* __restrict__ outdeg, unsigned long long* __restrict__ kept,
                             unsigned int* __restrict__ has_in = nullptr) {
    unsigned long long local = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < P.num_edges; e += stride) {
        uint32_t s, d;
        rmat_edge(P, (uint64_t)e, s, d);
        if (P.iperm != nullptr) {
            s = (uint32_t)P.iperm[s];
            d = (uint32_t)P.iperm[d];
        }
        atomicAdd(&outdeg[s], 1u);
        if (has_in != nullptr) {
            has_in[d] = 1u;
            if (P.symmetrize) has_in[s] = 1u;
        }
        if ((int64_t)d >= P.row_begin && (int64_t)d < P.row_end) ++local;
        if (P.symmetrize) {
            atomicAdd(&outdeg[d], 1u);
            if ((int64_t)s >= P.row_begin && (int64_t)s < P.row_end) ++local;
        }
    }
    // wavefront-aggregate before the global atomic
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(kept, local);
}

// relabelling keys of a partitioned graph, in place: count -> (count << 1) | receives an edge; live = ids with any edge
__global__ void k_part_keys(unsigned int* __restrict__ cnt, const unsigned int* __restrict__ has_in, int64_t n, unsigned int* __restrict__ live) {
    unsigned int mine = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int c = cnt[i] < 0x7fffffffu ? cnt[i] : 0x7fffffffu, h = has_in[i] != 0u ? 1u : 0u;
        cnt[i] = (c << 1) | h;
        mine += (c | h) != 0u ? 1u : 0u;
    }
    if (mine) atomicAdd(live, mine);
}

// pass 2: emit sort keys (local_row << 32 | src) for the kept edges
__global__ void k_rmat_fill(RmatParams P, uint64_t* __restrict__ keys, unsigned long long* __restrict__ cursor, int dense) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < P.num_edges; e += stride) {
        uint32_t s, d;
        rmat_edge(P, (uint64_t)e, s, d);
        if (P.iperm != nullptr) {
            s = (uint32_t)P.iperm[s];
            d = (uint32_t)P.iperm[d];
        }
        if (dense) {                       // whole row range, no symmetrisation: edge e owns slot e (deterministic)
            keys[e] = ((uint64_t)d << 32) | s;
            continue;
        }
        if ((int64_t)d >= P.row_begin && (int64_t)d < P.row_end) {
            const unsigned long long pos = atomicAdd(cursor, 1ULL);
            keys[pos] = ((uint64_t)(d - (uint32_t)P.row_begin) << 32) | s;
        }
        if (P.symmetrize && (int64_t)s >= P.row_begin && (int64_t)s < P.row_end) {
            const unsigned long long pos = atomicAdd(cursor, 1ULL);
            keys[pos] = ((uint64_t)(s - (uint32_t)P.row_begin) << 32) | d;
        }
    }
}

// ---- degrees without global atomics (round 6; whole-graph generation): the edge list sorted by one end, run lengths = degrees.
// k_rmat_count / k_indeg add one global atomic per edge into counters whose hubs take a large share of them: 17 + 19 ms of a 0.24-s
// build at scale 23 against ~4 + ~1 ms for a 32-bit sort + run-length encode / a run-length encode of the already sorted rows.
struct HiWord {
    __host__ __device__ __forceinline__ uint32_t operator()(const uint64_t& k) const { return (uint32_t)(k >> 32); }
};
__global__ void k_low_words(const uint64_t* __restrict__ keys, int64_t count, uint32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)(keys[i] & 0xffffffffu);
}
__global__ void k_scatter_runs(const uint32_t* __restrict__ ids, const int* __restrict__ lengths, const int* __restrict__ num_runs,
                               unsigned int* __restrict__ out) {
    const int runs = *num_runs;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < runs; i += gridDim.x * blockDim.x) out[ids[i]] = (unsigned int)lengths[i];
}

// in-degree weights of the local rows from the run-length encoded keys
__global__ void k_indeg(const uint64_t* __restrict__ ukeys, const int* __restrict__ counts, int64_t nnz,
                        unsigned int* __restrict__ indeg) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride)
        atomicAdd(&indeg[(uint32_t)(ukeys[k] >> 32)], (unsigned int)counts[k]);
}

// normalised values + column indices + row pointer + row sums of M (degrees)
__global__ void k_rmat_values(const uint64_t* __restrict__ ukeys, const int* __restrict__ counts, int64_t nnz, int64_t n_local,
                              int normalization, const unsigned int* __restrict__ outdeg,
                              const unsigned int* __restrict__ indeg, int32_t* __restrict__ col, float* __restrict__ val,
                              int32_t* __restrict__ rowptr, double* __restrict__ deg_acc) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const uint64_t key = ukeys[k];
        const int64_t row = (int64_t)(key >> 32);            // local destination row
        const uint32_t src = (uint32_t)(key & 0xffffffffu);
        const double w = (double)counts[k];
        double v;
        if (normalization == 0) {                            // "col": (1 / rowsum(A)[src]) * w, preprocessing.py:110-113
            const double s = (double)outdeg[src];
            v = (s != 0.0 ? 1.0 / s : 0.0) * w;
        } else if (normalization == 1) {                     // "symmetric": (l * w) * r, preprocessing.py:132-138
            const double l = sqrt((double)outdeg[src]);
            const double r = sqrt((double)indeg[row]);
            v = ((l != 0.0 ? 1.0 / l : 0.0) * w) * (r != 0.0 ? 1.0 / r : 0.0);
        } else {
            v = w;
        }
        col[k] = (int32_t)src;
        val[k] = (float)v;
        atomicAdd(&deg_acc[src], v);                         // row sums of M (numpy.py:76-77); f64, rounded once below
        const int64_t prev = (k == 0) ? -1 : (int64_t)(ukeys[k - 1] >> 32);
        for (int64_t r = prev + 1; r <= row; ++r) rowptr[r] = (int32_t)k;
        if (k == nnz - 1)
            for (int64_t r = row + 1; r <= n_local; ++r) rowptr[r] = (int32_t)nnz;
    }
}

// per-source / per-row scales of the value-free blocked format: M^T = diag(dst) * multiplicities * diag(src)
__global__ void k_rmat_scales(const unsigned int* __restrict__ outdeg, const unsigned int* __restrict__ indeg, int64_t n,
                              int64_t n_local, int normalization, float* __restrict__ src, float* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const double d = (double)outdeg[i];
        double s = 1.0;
        if (normalization == 0) s = d != 0.0 ? 1.0 / d : 0.0;
        if (normalization == 1) s = d != 0.0 ? 1.0 / sqrt(d) : 0.0;
        src[i] = (float)s;
        if (dst != nullptr && i < n_local) {
            const double e = (double)indeg[i];
            dst[i] = (float)(e != 0.0 ? 1.0 / sqrt(e) : 0.0);
        }
    }
}

__global__ void k_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

inline int blocks_for(int64_t n) {
    int64_t b = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rt().num_cus * 16;
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

}  // namespace

namespace {
__global__ void k_gather_u32(const unsigned int* __restrict__ src, const int32_t* __restrict__ perm, int64_t n, unsigned int* __restrict__ dst) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[perm[i]];
}
}  // namespace

// part_count == 0: ids as generated, rows [row_begin, row_end).  part_count > 0: ids globally relabelled by
// descending source count into B = max(part_count, auto) hot-first blocks (identical on every rank), this rank keeps
// the contiguous slice of new ids [part_rank * n / part_count, (part_rank + 1) * n / part_count).
static int rmat_build(int32_t scale, int32_t edge_factor, double a, double b, double c, uint64_t seed, int32_t normalization,
                      int32_t symmetrize, int64_t row_begin, int64_t row_end, int32_t part_rank, int32_t part_count,
                      pgh_graph_t* out) {
    PGH_TRY(ensure_init());
    PGH_CHECK(scale >= 1 && scale <= 30 && edge_factor >= 1, "pgh_graph_rmat: scale must be in [1, 30]");
    PGH_CHECK(normalization >= 0 && normalization <= 2, "pgh_graph_rmat: normalization must be 0 (col), 1 (symmetric) or 2 (none)");
    const int64_t n = 1LL << scale;
    if (row_end <= 0) row_end = n;
    int part_blocks = 0;
    if (part_count > 0) {
        PGH_CHECK((part_count & (part_count - 1)) == 0 && part_count <= n && part_rank >= 0 && part_rank < part_count,
                  "pgh_graph_rmat_part: the number of partitions must be a power of two and the rank inside it");
        part_blocks = bsf_auto_blocks(n);
        if (part_blocks < part_count) part_blocks = part_count;
        PGH_CHECK(part_blocks <= 8, "pgh_graph_rmat_part: at most 8 partitions per node");
        row_begin = (int64_t)part_rank * (n / part_count);
        row_end = row_begin + n / part_count;
    }
    PGH_CHECK(row_begin >= 0 && row_begin <= row_end && row_end <= n, "pgh_graph_rmat: bad row range");
    Runtime& r = rt();
    RmatParams P;
    P.scale = scale;
    P.num_edges = n * (int64_t)edge_factor;
    P.ta = (uint32_t)floor(a * 4294967296.0);
    P.tb = (uint32_t)floor((a + b) * 4294967296.0);
    P.tc = (uint32_t)floor((a + b + c) * 4294967296.0);
    P.seed = seed;
    P.symmetrize = symmetrize ? 1 : 0;
    P.row_begin = row_begin;
    P.row_end = row_end;
    P.iperm = nullptr;
    const int64_t n_local = row_end - row_begin;
    const bool dense = (row_begin == 0 && row_end == n && !symmetrize && part_count == 0);

    pgh_graph_s* g = new pgh_graph_s();
    g->n_rows = n;          // rows of M = sources = length of the gathered vector
    g->n_cols = n_local;    // rows of the stored M^T slice = length of the output
    g->row_begin = row_begin;
    build_clock_reset();
    int rc = [&]() -> int {
        DevBuf<unsigned int> outdeg, indeg;
        DevBuf<unsigned long long> counters;
        DevBuf<uint64_t> keys_a, keys_b, ukeys;
        DevBuf<int> counts, num_runs;
        DevBuf<double> deg_acc;
        PGH_TRY(outdeg.alloc(n, true));
        PGH_TRY(indeg.alloc(n_local, true));
        PGH_TRY(counters.alloc(2, true));
        DevBuf<int32_t> iperm;
        if (part_count > 0) {
            // pass 0: global source counts in the generated ids -> the relabelling every rank derives identically
            // (ties broken in favour of ids that receive an edge: the isolated ids -- no edge at all -- then form the tail of
            // every block, i.e. of every rank's slice; BsfFormat::iso_begin)
            DevBuf<unsigned int> outdeg_old, has_in, live_count;
            PGH_TRY(outdeg_old.alloc(n, true));
            PGH_TRY(has_in.alloc(n, true));
            PGH_TRY(live_count.alloc(1, true));
            RmatParams P0 = P;
            P0.row_begin = P0.row_end = 0;
            k_rmat_count<<<blocks_for(P.num_edges), kBlock, 0, r.stream>>>(P0, outdeg_old.p, counters.p, has_in.p);
            k_part_keys<<<blocks_for(n), kBlock, 0, r.stream>>>(outdeg_old.p, has_in.p, n, live_count.p);     // in place -> sort keys
            PGH_HIP(hipGetLastError());
            PGH_TRY(iperm.alloc(n));
            PGH_HIP(pooled_malloc(&g->part_perm, sizeof(int32_t) * (size_t)n));
            PGH_TRY(build_count_perm(outdeg_old.p, n, part_blocks, (int)(n / part_blocks), g->part_perm, iperm.p));
            unsigned int live_nodes = 0;
            PGH_HIP(hipMemcpyAsync(&live_nodes, live_count.p, sizeof(unsigned int), hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
            g->part_live_nodes = (int64_t)live_nodes;
            PGH_HIP(hipMemsetAsync(counters.p, 0, sizeof(unsigned long long) * 2, r.stream));
            P.iperm = iperm.p;
        }
        // whole graph, ids as generated: every edge is kept and the degrees come from sorted runs below (no atomic per edge)
        const bool by_runs = dense && P.num_edges < 2147483647LL && !(getenv("PGH_RMAT_ATOMICS") != nullptr && atoi(getenv("PGH_RMAT_ATOMICS")) != 0);
        unsigned long long kept = (unsigned long long)P.num_edges;
        if (!by_runs) {
            k_rmat_count<<<blocks_for(P.num_edges), kBlock, 0, r.stream>>>(P, outdeg.p, counters.p);
            PGH_HIP(hipGetLastError());
            PGH_HIP(hipMemcpyAsync(&kept, counters.p, sizeof(kept), hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
        }
        PGH_CHECK(kept < 2147483647ULL, "pgh_graph_rmat: more than 2^31 edges in one partition; use more partitions");
        build_mark("generator: out-degrees");
        const int64_t K = (int64_t)kept;
        int64_t nnz = 0;
        PGH_HIP(pooled_malloc(&g->rowptr, sizeof(int32_t) * (size_t)(n_local + 1)));
        PGH_HIP(pooled_malloc(&g->degrees, sizeof(float) * (size_t)n));
        PGH_TRY(deg_acc.alloc(n, true));
        if (K > 0) {
            PGH_TRY(keys_a.alloc(K));
            PGH_TRY(keys_b.alloc(K));
            build_mark("generator: allocations");
            k_rmat_fill<<<blocks_for(P.num_edges), kBlock, 0, r.stream>>>(P, keys_a.p, counters.p + 1, dense ? 1 : 0);
            PGH_HIP(hipGetLastError());
            build_mark("generator: edge keys");
            DevBuf<uint32_t> run_ids;
            DevBuf<int> run_lengths, run_count;
            if (by_runs) {
                // out-degrees: the sources alone (low words), sorted on `scale` bits, run lengths scattered to their ids
                PGH_TRY(run_ids.alloc(n));
                PGH_TRY(run_lengths.alloc(n));
                PGH_TRY(run_count.alloc(1));
                uint32_t* words = reinterpret_cast<uint32_t*>(keys_b.p);           // keys_b is free until the edge sort: 2 K words of 4 bytes
                uint32_t* words_sorted = words + K;
                k_low_words<<<blocks_for(K), kBlock, 0, r.stream>>>(keys_a.p, K, words);
                size_t bytes_sort = 0, bytes_rle = 0;
                PGH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes_sort, words, words_sorted, (int)K, 0, scale, r.stream));
                PGH_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, bytes_rle, words_sorted, run_ids.p, run_lengths.p, run_count.p, (int)K, r.stream));
                DevBuf<char> temp;
                PGH_TRY(temp.alloc(bytes_sort > bytes_rle ? bytes_sort : bytes_rle));
                PGH_HIP(hipcub::DeviceRadixSort::SortKeys(temp.p, bytes_sort, words, words_sorted, (int)K, 0, scale, r.stream));
                PGH_HIP(hipcub::DeviceRunLengthEncode::Encode(temp.p, bytes_rle, words_sorted, run_ids.p, run_lengths.p, run_count.p, (int)K, r.stream));
                k_scatter_runs<<<blocks_for(n), kBlock, 0, r.stream>>>(run_ids.p, run_lengths.p, run_count.p, outdeg.p);
                PGH_HIP(hipGetLastError());
                PGH_HIP(hipStreamSynchronize(r.stream));
                build_mark("generator: out-degrees");
            }
            int bits_row = 1;
            while ((1LL << bits_row) < n_local) ++bits_row;
            size_t temp_bytes = 0;
            PGH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys_a.p, keys_b.p, (int)K, 0, 32 + bits_row, r.stream));
            {
                DevBuf<char> temp;
                PGH_TRY(temp.alloc(temp_bytes));
                PGH_HIP(hipcub::DeviceRadixSort::SortKeys(temp.p, temp_bytes, keys_a.p, keys_b.p, (int)K, 0, 32 + bits_row, r.stream));
                PGH_HIP(hipStreamSynchronize(r.stream));
            }
            build_mark("generator: edge sort");
            if (by_runs) {
                // in-degrees: the rows (high words) of the sorted edge list, run-length encoded
                hipcub::TransformInputIterator<uint32_t, HiWord, const uint64_t*> rows(keys_b.p, HiWord());
                size_t bytes_rle = 0;
                PGH_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, bytes_rle, rows, run_ids.p, run_lengths.p, run_count.p, (int)K, r.stream));
                DevBuf<char> temp;
                PGH_TRY(temp.alloc(bytes_rle));
                PGH_HIP(hipcub::DeviceRunLengthEncode::Encode(temp.p, bytes_rle, rows, run_ids.p, run_lengths.p, run_count.p, (int)K, r.stream));
                k_scatter_runs<<<blocks_for(n), kBlock, 0, r.stream>>>(run_ids.p, run_lengths.p, run_count.p, indeg.p);
                PGH_HIP(hipGetLastError());
                PGH_HIP(hipStreamSynchronize(r.stream));
                build_mark("generator: in-degrees");
            }
            // run-length encode duplicates into weights; keys_a is reused for the unique keys
            PGH_TRY(counts.alloc(K));
            PGH_TRY(num_runs.alloc(1));
            temp_bytes = 0;
            PGH_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, temp_bytes, keys_b.p, keys_a.p, counts.p, num_runs.p, (int)K, r.stream));
            {
                DevBuf<char> temp;
                PGH_TRY(temp.alloc(temp_bytes));
                PGH_HIP(hipcub::DeviceRunLengthEncode::Encode(temp.p, temp_bytes, keys_b.p, keys_a.p, counts.p, num_runs.p, (int)K, r.stream));
                int runs = 0;
                PGH_HIP(hipMemcpyAsync(&runs, num_runs.p, sizeof(int), hipMemcpyDeviceToHost, r.stream));
                PGH_HIP(hipStreamSynchronize(r.stream));
                nnz = runs;
            }
            build_mark("generator: run lengths");
            PGH_HIP(pooled_malloc(&g->col, sizeof(int32_t) * (size_t)nnz));
            PGH_HIP(pooled_malloc(&g->val, sizeof(float) * (size_t)nnz));
            if (!by_runs) k_indeg<<<blocks_for(nnz), kBlock, 0, r.stream>>>(keys_a.p, counts.p, nnz, indeg.p);
            k_rmat_values<<<blocks_for(nnz), kBlock, 0, r.stream>>>(keys_a.p, counts.p, nnz, n_local, normalization, outdeg.p,
                                                                    indeg.p, g->col, g->val, g->rowptr, deg_acc.p);
            PGH_HIP(hipGetLastError());
        } else {
            PGH_HIP(pooled_malloc(&g->col, sizeof(int32_t)));
            PGH_HIP(pooled_malloc(&g->val, sizeof(float)));
            k_fill_i32<<<blocks_for(n_local + 1), kBlock, 0, r.stream>>>(g->rowptr, n_local + 1, 0);
        }
        g->nnz = nnz;
        k_f64_to_f32<<<blocks_for(n), kBlock, 0, r.stream>>>(deg_acc.p, g->degrees, n);
        PGH_HIP(hipGetLastError());
        PGH_HIP(hipStreamSynchronize(r.stream));
        build_mark("generator: values of CSR(M^T), row sums of M");
        PGH_TRY(finish_graph(g));
        build_mark("row-major tile table");
        // blocked value-free layout: multiplicities as repeated entries, the normalisation as vector scales
        const char* fmt = getenv("PGH_FORMAT");
        if (nnz > 0 && (fmt == nullptr || std::string(fmt) != "csr")) {
            const char* rl = getenv("PGH_RELABEL");
            const bool valfree = getenv("PGH_VALUES") == nullptr || atoi(getenv("PGH_VALUES")) == 0;
            if (valfree) {
                DevBuf<float> src, dst;
                PGH_TRY(src.alloc(n));
                if (normalization == 1) PGH_TRY(dst.alloc(n_local));
                k_rmat_scales<<<blocks_for(n), kBlock, 0, r.stream>>>(outdeg.p, indeg.p, n, n_local, normalization, src.p,
                                                                      normalization == 1 ? dst.p : nullptr);
                PGH_HIP(hipGetLastError());
                if (part_count == 0 && row_begin == 0 && row_end == n && !symmetrize) {
                    // every generated edge references its source once: the out-degrees ARE the relabelling's reference counts
                    g->src_counts = outdeg.release();
                    g->src_counts_weighted = true;
                }
                PGH_TRY(bsf_build(g, nullptr, counts.p, normalization == 2 ? nullptr : src.p,
                                  normalization == 1 ? dst.p : nullptr, part_count == 0 && (rl == nullptr || atoi(rl) != 0),
                                  part_blocks));
                // keep the factorisation for the multi-seed layout, which is built on first use (pgh_spmm.hip)
                g->keep_mult = (int32_t*)counts.release();
                if (normalization != 2) g->keep_src = src.release();
                if (normalization == 1) g->keep_dst = dst.release();
            } else {
                PGH_TRY(bsf_build(g, g->val, nullptr, nullptr, nullptr, part_count == 0 && (rl == nullptr || atoi(rl) != 0), part_blocks));
            }
        }
        return 0;
    }();
    if (rc != 0) {
        std::string keep = pgh_last_error();
        pgh_graph_destroy(g);
        return fail(keep);
    }
    *out = g;
    return 0;
}

extern "C" int pgh_graph_rmat(int32_t scale, int32_t edge_factor, double a, double b, double c, uint64_t seed,
                              int32_t normalization, int32_t symmetrize, int64_t row_begin, int64_t row_end,
                              pgh_graph_t* out) {
    return rmat_build(scale, edge_factor, a, b, c, seed, normalization, symmetrize, row_begin, row_end, 0, 0, out);
}

extern "C" int pgh_graph_rmat_part(int32_t scale, int32_t edge_factor, double a, double b, double c, uint64_t seed,
                                   int32_t normalization, int32_t symmetrize, int32_t part_rank, int32_t part_count,
                                   pgh_graph_t* out) {
    PGH_CHECK(part_count >= 1, "pgh_graph_rmat_part: part_count must be >= 1");
    return rmat_build(scale, edge_factor, a, b, c, seed, normalization, symmetrize, 0, 0, part_rank, part_count, out);
}

PGH_WARM_KERNEL(k_low_words)
