// Graph upload: host CSR(M) -> device CSR(M^T) + degrees(M) + merge-path tile table.
//
// Reference counterpart: scipy_sparse_to_backend(M) (pygrank/core/backend/specification.py:70-71), called once
// per (graph, preprocessor) at pygrank/core/utils/preprocessing.py:144; the fp32 device engines of the
// reference also store the transpose (pytorch.py:63-65, torch_sparse.py:72-76).  degrees(M) = row sums of the
// un-transposed matrix (numpy.py:76-77).
//
// The transposition is a device radix sort of (column << 32 | row) keys (rocPRIM via hipCUB -- a one-time
// format conversion, not part of the per-iteration hot path), so the stored rows of M^T come out with their
// column indices ascending and the result is deterministic.
#include <cstring>
#include "pgh_kernels.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

using namespace pgh;

namespace {

constexpr int kBlock = 256;
constexpr int kItemsPerTile = 256 * PGH_IPT;

inline int blocks_for(int64_t n, int cap_mult = 16) {
    int64_t b = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rt().num_cus * cap_mult;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// row sums of M in f64 (one wavefront per row keeps hub rows parallel) -> f32 degrees
__global__ void k_row_sums(const int64_t* __restrict__ indptr, const double* __restrict__ data, int64_t n_rows,
                           float* __restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        double acc = 0.0;
        for (int64_t k = b + lane; k < e; k += 64) acc += data[k];
        acc = wave_reduce_sum(acc);
        if (lane == 0) deg[r] = (float)acc;
    }
}

// reference count of every source from the caller's CSR(M): the entries of its row, multiplicities summed on value-free graphs
// (what k_source_counts would count with global atomics over CSR(M^T))
__global__ void k_row_counts(const int64_t* __restrict__ indptr, const int32_t* __restrict__ mult, int64_t n_rows, unsigned int* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    if (mult == nullptr) {
        for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
            cnt[r] = (unsigned int)(indptr[r + 1] - indptr[r]);
        return;
    }
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        double acc = 0.0;
        for (int64_t k = b + lane; k < e; k += 64) acc += (double)mult[k];
        acc = wave_reduce_sum(acc);
        if (lane == 0) cnt[r] = (unsigned int)acc;
    }
}

// expand CSR(M) into sort keys (col << 32 | row) and f32 values
__global__ void k_make_keys(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                            const double* __restrict__ data, int64_t n_rows, uint64_t* __restrict__ keys,
                            float* __restrict__ vals) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        for (int64_t k = b + lane; k < e; k += 64) {
            keys[k] = ((uint64_t)(uint32_t)indices[k] << 32) | (uint64_t)(uint32_t)r;
            vals[k] = (float)data[k];
        }
    }
}

// sorted keys -> col array of M^T (= row of M) and rowptr of M^T (boundaries of the high word)
__global__ void k_split_keys(const uint64_t* __restrict__ keys, int64_t nnz, int64_t n_cols, int32_t* __restrict__ colT,
                             int32_t* __restrict__ rowptrT) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const uint64_t key = keys[k];
        const int64_t c = (int64_t)(key >> 32);
        colT[k] = (int32_t)(key & 0xffffffffu);
        const int64_t prev = (k == 0) ? -1 : (int64_t)(keys[k - 1] >> 32);
        for (int64_t r = prev + 1; r <= c; ++r) rowptrT[r] = (int32_t)k;     // rows (prev, c] start at k
        if (k == nnz - 1)
            for (int64_t r = c + 1; r <= n_cols; ++r) rowptrT[r] = (int32_t)nnz;
    }
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// merge-path start coordinate of every tile: diagonal d = tile * items; the A list are the row-end
// offsets rowptr[1..n], the B list the nnz indices (Merrill & Garland merge-based SpMV)
__global__ void k_tile_coords(const int32_t* __restrict__ rowptr, int n, int nnz, int items, int num_tiles,
                              int2* __restrict__ coord) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= num_tiles; t += gridDim.x * blockDim.x) {
        const int64_t total = (int64_t)n + nnz;
        int64_t d = (int64_t)t * items;
        if (d > total) d = total;
        int64_t lo = d - nnz > 0 ? d - nnz : 0;
        int64_t hi = d < n ? d : n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)rowptr[mid + 1] <= d - mid - 1) lo = mid + 1; else hi = mid;
        }
        coord[t] = make_int2((int)lo, (int)(d - lo));
    }
}

// chain_first[t]: tile t's first row began in an earlier tile and ends inside t -> index of the first tile
// whose tail carry belongs to that row; -1 otherwise
__global__ void k_chain_first(const int32_t* __restrict__ rowptr, const int2* __restrict__ coord, int n, int num_tiles,
                              int32_t* __restrict__ chain_first) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < num_tiles; t += gridDim.x * blockDim.x) {
        const int row0 = coord[t].x, z0 = coord[t].y;
        int first = -1;
        if (row0 < n && rowptr[row0] < z0 && coord[t + 1].x > row0) {
            int s = t - 1;
            while (s > 0 && coord[s].x == row0) --s;     // tile s-1 also ends inside row0 when tile s starts in it
            // now coord[s].x != row0 (tile s started before row0 and ends in it) or s == 0
            first = s;
        }
        chain_first[t] = first;
    }
}

__global__ void k_flags_from_exclude(const float* __restrict__ ex, int64_t n, char* __restrict__ flags) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flags[i] = ex[i] == 0.f ? 1 : 0;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() {
        if (p) (void)pooled_free(p);
    }
    int alloc(size_t count) {
        PGH_HIP(pooled_malloc(&p, sizeof(T) * (count > 0 ? count : 1)));
        return 0;
    }
    T* release() {
        T* q = p;
        p = nullptr;
        return q;
    }
};

}  // namespace

namespace pgh {

// Builds tile table + carry buffers for a graph whose CSR(M^T) arrays are already on the device.
int finish_graph(pgh_graph_s* g) {
    Runtime& r = rt();
    const int64_t nT = g->n_cols;
    const int64_t total = nT + g->nnz;
    PGH_CHECK(total / kItemsPerTile < (1 << 30), "graph too large for the int32 tile table");
    g->items_per_tile = kItemsPerTile;
    g->num_tiles = (int)((total + kItemsPerTile - 1) / kItemsPerTile);
    const int nt = g->num_tiles;
    PGH_HIP(pooled_malloc(&g->tile_coord, sizeof(int2) * (size_t)(nt + 1)));
    PGH_HIP(pooled_malloc(&g->chain_first, sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1)));
    PGH_HIP(pooled_malloc(&g->tail_carry, sizeof(double) * (size_t)(nt > 0 ? nt : 1)));
    PGH_HIP(pooled_malloc(&g->head_partial, sizeof(double) * (size_t)(nt > 0 ? nt : 1)));
    PGH_HIP(hipMemsetAsync(g->tail_carry, 0, sizeof(double) * (size_t)(nt > 0 ? nt : 1), r.stream));
    PGH_HIP(hipMemsetAsync(g->head_partial, 0, sizeof(double) * (size_t)(nt > 0 ? nt : 1), r.stream));
    k_tile_coords<<<blocks_for(nt + 1), kBlock, 0, r.stream>>>(g->rowptr, (int)nT, (int)g->nnz, kItemsPerTile, nt, g->tile_coord);
    if (nt > 0)
        k_chain_first<<<blocks_for(nt), kBlock, 0, r.stream>>>(g->rowptr, g->tile_coord, (int)nT, nt, g->chain_first);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));
    g->device_bytes = (int64_t)sizeof(int32_t) * (nT + 1) + (int64_t)(sizeof(int32_t) + sizeof(float)) * g->nnz +
                      (int64_t)sizeof(float) * g->n_rows + (int64_t)(sizeof(int2) + 4 + 16) * (nt + 1);
    return 0;
}

}  // namespace pgh

namespace {

// factored upload: data[k] <- (left[row] * w[k]) * right[col] in fp64 (the reference's evaluation order,
// preprocessing.py:113,138); all_int records whether every weight is a small positive integer
__global__ void k_apply_factors(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, double* __restrict__ data,
                                const double* __restrict__ left, const double* __restrict__ right, int64_t n_rows,
                                int32_t* __restrict__ mult, unsigned long long* __restrict__ stats /* [0]=non-integer count, [1]=sum w */) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long bad = 0, sum = 0;
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        const double l = left ? left[r] : 1.0;
        for (int64_t k = indptr[r] + lane; k < indptr[r + 1]; k += 64) {
            const double w = data[k];
            const double rw = right ? right[indices[k]] : 1.0;
            const double m = rint(w);
            if (!(w == m && m >= 1.0 && m <= 32768.0)) ++bad; else sum += (unsigned long long)m;
            mult[k] = (int32_t)(m >= 1.0 && m <= 32768.0 ? m : 1.0);
            data[k] = (l * w) * rw;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_down(bad, off, 64);
        sum += __shfl_down(sum, off, 64);
    }
    if (lane == 0) {
        if (bad) atomicAdd(&stats[0], bad);
        if (sum) atomicAdd(&stats[1], sum);
    }
}

// W + self_loops * I and the identity of the laplacian (pgh_graph_from_adjacency_ex): row r of the augmented CSR keeps its entries and
// gains `extra` diagonal entries at its end (weights w0, w1); indptr2[r] = indptr[r] + extra * r
__global__ void k_augment_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const double* __restrict__ data,
                               int64_t n_rows, int extra, double w0, double w1, int64_t* __restrict__ indptr2, int32_t* __restrict__ indices2,
                               double* __restrict__ data2) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        const int64_t b = indptr[r], e = indptr[r + 1], shift = (int64_t)extra * r;
        for (int64_t k = b + lane; k < e; k += 64) {
            indices2[k + shift] = indices[k];
            data2[k + shift] = data[k];
        }
        if (lane < extra) {
            indices2[e + shift + lane] = (int32_t)r;
            data2[e + shift + lane] = lane == 0 ? w0 : w1;
        }
        if (lane == 0) {
            indptr2[r] = b + shift;
            if (r == n_rows - 1) indptr2[n_rows] = e + shift + extra;
        }
    }
}
// M = I - N (preprocessing.py:122): every value negated, the last entry of every row (the placeholder k_augment_rows appended) is the 1
__global__ void k_laplacian_finish(const int64_t* __restrict__ indptr, double* __restrict__ data, int64_t n_rows, int64_t nnz) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride) data[k] = -data[k];
}
__global__ void k_laplacian_ones(const int64_t* __restrict__ indptr, double* __restrict__ data, int64_t n_rows) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) data[indptr[r + 1] - 1] = 1.0;
}

// ---- entry-parallel forms of the per-row passes of an upload (round 6).  A wavefront per row (k_make_keys_idx, k_row_sums_f64,
// k_apply_factors, k_row_sums, k_row_counts) spends a whole wavefront on every empty row and walks a hub row serially: 72 + 25 ms of a
// 0.16-s upload at scale 23.  Here the row of every entry is materialised once (row heads scattered, a running maximum spreads them),
// elementwise passes stream the entries, and row sums are hipcub's load-balanced reduce-by-key over the (ascending) row ids.
__global__ void k_row_heads64(const int64_t* __restrict__ indptr, int64_t n_rows, int32_t* __restrict__ row_of) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
        if (indptr[r + 1] > indptr[r]) row_of[indptr[r]] = (int32_t)r;
}
__global__ void k_make_keys_idx_flat(const int32_t* __restrict__ row_of, const int32_t* __restrict__ indices, int64_t nnz,
                                     uint64_t* __restrict__ keys, int32_t* __restrict__ idx) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        keys[k] = ((uint64_t)(uint32_t)indices[k] << 32) | (uint64_t)(uint32_t)row_of[k];
        idx[k] = (int32_t)k;
    }
}
__global__ void k_apply_factors_flat(const int32_t* __restrict__ row_of, const int32_t* __restrict__ indices, double* __restrict__ data,
                                     const double* __restrict__ left, const double* __restrict__ right, int64_t nnz,
                                     int32_t* __restrict__ mult, unsigned long long* __restrict__ stats) {
    unsigned long long bad = 0, sum = 0;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const double l = left ? left[row_of[k]] : 1.0;
        const double w = data[k];
        const double rw = right ? right[indices[k]] : 1.0;
        const double m = rint(w);
        if (!(w == m && m >= 1.0 && m <= 32768.0)) ++bad; else sum += (unsigned long long)m;
        mult[k] = (int32_t)(m >= 1.0 && m <= 32768.0 ? m : 1.0);
        data[k] = (l * w) * rw;
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_down(bad, off, 64);
        sum += __shfl_down(sum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(&stats[0], bad);
        if (sum) atomicAdd(&stats[1], sum);
    }
}
__global__ void k_gather_f64(const int32_t* __restrict__ idx, const double* __restrict__ data, int64_t nnz, double* __restrict__ out) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) out[k] = data[idx[k]];
}
template <typename T, typename O>
__global__ void k_scatter_sums(const int32_t* __restrict__ ids, const T* __restrict__ sums, const int* __restrict__ num, O* __restrict__ out) {
    const int runs = *num;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < runs; i += gridDim.x * blockDim.x) out[ids[i]] = (O)sums[i];
}
struct HiWord32 {
    __host__ __device__ __forceinline__ int32_t operator()(const uint64_t& k) const { return (int32_t)(k >> 32); }
};

// expand CSR(M) into sort keys (col << 32 | row) with the entry index as payload
__global__ void k_make_keys_idx(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t n_rows,
                                uint64_t* __restrict__ keys, int32_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += nwaves)
        for (int64_t k = indptr[r] + lane; k < indptr[r + 1]; k += 64) {
            keys[k] = ((uint64_t)(uint32_t)indices[k] << 32) | (uint64_t)(uint32_t)r;
            idx[k] = (int32_t)k;
        }
}

__global__ void k_gather_vals(const int32_t* __restrict__ idx, const double* __restrict__ data, const int32_t* __restrict__ mult,
                              int64_t nnz, float* __restrict__ val_t, int32_t* __restrict__ mult_t) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const int32_t src = idx[k];
        val_t[k] = (float)data[src];
        if (mult_t) mult_t[k] = mult[src];
    }
}

__global__ void k_f64_to_f32_g(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

// f64 row sums of a CSR matrix, one wavefront per row, fixed summation order.  idx != null: values are data[idx[k]]
// (the transposed structure after the key sort: sums over the COLUMNS of the uploaded matrix)
template <typename PtrT>
__global__ void k_row_sums_f64(const PtrT* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ data,
                               int64_t n_rows, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        double acc = 0.0;
        for (int64_t k = (int64_t)ptr[r] + lane; k < (int64_t)ptr[r + 1]; k += 64) acc += idx ? data[idx[k]] : data[k];
        acc = wave_reduce_sum(acc);
        if (lane == 0) out[r] = acc;
    }
}

// preprocessing.py:111 / numpy semantics of the reference's scaling vectors: v -> sqrt(v) (symmetric), then 1 / v where v != 0
__global__ void k_inv_nonzero(double* __restrict__ v, int64_t n, int take_sqrt) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x = v[i];
        if (take_sqrt) x = sqrt(x);
        v[i] = x != 0.0 ? 1.0 / x : x;
    }
}

__global__ void k_fill_f64(double* p, int64_t n, double v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// scipy raises on a CSR whose row pointers decrease or whose column indices leave [0, n_cols); an unchecked one would
// make k_split_keys write past rowptr and the radix sort drop key bits.  bit 0: indptr not monotone within [0, nnz],
// bit 1: a column index out of range.
__global__ void k_validate_csr(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t n_rows, int64_t n_cols,
                               int64_t nnz, unsigned int* __restrict__ bad) {
    unsigned int mine = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += stride) {
        const int64_t a = indptr[i], b = indptr[i + 1];
        if (a > b || a < 0 || b > nnz) mine |= 1u;
    }
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const int32_t c = indices[k];
        if (c < 0 || (int64_t)c >= n_cols) mine |= 2u;
    }
    if (mine) atomicOr(bad, mine);
}

// Shared implementation of the uploads.
//   factored == false: data holds the values of M.
//   factored == true : data holds the weights W (null = all ones) and M = diag(left) W diag(right); the scale vectors
//                      come from the host (left / right) or are evaluated here from W (device_norm = PGH_NORM_*).
// part (nullable): the matrix is a slice of a globally relabelled graph -- columns [row_begin, row_begin + n_cols) of M in
// the new id space, i.e. this rank's rows of M^T; the blocked layout then keeps the caller's ids (no relabelling) and
// uses the column-block count every rank agreed on.
struct PartSpec {
    int64_t        row_begin;
    int32_t        num_blocks;
    const int32_t* perm;        // host, [n_rows] new id -> original id (-1 = padding)
};
// out[r] = sum of vals over the entries whose (ascending) key is r; rows without entries get 0.  T: the type the sums are formed in.
template <typename T, typename O, typename KeyIt, typename ValIt>
int sums_by_row(KeyIt keys, ValIt vals, int64_t count, int64_t n_rows, O* out) {
    Runtime& r = rt();
    PGH_HIP(hipMemsetAsync(out, 0, sizeof(O) * (size_t)(n_rows > 0 ? n_rows : 1), r.stream));
    if (count <= 0 || n_rows <= 0) return 0;
    DevBuf<int32_t> ids;
    DevBuf<T> sums;
    DevBuf<int> num;
    DevBuf<char> temp;
    PGH_TRY(ids.alloc(n_rows));
    PGH_TRY(sums.alloc(n_rows));
    PGH_TRY(num.alloc(1));
    size_t bytes = 0;
    PGH_HIP(hipcub::DeviceReduce::ReduceByKey(nullptr, bytes, keys, ids.p, vals, sums.p, num.p, hipcub::Sum(), (int)count, r.stream));
    PGH_TRY(temp.alloc(bytes));
    PGH_HIP(hipcub::DeviceReduce::ReduceByKey(temp.p, bytes, keys, ids.p, vals, sums.p, num.p, hipcub::Sum(), (int)count, r.stream));
    k_scatter_sums<T, O><<<blocks_for(n_rows), kBlock, 0, r.stream>>>(ids.p, sums.p, num.p, out);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(r.stream));               // (the temporaries go out of scope)
    return 0;
}

int graph_from_csr_impl(int64_t n_rows, int64_t n_cols, int64_t nnz_in, const int64_t* indptr, const int32_t* indices,
                        const double* data, bool factored, const double* left, const double* right, int device_norm,
                        pgh_graph_t* out, const PartSpec* part = nullptr, double self_loops = 0.0) {
    int64_t nnz = nnz_in;             // (grows by the diagonal entries of pgh_graph_from_adjacency_ex)
    const bool laplacian = device_norm == PGH_NORM_LAPLACIAN;
    const int extra_diag = (self_loops != 0.0 ? 1 : 0) + (laplacian ? 1 : 0);
    PGH_CHECK(extra_diag == 0 || (n_rows == n_cols && device_norm >= 0), "pgh_graph_from_adjacency_ex: self-loops / the laplacian need a square adjacency");
    PGH_TRY(ensure_init());
    PGH_CHECK(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "pgh_graph_from_csr: negative size");
    PGH_CHECK(n_rows < 2147483647LL && n_cols < 2147483647LL && nnz + extra_diag * n_rows < 2147483647LL,
              "pgh_graph_from_csr: int32 index space exceeded; row-partition the graph (SURVEY.md 8e)");
    PGH_CHECK(indptr != nullptr && (nnz == 0 || (indices && (data || device_norm >= 0))), "pgh_graph_from_csr: null array");
    PGH_CHECK(indptr[0] == 0 && indptr[n_rows] == nnz, "pgh_graph_from_csr: indptr does not match nnz");
    Runtime& r = rt();
    pgh_graph_s* g = new pgh_graph_s();
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    build_clock_reset();
    int rc = [&]() -> int {
        DevBuf<int64_t> d_indptr;
        DevBuf<int32_t> d_indices, d_mult, idx_a, idx_b, mult_t;
        DevBuf<double> d_data, d_left, d_right;
        DevBuf<uint64_t> keys_a, keys_b;
        DevBuf<unsigned long long> stats;
        DevBuf<int32_t> row_tmp;
        int32_t* row_of = nullptr;                             // [nnz] the row of every entry of the caller's CSR(M) (entry-parallel passes)
        const bool by_row = getenv("PGH_UPLOAD_BY_ROW") != nullptr && atoi(getenv("PGH_UPLOAD_BY_ROW")) != 0;     // rounds 1-5 (A/B)
        PGH_TRY(d_indptr.alloc(n_rows + 1));
        PGH_TRY(d_indices.alloc(nnz));
        PGH_TRY(d_data.alloc(nnz));
        PGH_HIP(hipMemcpyAsync(d_indptr.p, indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice, r.stream));
        if (nnz > 0) {
            PGH_HIP(hipMemcpyAsync(d_indices.p, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, r.stream));
            if (data != nullptr) PGH_HIP(hipMemcpyAsync(d_data.p, data, sizeof(double) * nnz, hipMemcpyHostToDevice, r.stream));
            else k_fill_f64<<<blocks_for(nnz), kBlock, 0, r.stream>>>(d_data.p, nnz, 1.0);
        }
        {
            DevBuf<unsigned int> bad;
            PGH_TRY(bad.alloc(1));
            PGH_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned int), r.stream));
            k_validate_csr<<<blocks_for(nnz > n_rows ? nnz : n_rows), kBlock, 0, r.stream>>>(d_indptr.p, d_indices.p, n_rows, n_cols, nnz, bad.p);
            unsigned int h_bad = 0;
            PGH_HIP(hipMemcpyAsync(&h_bad, bad.p, sizeof(unsigned int), hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
            PGH_CHECK((h_bad & 1u) == 0, "pgh_graph_from_csr: indptr is not non-decreasing within [0, nnz]");
            PGH_CHECK((h_bad & 2u) == 0, "pgh_graph_from_csr: a column index lies outside [0, n_cols)");
        }
        if (extra_diag > 0 && n_rows > 0) {
            const int64_t nnz2 = nnz + (int64_t)extra_diag * n_rows;
            DevBuf<int64_t> ip2;
            DevBuf<int32_t> idx2;
            DevBuf<double> data2;
            PGH_TRY(ip2.alloc(n_rows + 1));
            PGH_TRY(idx2.alloc(nnz2));
            PGH_TRY(data2.alloc(nnz2));
            // (the laplacian's identity rides along as a zero weight: no sum sees it; k_laplacian_ones makes it the 1 after the scaling)
            k_augment_rows<<<blocks_for(n_rows * 64), kBlock, 0, r.stream>>>(d_indptr.p, d_indices.p, d_data.p, n_rows, extra_diag,
                                                                             self_loops != 0.0 ? self_loops : 0.0, 0.0, ip2.p, idx2.p, data2.p);
            PGH_HIP(hipGetLastError());
            PGH_HIP(hipStreamSynchronize(r.stream));
            std::swap(d_indptr.p, ip2.p);
            std::swap(d_indices.p, idx2.p);
            std::swap(d_data.p, data2.p);
            nnz = nnz2;
            g->nnz = nnz;
        }
        build_mark("upload: host arrays to HBM, validation");
        PGH_HIP(pooled_malloc(&g->degrees, sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1)));
        PGH_HIP(pooled_malloc(&g->rowptr, sizeof(int32_t) * (size_t)(n_cols + 1)));
        PGH_HIP(pooled_malloc(&g->col, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        PGH_HIP(pooled_malloc(&g->val, sizeof(float) * (size_t)(nnz > 0 ? nnz : 1)));
        // ---- structure first: sort (col, row) keys with the entry index as payload -> rows of M^T, ascending columns
        if (nnz > 0) {
            PGH_TRY(keys_a.alloc(nnz));
            PGH_TRY(keys_b.alloc(nnz));
            PGH_TRY(idx_a.alloc(nnz));
            PGH_TRY(idx_b.alloc(nnz));
            if (by_row) {
                k_make_keys_idx<<<blocks_for(n_rows * 64), kBlock, 0, r.stream>>>(d_indptr.p, d_indices.p, n_rows, keys_a.p, idx_a.p);
            } else {
                // the row of every entry of the caller's CSR(M): heads scattered, a running maximum spreads them
                PGH_TRY(row_tmp.alloc(2 * (size_t)nnz));
                row_of = row_tmp.p + nnz;
                PGH_HIP(hipMemsetAsync(row_tmp.p, 0, sizeof(int32_t) * (size_t)nnz, r.stream));
                k_row_heads64<<<blocks_for(n_rows), kBlock, 0, r.stream>>>(d_indptr.p, n_rows, row_tmp.p);
                size_t scan_bytes = 0;
                PGH_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, row_tmp.p, row_of, hipcub::Max(), (int)nnz, r.stream));
                DevBuf<char> scan_temp;
                PGH_TRY(scan_temp.alloc(scan_bytes));
                PGH_HIP(hipcub::DeviceScan::InclusiveScan(scan_temp.p, scan_bytes, row_tmp.p, row_of, hipcub::Max(), (int)nnz, r.stream));
                k_make_keys_idx_flat<<<blocks_for(nnz), kBlock, 0, r.stream>>>(row_of, d_indices.p, nnz, keys_a.p, idx_a.p);
                PGH_HIP(hipStreamSynchronize(r.stream));
            }
            PGH_HIP(hipGetLastError());
            int bits_col = 1;
            while ((1LL << bits_col) < n_cols) ++bits_col;
            const int end_bit = 32 + bits_col;
            size_t temp_bytes = 0;
            PGH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_a.p, keys_b.p, idx_a.p, idx_b.p, (int)nnz, 0, end_bit, r.stream));
            DevBuf<char> temp;
            PGH_TRY(temp.alloc(temp_bytes));
            PGH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_a.p, keys_b.p, idx_a.p, idx_b.p, (int)nnz, 0, end_bit, r.stream));
            k_split_keys<<<blocks_for(nnz), kBlock, 0, r.stream>>>(keys_b.p, nnz, n_cols, g->col, g->rowptr);
            PGH_HIP(hipGetLastError());
        } else {
            k_fill_i32<<<blocks_for(n_cols + 1), kBlock, 0, r.stream>>>(g->rowptr, n_cols + 1, 0);
            PGH_HIP(hipGetLastError());
        }
        build_mark("upload: transposition (sort by column)");
        // ---- scale vectors
        bool have_left = left != nullptr, have_right = right != nullptr;
        if (factored && device_norm >= 0) {
            // to_sparse_matrix's normalisations evaluated in HBM (preprocessing.py:109-138): left from the row sums of W,
            // right from its column sums (rows of the transposed structure); zero-degree rows / columns stay zero
            have_left = device_norm == PGH_NORM_COL || device_norm == PGH_NORM_SYMMETRIC || device_norm == PGH_NORM_BOTH || laplacian;
            have_right = device_norm == PGH_NORM_SYMMETRIC || device_norm == PGH_NORM_BOTH || laplacian;
            const int take_sqrt = device_norm == PGH_NORM_SYMMETRIC || laplacian;
            if (have_left) {
                PGH_TRY(d_left.alloc(n_rows));
                if (n_rows > 0) {
                    if (row_of == nullptr) k_row_sums_f64<int64_t><<<blocks_for(n_rows * 64), kBlock, 0, r.stream>>>(d_indptr.p, nullptr, d_data.p, n_rows, d_left.p);
                    else PGH_TRY((sums_by_row<double, double>(row_of, d_data.p, nnz, n_rows, d_left.p)));
                    k_inv_nonzero<<<blocks_for(n_rows), kBlock, 0, r.stream>>>(d_left.p, n_rows, take_sqrt);
                }
            }
            if (have_right) {
                PGH_TRY(d_right.alloc(n_cols));
                if (n_cols > 0) {
                    if (row_of == nullptr) {
                        k_row_sums_f64<int32_t><<<blocks_for(n_cols * 64), kBlock, 0, r.stream>>>(g->rowptr, idx_b.p, d_data.p, n_cols, d_right.p);
                    } else {
                        // column sums = row sums of the transposed structure: the weights in its order, keyed by the sorted keys' high words
                        DevBuf<double> data_t;
                        PGH_TRY(data_t.alloc(nnz));
                        k_gather_f64<<<blocks_for(nnz), kBlock, 0, r.stream>>>(idx_b.p, d_data.p, nnz, data_t.p);
                        hipcub::TransformInputIterator<int32_t, HiWord32, const uint64_t*> cols(keys_b.p, HiWord32());
                        PGH_TRY((sums_by_row<double, double>(cols, data_t.p, nnz, n_cols, d_right.p)));
                    }
                    k_inv_nonzero<<<blocks_for(n_cols), kBlock, 0, r.stream>>>(d_right.p, n_cols, take_sqrt);
                }
            }
            PGH_HIP(hipGetLastError());
        } else if (factored) {
            if (have_left) {
                PGH_TRY(d_left.alloc(n_rows));
                PGH_HIP(hipMemcpyAsync(d_left.p, left, sizeof(double) * n_rows, hipMemcpyHostToDevice, r.stream));
            }
            if (have_right) {
                PGH_TRY(d_right.alloc(n_cols));
                PGH_HIP(hipMemcpyAsync(d_right.p, right, sizeof(double) * n_cols, hipMemcpyHostToDevice, r.stream));
            }
        }
        // ---- values: data[k] <- (left[row] * w[k]) * right[col] in place (original order), integer-weight census
        bool value_free = false;
        if (factored) {
            PGH_TRY(d_mult.alloc(nnz));
            PGH_TRY(stats.alloc(2));
            PGH_HIP(hipMemsetAsync(stats.p, 0, sizeof(unsigned long long) * 2, r.stream));
            if (n_rows > 0 && nnz > 0) {
                if (row_of == nullptr)
                    k_apply_factors<<<blocks_for(n_rows * 64), kBlock, 0, r.stream>>>(d_indptr.p, d_indices.p, d_data.p, have_left ? d_left.p : nullptr,
                                                                                     have_right ? d_right.p : nullptr, n_rows, d_mult.p, stats.p);
                else
                    k_apply_factors_flat<<<blocks_for(nnz), kBlock, 0, r.stream>>>(row_of, d_indices.p, d_data.p, have_left ? d_left.p : nullptr,
                                                                                  have_right ? d_right.p : nullptr, nnz, d_mult.p, stats.p);
            }
            unsigned long long h[2] = {0, 0};
            PGH_HIP(hipMemcpyAsync(h, stats.p, sizeof(h), hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipStreamSynchronize(r.stream));
            // value-free when every weight is a small positive integer and expanding the multiplicities costs < 25 % entries
            value_free = nnz > 0 && h[0] == 0 && (double)h[1] <= 1.25 * (double)nnz + 1024.0;
            const char* vf = getenv("PGH_VALUES");
            if (vf != nullptr && atoi(vf) != 0) value_free = false;
        }
        if (laplacian && n_rows > 0) {                        // M = I - N: the values negated, the placeholders of the diagonal become 1
            k_laplacian_finish<<<blocks_for(nnz), kBlock, 0, r.stream>>>(d_indptr.p, d_data.p, n_rows, nnz);
            k_laplacian_ones<<<blocks_for(n_rows), kBlock, 0, r.stream>>>(d_indptr.p, d_data.p, n_rows);
            value_free = false;
        }
        if (n_rows > 0) {
            if (row_of == nullptr) k_row_sums<<<blocks_for(n_rows * 64), kBlock, 0, r.stream>>>(d_indptr.p, d_data.p, n_rows, g->degrees);
            else PGH_TRY((sums_by_row<double, float>(row_of, d_data.p, nnz, n_rows, g->degrees)));
        }
        if (part == nullptr && n_rows > 0 && n_rows == n_cols && pooled_malloc(&g->src_counts, sizeof(unsigned int) * (size_t)n_rows) == hipSuccess) {
            // the relabelling key of the images (bsf_build), from the rows of the caller's matrix: no histogram over the entries
            g->src_counts_weighted = value_free;
            if (value_free && row_of != nullptr)
                PGH_TRY((sums_by_row<int, unsigned int>(row_of, d_mult.p, nnz, n_rows, g->src_counts)));
            else
                k_row_counts<<<blocks_for(value_free ? n_rows * 64 : n_rows), kBlock, 0, r.stream>>>(d_indptr.p, value_free ? d_mult.p : nullptr, n_rows,
                                                                                                  g->src_counts);
        }
        if (nnz > 0) {
            if (value_free) PGH_TRY(mult_t.alloc(nnz));
            k_gather_vals<<<blocks_for(nnz), kBlock, 0, r.stream>>>(idx_b.p, d_data.p, value_free ? d_mult.p : nullptr, nnz, g->val,
                                                                    value_free ? mult_t.p : nullptr);
            PGH_HIP(hipGetLastError());
        }
        PGH_HIP(hipStreamSynchronize(r.stream));
        build_mark("upload: normalisation, values of CSR(M^T)");
        PGH_TRY(finish_graph(g));
        build_mark("row-major tile table");
        // the layout the propagation kernels stream (PGH_FORMAT=csr keeps only the row-major merge-path route)
        const char* fmt = getenv("PGH_FORMAT");
        if (fmt == nullptr || std::string(fmt) != "csr") {
            const char* rl = getenv("PGH_RELABEL");
            const bool relabel = part == nullptr && (rl == nullptr || atoi(rl) != 0);
            const int force_blocks = part != nullptr ? part->num_blocks : 0;
            if (part != nullptr) {
                g->row_begin = part->row_begin;
                PGH_HIP(pooled_malloc(&g->part_perm, sizeof(int32_t) * (size_t)(n_rows > 0 ? n_rows : 1)));
                PGH_HIP(hipMemcpyAsync(g->part_perm, part->perm, sizeof(int32_t) * (size_t)n_rows, hipMemcpyHostToDevice, r.stream));
                PGH_HIP(hipStreamSynchronize(r.stream));
            }
            if (value_free) {
                // M^T = diag(right) * W^T * diag(left): the source scale is `left` (rows of M), the output scale `right`
                if (have_left) {
                    PGH_HIP(pooled_malloc(&g->keep_src, sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1)));
                    k_f64_to_f32_g<<<blocks_for(n_rows), kBlock, 0, r.stream>>>(d_left.p, g->keep_src, n_rows);
                }
                if (have_right) {
                    PGH_HIP(pooled_malloc(&g->keep_dst, sizeof(float) * (size_t)(n_cols > 0 ? n_cols : 1)));
                    k_f64_to_f32_g<<<blocks_for(n_cols), kBlock, 0, r.stream>>>(d_right.p, g->keep_dst, n_cols);
                }
                g->keep_mult = mult_t.release();
                PGH_TRY(bsf_build(g, nullptr, g->keep_mult, g->keep_src, g->keep_dst, relabel, force_blocks));
            } else {
                PGH_TRY(bsf_build(g, g->val, nullptr, nullptr, nullptr, relabel, force_blocks));
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

}  // namespace

extern "C" int pgh_graph_from_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                                  const int32_t* indices, const double* data, int flags, pgh_graph_t* out) {
    (void)flags;
    return graph_from_csr_impl(n_rows, n_cols, nnz, indptr, indices, data, false, nullptr, nullptr, -1, out);
}

// Row-partitioned upload of a caller's graph (the synthetic counterpart is pgh_graph_rmat_part).
extern "C" int pgh_graph_from_csr_part(int64_t n_rows, int64_t n_cols_local, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                                       const double* data, int64_t row_begin, int32_t num_blocks, const int32_t* perm,
                                       pgh_graph_t* out) {
    PGH_CHECK(perm != nullptr, "pgh_graph_from_csr_part: null permutation");
    PGH_CHECK(num_blocks == 1 || num_blocks == 2 || num_blocks == 4 || num_blocks == 8, "pgh_graph_from_csr_part: 1, 2, 4 or 8 column blocks");
    PGH_CHECK(n_rows % num_blocks == 0, "pgh_graph_from_csr_part: the (padded) id space must be a multiple of the column-block count");
    PGH_CHECK(row_begin >= 0 && row_begin + n_cols_local <= n_rows, "pgh_graph_from_csr_part: slice outside the id space");
    const char* fmt = getenv("PGH_FORMAT");
    PGH_CHECK(fmt == nullptr || std::string(fmt) != "csr", "pgh_graph_from_csr_part: partitioned graphs need the blocked layout");
    PartSpec part{row_begin, num_blocks, perm};
    return graph_from_csr_impl(n_rows, n_cols_local, nnz, indptr, indices, data, false, nullptr, nullptr, -1, out, &part);
}

extern "C" int pgh_graph_from_factored_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                                           const int32_t* indices, const double* weights, const double* left,
                                           const double* right, int flags, pgh_graph_t* out) {
    (void)flags;
    PGH_CHECK(weights != nullptr || nnz == 0, "pgh_graph_from_factored_csr: null weights");
    return graph_from_csr_impl(n_rows, n_cols, nnz, indptr, indices, weights, true, left, right, -1, out);
}

extern "C" int pgh_graph_from_adjacency(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                                        const int32_t* indices, const double* weights, int32_t normalization, int flags,
                                        pgh_graph_t* out) {
    (void)flags;
    PGH_CHECK(normalization == PGH_NORM_COL || normalization == PGH_NORM_SYMMETRIC || normalization == PGH_NORM_NONE ||
                  normalization == PGH_NORM_BOTH, "pgh_graph_from_adjacency: unknown normalization");
    PGH_CHECK(normalization == PGH_NORM_NONE || normalization == PGH_NORM_COL || n_rows == n_cols,
              "pgh_graph_from_adjacency: symmetric / both normalisation needs a square adjacency");
    return graph_from_csr_impl(n_rows, n_cols, nnz, indptr, indices, weights, true, nullptr, nullptr, normalization, out);
}

extern "C" int pgh_graph_from_adjacency_ex(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                                           const double* weights, int32_t normalization, double self_loops, int flags, pgh_graph_t* out) {
    (void)flags;
    PGH_CHECK(normalization >= PGH_NORM_COL && normalization <= PGH_NORM_LAPLACIAN, "pgh_graph_from_adjacency_ex: unknown normalization");
    PGH_CHECK(normalization == PGH_NORM_NONE || normalization == PGH_NORM_COL || n_rows == n_cols,
              "pgh_graph_from_adjacency_ex: symmetric / both / laplacian normalisation needs a square adjacency");
    return graph_from_csr_impl(n_rows, n_cols, nnz, indptr, indices, weights, true, nullptr, nullptr, normalization, out, nullptr, self_loops);
}

extern "C" int pgh_graph_destroy(pgh_graph_t g) {
    if (!g) return 0;
    if (rt().initialised) (void)hipStreamSynchronize(rt().stream);
    (void)pooled_free(g->rowptr);
    (void)pooled_free(g->col);
    (void)pooled_free(g->val);
    (void)pooled_free(g->degrees);
    (void)pooled_free(g->tile_coord);
    (void)pooled_free(g->chain_first);
    (void)pooled_free(g->tail_carry);
    (void)pooled_free(g->head_partial);
    bsf_destroy(g->bsf);
    bsf_destroy(g->bsf_mm);
    bsf_destroy(g->bsf64);
    (void)pooled_free(g->keep_mult);
    (void)pooled_free(g->src_counts);
    (void)pooled_free(g->keep_src);
    (void)pooled_free(g->keep_dst);
    (void)pooled_free(g->part_perm);
    delete g;
    return 0;
}

extern "C" int pgh_graph_info(pgh_graph_t g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz, int64_t* device_bytes) {
    PGH_CHECK(g, "pgh_graph_info: null graph");
    if (n_rows) *n_rows = g->n_rows;
    if (n_cols) *n_cols = g->n_cols;
    if (nnz) *nnz = g->nnz;
    if (device_bytes) *device_bytes = g->device_bytes;
    return 0;
}

extern "C" int pgh_graph_perm(pgh_graph_t g, int32_t* new_to_old, int64_t* row_begin) {
    PGH_CHECK(g && new_to_old, "pgh_graph_perm: null argument");
    if (row_begin) *row_begin = g->row_begin;
    if (g->part_perm != nullptr) {
        PGH_HIP(hipMemcpyAsync(new_to_old, g->part_perm, sizeof(int32_t) * g->n_rows, hipMemcpyDeviceToHost, rt().stream));
        PGH_HIP(hipStreamSynchronize(rt().stream));
    } else {
        for (int64_t i = 0; i < g->n_rows; ++i) new_to_old[i] = (int32_t)i;
    }
    return 0;
}

extern "C" int pgh_graph_format(pgh_graph_t g, char* buf, int buflen) {
    PGH_CHECK(g && buf && buflen > 0, "pgh_graph_format: null argument");
    if (g->bsf.enabled) {
        const BsfFormat& f = g->bsf;
        int used = snprintf(buf, buflen, "bsf: %d column blocks x %d sources, %s, %s entries (%d B/edge), %lld entries, %d wavefront tiles, LDS hot cache %d",
                            f.num_blocks, f.blk_size, f.relabelled ? "relabelled by source count" : "original ids",
                            f.val ? "f32-valued" : "value-free", (f.val ? 4 : 0) + (f.colf16 ? 2 : 4), (long long)f.num_entries, f.num_tiles, PGH_BSF_HOT);
        if (f.pb.enabled && used > 0 && used < buflen)
            snprintf(buf + used, buflen - used, "; cold tail: propagation-blocking image in %d slices, first: %lld entries in %d chunks x %d bins%s",
                     f.pb_slices, (long long)f.pb.num_entries, f.pb.num_chunks, f.pb.num_bins, f.pb.k1_cold ? " (heavy rows stay in the stream)" : "");
    } else {
        snprintf(buf, buflen, "csr32+f32 row-major merge-path (8 B/edge), %d tiles of %d items", g->num_tiles, g->items_per_tile);
    }
    if (g->bsf64.enabled) {                                // the f64 image exists once an f64 route has run (pgh_bsf64.hip)
        const BsfFormat& f = g->bsf64;
        int used = (int)strlen(buf);
        if (used < buflen - 1)
            used += snprintf(buf + used, buflen - used, "; f64 image: %d column blocks, %lld entries (%d B/entry)", f.num_blocks, (long long)f.num_entries,
                             (f.val ? 4 : 0) + (f.colf16 ? 2 : 4));
        if (f.pb.enabled && used > 0 && used < buflen - 1)
            snprintf(buf + used, buflen - used, ", cold tail: f64 propagation-blocking image of %lld entries in %d chunks x %d bins%s",
                     (long long)f.pb.num_entries, f.pb.num_chunks, f.pb.num_bins, f.pb.k1_cold ? " (heavy rows stay in the stream)" : "");
    }
    return 0;
}

extern "C" int pgh_graph_degrees(pgh_graph_t g, pgh_vec_t out) {
    PGH_CHECK(g && out && out->n == g->n_rows, "pgh_graph_degrees: length mismatch");
    if (g->n_rows == 0) return 0;
    PGH_HIP(hipMemcpyAsync(out->data, g->degrees, sizeof(float) * g->n_rows, hipMemcpyDeviceToDevice, rt().stream));
    return 0;
}

extern "C" int pgh_graph_download(pgh_graph_t g, int64_t* indptr_t, int32_t* indices_t, float* data_t) {
    PGH_CHECK(g && indptr_t, "pgh_graph_download: null argument");
    Runtime& r = rt();
    std::string err;
    int32_t* tmp = new int32_t[g->n_cols + 1];
    hipError_t e = hipMemcpyAsync(tmp, g->rowptr, sizeof(int32_t) * (g->n_cols + 1), hipMemcpyDeviceToHost, r.stream);
    if (e == hipSuccess && g->nnz > 0 && indices_t)
        e = hipMemcpyAsync(indices_t, g->col, sizeof(int32_t) * g->nnz, hipMemcpyDeviceToHost, r.stream);
    if (e == hipSuccess && g->nnz > 0 && data_t)
        e = hipMemcpyAsync(data_t, g->val, sizeof(float) * g->nnz, hipMemcpyDeviceToHost, r.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r.stream);
    if (e == hipSuccess)
        for (int64_t i = 0; i <= g->n_cols; ++i) indptr_t[i] = tmp[i];
    delete[] tmp;
    if (e != hipSuccess) return fail(std::string("pgh_graph_download: ") + hipGetErrorString(e));
    return 0;
}

// filter_out(x, exclude) = x[exclude == 0] (specification.py:113): order-preserving stream compaction
extern "C" int pgh_filter_out(pgh_vec_t x, pgh_vec_t exclude, pgh_vec_t out, int64_t* out_len) {
    PGH_CHECK(x && exclude && out && out_len && x->n == exclude->n && out->n >= x->n, "pgh_filter_out: length mismatch");
    PGH_CHECK(x->n < 2147483647LL, "pgh_filter_out: vector too long");
    Runtime& r = rt();
    const int64_t n = x->n;
    if (n == 0) {
        *out_len = 0;
        return 0;
    }
    DevBuf<char> flags;
    DevBuf<int> d_count;
    PGH_TRY(flags.alloc(n));
    PGH_TRY(d_count.alloc(1));
    k_flags_from_exclude<<<blocks_for(n), kBlock, 0, r.stream>>>(exclude->data, n, flags.p);
    size_t temp_bytes = 0;
    PGH_HIP(hipcub::DeviceSelect::Flagged(nullptr, temp_bytes, x->data, flags.p, out->data, d_count.p, (int)n, r.stream));
    DevBuf<char> temp;
    PGH_TRY(temp.alloc(temp_bytes));
    PGH_HIP(hipcub::DeviceSelect::Flagged(temp.p, temp_bytes, x->data, flags.p, out->data, d_count.p, (int)n, r.stream));
    int count = 0;
    PGH_HIP(hipMemcpyAsync(&count, d_count.p, sizeof(int), hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    *out_len = count;
    return 0;
}

PGH_WARM_KERNEL(k_row_sums)
