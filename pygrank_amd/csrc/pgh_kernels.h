// Shared device-side definitions of the propagation kernels (internal; not part of the C-ABI).
#pragma once
#include "pgh_common.h"

namespace pgh {

constexpr int WG = 256;

enum EpiMode { EPI_PLAIN = 0, EPI_AXPBY = 1, EPI_ABSORB = 2, EPI_POLY = 3 };

struct EpiParams {
    double       a;        // multiplies the row sum (alpha, or 1 / 2 for the polynomial recurrences)
    double       b;        // multiplies v[row]  ((1 - alpha) for PageRank, 0 / -1 for polynomial terms)
    const float* v;        // personalization p (PPR / ABSORB) or the current term (POLY with b != 0)
    const float* deg;      // ABSORB: degrees(M)
    const float* lam;      // ABSORB: absorption * (1 - alpha) / alpha
    float*       y;        // output vector
    float*       r;        // POLY: result accumulator (in place)
    double       c;        // POLY: coefficient of the new term
    int          err_linf; // POLY: delta is a max instead of a sum
    float*       xg_out;   // blocked format with a source scale: next gather source y * src_scale (else null)
    const float* src_scale;
    int          xg_blk;   // trimmed gather vector (BsfFormat::xg_live): slots per block in the id space ...
    int          xg_live;  // ... and slots per block actually stored (0 = stored in full, slot = row)
};

// position of internal id `row` inside a gather vector that keeps only the first `live` slots of every block of `blk`
// ids (never-referenced sources sort last inside a block and are not stored); -1 = not stored
__device__ __forceinline__ int xg_slot(int row, int blk, int live) {
    if (live == 0) return row;
    int b = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) b += (row >= k * blk) ? 1 : 0;
    const int loc = row - b * blk;
    return loc < live ? b * live + loc : -1;
}

// Device-resident loop state (ConvergenceManager on the device, convergence.py:77-101).
struct LoopState {
    double scale;       // lazily applied L1 quotient of the current iterate (abstract_filters.py:133-134)
    double err;         // last residual
    double sum;         // sum(y) of the last step
    int    done;        // convergence flag: once set every later kernel of the loop is a no-op
    int    steps;       // propagation steps executed so far
    int    converged;   // 1 when the tolerance was met
    int    pad;
};

struct GraphView {
    const int32_t* rowptr;
    const int32_t* col;
    const float*   val;
    const int2*    tile_coord;
    const int32_t* chain_first;
    double*        tail_carry;
    double*        head_partial;
    int            n;          // rows of M^T (= length of y)
    int            num_tiles;
};

template <int MODE>
__device__ __forceinline__ float apply_epilogue(const EpiParams& ep, float a_eff, int row, float sum,
                                                double& sum_y, double& delta) {
    float y;
    if (MODE == EPI_PLAIN) {
        y = a_eff * sum;
    } else if (MODE == EPI_AXPBY || MODE == EPI_POLY) {
        y = a_eff * sum;
        if (ep.v != nullptr) y += (float)ep.b * ep.v[row];
    } else {   // EPI_ABSORB: ((M^T x) * deg + p * lam) / (lam + deg), adhoc.py:167-168
        const float d = ep.deg[row], l = ep.lam[row];
        y = (a_eff * sum * d + ep.v[row] * l) / (l + d);
    }
    ep.y[row] = y;
    if (ep.xg_out != nullptr) {
        const int slot = xg_slot(row, ep.xg_blk, ep.xg_live);
        if (slot >= 0) ep.xg_out[slot] = y * ep.src_scale[row];
    }
    sum_y += (double)y;
    if (MODE == EPI_POLY) {
        const float r_old = ep.r[row];
        const float r_new = r_old + (float)ep.c * y;
        ep.r[row] = r_new;
        const double d = fabs((double)r_new - (double)r_old);
        delta = ep.err_linf ? fmax(delta, d) : delta + d;
    }
    return y;
}



// blocked-format entry points (pgh_bsf.hip)
// pgh_pb.hip: propagation-blocking image of the cold entries
struct PbPlan {
    int32_t* row_bin = nullptr;    // device: bin of every output row, -1 = too heavy for a bin
    int      num_bins = 0, num_chunks = 0, bin_rows = 0, heavy_row = 0;
    int64_t  entries = 0;          // cold entries that go into the image
    int      slices = 1;           // the bins are cut into `slices` consecutive groups of about equal entry counts
    int      slice_first[kPbMaxSlices + 1] = {0};   // first bin of every slice
    int64_t  slice_entries[kPbMaxSlices] = {0};
    int      num_split = 0;
    int4*    host_split = nullptr; // {row, first bin, pieces, -} of the hub rows with several bins (new[]), owned by the plan
    int4*    host_bins = nullptr;  // {first row, rows, -, entries} of every bin (new[]), owned by the plan
    bool     heavy_rows = false;   // some rows keep their cold entries in the blocked stream
};
int pb_plan(BsfFormat& f, const uint64_t* keys, int64_t E, const int* live, int hot, unsigned char* is_hot, PbPlan* plan, bool* use);
int pb_build(BsfFormat& f, PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, const int* live, int hot);
// compacts the entries of one slice (stream keys whose row belongs to bins [slice_first[s], slice_first[s + 1])) out of the cold keys
int pb_select_slice(const PbPlan* plan, int slice, const uint64_t* cold_keys, const float* cold_vals, int64_t count, uint64_t* keys_out,
                    float* vals_out, int64_t* selected);
void pb_plan_release(PbPlan* plan);
int pb_launch(pgh_graph_s* g, const float* xg, const LoopState* state);
void pb_destroy(PbFormat& p);
int bsf_launch_partial(pgh_graph_s* g, const float* xg, const LoopState* state);
template <int MODE>
int bsf_launch_combine(pgh_graph_s* g, const EpiParams& ep, const LoopState* state, int* num_partials);
template <int MODE>
int bsf_launch(pgh_graph_s* g, const EpiParams& ep, const float* xg, const LoopState* state, int* num_partials,
               hipEvent_t before_combine = nullptr);
int bsf_to_internal(pgh_graph_s* g, const float* src, float* dst, bool prescale, float hole);
bool bsf_can_bring_pair(const pgh_graph_s* g);
int bsf_bring_pair(pgh_graph_s* g, const float* v, const float* ranks, float* v_int, float* y0, bool want_xg, float in_norm,
                   bool start_from_v);
int bsf_out_to_internal(pgh_graph_s* g, const float* src, float* dst, float hole);
int bsf_to_original(pgh_graph_s* g, const float* src, float* dst, double factor);
int bsf_build(pgh_graph_s* g, const float* val, const int32_t* mult, const float* src_old, const float* dst_old, bool relabel,
              int force_blocks = 0, BsfFormat* target = nullptr);
int build_count_perm(const unsigned int* cnt, int64_t n, int B, int blk, int32_t* perm, int32_t* iperm);
int bsf_auto_blocks(int64_t n_src);
void bsf_destroy(BsfFormat& f);
int finish_graph(pgh_graph_s* g);

}  // namespace pgh
