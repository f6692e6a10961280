// Internal definitions shared by the engine's translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "pgh.h"

namespace pgh {

// ---------------------------------------------------------------- error plumbing
void set_error(const std::string& msg);
int  fail(const std::string& msg);

#define PGH_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            return ::pgh::fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                               ":" + std::to_string(__LINE__) + ")");                              \
        }                                                                                          \
    } while (0)

#define PGH_CHECK(cond, msg)                 \
    do {                                     \
        if (!(cond)) return ::pgh::fail(msg); \
    } while (0)

#define PGH_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

// ---------------------------------------------------------------- runtime state
struct Runtime {
    bool        initialised = false;
    int         device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;        // the stream every kernel is launched on
    int         num_cus = 256;
    // scratch for reductions: device partials + pinned host mirror for scalar results
    double*     d_partials = nullptr;    // [kMaxPartials * kPartialRegions]
    double*     d_scalars = nullptr;     // [kNumScalars]
    double*     h_scalars = nullptr;     // pinned
    // scalar results come to the host through a MAILBOX in mapped pinned memory (scalars_to_host): the last kernel of a reduction is
    // followed by a one-thread post {values, tag}, the host polls the tag -- a few microseconds instead of a D2H copy + a stream
    // synchronisation per backend.sum / residual (the backend-primitive route asks for two scalars per iteration)
    volatile unsigned long long* mail_host = nullptr;     // [8]: tag, 7 values (bit patterns of doubles)
    unsigned long long*          mail_dev = nullptr;
    unsigned long long           mail_tag = 0;
    // profiling
    bool        profiling = false;
    hipEvent_t  ev_a = nullptr, ev_b = nullptr;
    int64_t     prof_count[PGH_K_COUNT] = {0};
    double      prof_ms[PGH_K_COUNT] = {0};
};
Runtime& rt();
int ensure_init();
// h_scalars[first .. first + count) <- d_scalars[first .. first + count) once everything enqueued on the engine's stream has run
// (count <= 7); PGH_MAILBOX=0: the copy + stream synchronisation of rounds 1-5
int scalars_to_host(int first, int count);
// Where a graph build spends its time (pgh_last_build_profile, include/pgh.h): build_clock_reset() at the start of a build,
// build_mark(label) behind every phase -- it waits for the engine's stream and adds the wall time since the previous mark to `label`.
// Code objects are loaded lazily, per translation unit, by the first launch out of it: 5-25 ms each, paid inside the first graph build of
// a process (measured: 186 ms for the first scale-23 build against 87 ms for the second).  Every translation unit registers one of its
// kernels; pgh_init resolves them all (hipFuncGetAttributes), so the load happens at backend_init and not inside the first rank().
void register_warm_kernel(const void* kernel);
#define PGH_WARM_KERNEL(KERNEL)                                                                  \
    namespace {                                                                                  \
    struct WarmRegistration {                                                                    \
        WarmRegistration() { ::pgh::register_warm_kernel(reinterpret_cast<const void*>(&KERNEL)); } \
    } g_warm_registration;                                                                       \
    }
void build_clock_reset();
void build_mark(const char* label);

// Stream-ordered caching allocator for vectors, slabs and loop work buffers.  hipMalloc / hipFree synchronise the
// device and cost 100+ us each; a PageRank run allocates ~10 n-vectors.  A released block is reused by the next
// request of the same size; every user enqueues on the engine stream, so reuse is ordered after the last use.
// Idle blocks are capped (PGH_POOL_MB, default a quarter of the device memory; largest evicted first) and dropped on
// allocation failure.
int  pool_alloc(size_t bytes, void** out);
void pool_free(void* p);
void pool_trim();
// Every buffer of a graph build -- temporaries AND the images a graph keeps -- comes from that pool (round 6): hipMalloc / hipFree take
// locks of the kernel driver that other processes of the node contend for, and a build of ~150 of them met random stalls of 0.2-0.9 s in
// whatever phase allocated next (profiles/r06/build_stalls.log: 80 ms builds interleaved with 400-1000 ms ones).  From the second build of
// a graph of the same shape on -- what rank() does on every call when the caller does not promise immutability
// (pygrank/core/utils/preprocessing.py:233-287) -- no driver call is left.  pooled_free takes pointers of either origin.
inline hipError_t pooled_malloc_bytes(void** p, size_t bytes) { return pool_alloc(bytes, p) == 0 ? hipSuccess : hipErrorOutOfMemory; }
template <typename T>
inline hipError_t pooled_malloc(T** p, size_t bytes) {
    return pooled_malloc_bytes(reinterpret_cast<void**>(p), bytes);
}
inline hipError_t pooled_free(void* p) {
    pool_free(p);
    return hipSuccess;
}

// merge items (rows + nnz) per thread of a 256-thread workgroup; the tile table is built for 256 * PGH_IPT
#ifndef PGH_IPT
#define PGH_IPT 7
#endif
// blocked format: entries per lane of a wavefront tile, and the LDS hot cache (f32 entries) per workgroup
#ifndef PGH_BSF_IPT
#define PGH_BSF_IPT 8
#endif
#ifndef PGH_BSF_HOT
#define PGH_BSF_HOT 29696
#endif
// cache-policy bits of the cold gather (buffer_load aux: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef PGH_COLD_AUX
#define PGH_COLD_AUX 0
#endif
// diagnostic builds only (tools/probe_variants.py): 1 = gather from 4 KB, 2 = from 4 MB, 3 = no gather
#ifndef PGH_PROBE_GATHER
#define PGH_PROBE_GATHER 0
#endif
// physical order of a tile's entries (k_bsf_pack) and cache policy of the stream loads
#ifndef PGH_TILE_TRANSPOSE
#define PGH_TILE_TRANSPOSE 1
#endif
#ifndef PGH_STREAM_AUX
#define PGH_STREAM_AUX 2
#endif
// diagnostic builds only: bit 0 = no output stage, bit 1 = no strip writes, bit 2 = no carry stores
#ifndef PGH_PROBE_SKIP
#define PGH_PROBE_SKIP 0
#endif
// The relabelling deals the ranks (sources by descending reference count) to the B column blocks: the first `head` ranks -- the heavy
// hitters, whose entries must spread evenly over the blocks / XCDs / ranks -- one by one (new = (r % B) * blk + r / B, as rounds 1-3
// dealt every rank), the long tail in RUNS of kDealRun: 32 consecutive ranks are 32 consecutive slots of one block -- one 128-byte
// line of every internal-space vector.  Ids of equal reference count are consecutive ranks in ascending id order (stable sort), so
// the way out of the id space (dst[old] = src[new[old]]) touches a third of the lines per gather instruction that the one-by-one deal
// did (tools/permute_probe.hip: 40.4 -> 23.7 us for the gathers of the bench graph).  head is a multiple of B * kDealRun, block sizes
// are multiples of kDealRun; head >= n (partitions: kDealHeadAll) = the one-by-one deal throughout.
constexpr int kDealRun = 32;
constexpr int64_t kDealHeadAll = (int64_t)1 << 40;
__host__ __device__ inline int64_t deal_new_id(int64_t r, int B, int64_t blk, int64_t head) {
    if (r < head) return (r % B) * blk + r / B;
    const int64_t t = r - head, run = t / kDealRun;
    return (run % B) * blk + head / B + (run / B) * kDealRun + t % kDealRun;
}
// rank of slot `loc` of block b (ascending in loc), and the first slot of block b whose rank is >= `rank` (blk when there is none)
__host__ __device__ inline int64_t deal_rank_of(int b, int64_t loc, int B, int64_t head) {
    if (loc < head / B) return loc * B + b;
    const int64_t t = loc - head / B;
    return head + ((t / kDealRun) * B + b) * kDealRun + t % kDealRun;
}
inline int64_t deal_first_slot(int64_t rank, int b, int B, int64_t blk, int64_t head) {
    int64_t lo = 0, hi = blk;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (deal_rank_of(b, mid, B, head) >= rank) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}
constexpr int kMaxPartials = 4096;   // upper bound on workgroups contributing block partials
constexpr int kPartialRegions = 6;   // Runtime::d_partials: sums, deltas / residuals, and the fused residual's R, D, T (+ spare)
constexpr int kNumScalars = 64;

// RAII-less helper: time one launch with events when profiling is on.
struct ProfScope {
    int id;
    bool on;
    hipEvent_t end = nullptr;   // this scope's own end event (scopes may nest; the pending list may be drained meanwhile)
    explicit ProfScope(int kernel_id);
    ~ProfScope();
};

}  // namespace pgh

// ---------------------------------------------------------------- handle types
struct pgh_vec_s {
    float*  data = nullptr;
    int64_t n = 0;
    bool    owns = true;
};

struct pgh_mat_s {
    float*  data = nullptr;
    int64_t n = 0;
    int32_t b = 0;
};

struct pgh_timer_s {
    hipEvent_t start = nullptr, stop = nullptr;
};

// Column-blocked segment-flag format ("BSF") of CSR(M^T): the layout the propagation kernels stream.
//   * sources are (optionally) relabelled by descending reference count and dealt round-robin to B column
//     blocks, so each block's slice of the gather vector is a contiguous, hot-first range that stays resident
//     in one XCD's 4 MB L2;
//   * per block the entries are sorted by (row, col); a row segment starts at an entry whose bit 31 is set,
//     so neither row pointers nor empty rows are read by the SpMV pass;
//   * value-free when M^T = diag(dst_scale) * W * diag(src_scale) with small integer W: multiplicities are
//     stored as repeated entries (4 B/edge), the scales move into the gather vector and the epilogue.
// Propagation-blocking image of the COLD entries of the blocked stream (sources outside the LDS hot cache), see
// pgh_pb.hip: two streaming passes replace their random 4-byte gathers.
struct PbFormat {
    bool      enabled = false;
    bool      k1_cold = false;      // rows too heavy for a bin keep their cold entries in the blocked stream
    int64_t   num_entries = 0;      // cold entries in the image (multiplicities expanded)
    int       chunk = 0;            // sources per chunk (phase A keeps one chunk of the gather vector in LDS)
    int       num_chunks = 0;
    int       num_bins = 0;         // bins = runs of consecutive output rows (bounded rows and cold entries)
    int       bin_rows = 0;         // rows per bin at most (selects the phase B kernel shape)
    int       hot = 0;              // sources of every block that stay in the hot cache (not part of this image)
    int64_t   cold_prefix[9] = {0}; // first cold id of every block (cold ids number the referenced cold sources, block-major)
    // A order: [chunk][bin] runs, each padded to whole groups of 8 entries
    uint16_t* sloc = nullptr;       // [padded] source index inside its chunk
    float*    val = nullptr;        // [padded] or null (value-free)
    int32_t*  drop_edge = nullptr;  // [padded] index in CSR(M^T) order of every A-order entry (graph_dropout; bsf_ensure_edge_ids) or null
    uint32_t* dstg = nullptr;       // [padded / 8] group of B order that receives this group's values
    int       num_tasks = 0;
    int       tmp_planes = 1;       // tmp keeps the low / high quads of 64 groups in two planes (long (chunk, bin) runs) or side by side
    int       short_piece = 16384;  // phase A: pieces below this many entries run rounds of one group per lane (PGH_GATHER_SHORT)
    int64_t   avg_piece = 0;        // entries per phase A piece on average (selects the round size of k_pb_gather)
    int4*     task = nullptr;       // phase A pieces {chunk, entry_begin, entry_end, 0}: consecutive ranges of the entry stream
    int*      task_range = nullptr; // [num_tasks + 1] pieces of every phase A workgroup (equal shares of the stream)
    int4*     first_task = nullptr; // [num_tasks] task[task_range[w]]: a workgroup reads its first piece with its range
    // B order: [bin][chunk] runs (the same runs): a bin is one contiguous range
    float*    tmp = nullptr;        // [padded] gathered (and weighted) source values, written by phase A
    int4*     bin = nullptr;        // [num_bins] {first output row, rows | log2ceil(largest row's entries) << 16, first group, groups}
    int       num_split = 0;        // hub rows spread over several bins ("pieces")
    int4*     split = nullptr;      // [num_split] {row, first bin, pieces, -}
    double*   hub_part = nullptr;   // [hub items] piece sums of split hub rows (the last piece to arrive adds them in index order)
    uint32_t* amax = nullptr;       // [2] max |value| phase A wrote (bit pattern); phase B's exit tickets
    uint16_t* drow = nullptr;       // [padded] output row inside the bin (0xffff = pad entry)
    // work list of the finishing pass (k_pb_finish: phase B + the filter's epilogue in one launch): every output row
    // belongs to exactly one item.  item_a = the bin (as `bin`; rows = 0 for stretches without cold entries),
    // item_b = {first row of the epilogue range, rows, split index or -1, first bin of the split row}
    int       num_items = 0;
    int4*     item_a = nullptr;
    int4*     item_b = nullptr;
    uint32_t* hub_ticket = nullptr; // [num_split] arrival counters of the pieces of split hub rows (re-armed by the last arriver)
    // static schedule of k_pb_finish: workgroup w walks sched[sched_begin[w] .. sched_begin[w + 1]) -- item indices dealt by
    // estimated cost (longest first onto the least loaded workgroup), so that the persistent workgroups finish together
    int       sched_groups = 0;
    int*      sched = nullptr;      // [num_items]
    int*      sched_begin = nullptr;// [sched_groups + 1]
    int4*     first_a = nullptr;    // [sched_groups] item_a / item_b / index of every workgroup's FIRST static item (zeros / -1: none): read with the
    int4*     first_b = nullptr;    // kernel's other start-up words in one round trip instead of the chain slice -> schedule -> item
    int*      first_item = nullptr;
    uint32_t* work_counter = nullptr; // hand-out of the schedule's tail (one device word), or null without a tail
    int       tail_begin = 0, tail_count = 0;   // sched[tail_begin .. +tail_count): items handed out on the device
    // the cold tail of the f64 image (pgh_bsf64.hip; BsfFormat::pb64): the same orders and work list, doubles handed from A to B
    bool      f64 = false;
    double*   tmp64 = nullptr;      // [padded] gathered (and weighted) source values as doubles, group g at [8 g, 8 g + 8); `tmp` stays null
    unsigned long long* amax64 = nullptr;   // [2] bit pattern of max |value| phase A wrote; the finish kernel's exit tickets
    int64_t   device_bytes = 0;
};
constexpr int kPbMaxSlices = 8;

// column blocks of an image: at most 8 in the SpMV / multi-seed layouts, up to 64 in the f64 image (pgh_bsf64.hip)
constexpr int kMaxBlocks = 64;

// one word of the row -> segment map of the blocked SpMV layout (see BsfFormat::psum)
struct SegMeta {
    unsigned long long mask;
    int32_t            base;
    int32_t            pad;
};

struct BsfFormat {
    bool      enabled = false;
    bool      whole_graph = false;  // not a slice of a row partition (pb_plan: the middle bin shape is for whole graphs)
    PbFormat  pb;                   // first slice of the cold image (bins of the lowest rows); flags for the whole image
    PbFormat  pb_more[kPbMaxSlices - 1];   // further slices: each is run (phase A, then phase B) before the next, so that the
    int       pb_slices = 0;        // values handed from A to B are still in the L2 / Infinity Cache when B reads them
    int       num_blocks = 1;       // B in {1, 2, 4, 8}
    int       blk_size = 0;         // sources per block
    int       n_src = 0;            // length of the gather vector (rows of M)
    int       n_src_pad = 0;        // B * blk_size; slot n_src_pad is a permanent zero
    int       n_out = 0;            // outputs in the internal id space (= n_src_pad when relabelled)
    int       n_out_orig = 0;       // outputs in the caller's id space (rows of M^T held by this graph)
    bool      relabelled = false;
    int64_t   num_entries = 0;      // incl. one sentinel per block
    int64_t   num_segs = 0;
    uint32_t* colf = nullptr;       // [num_entries] SpMV layout: byte offset of the source inside its block (k_bsf_pack);
                                    // multi-seed layout: column (new id) | bit31 = first entry of a row segment
    uint16_t* colf16 = nullptr;     // hot-only SpMV streams (all cold entries in `pb`): byte offset / 2 into the LDS hot cache,
                                    // [tile][lane][8]; colf is freed then
    int32_t*  fix_seg = nullptr;    // [num_tiles] SpMV layout: segment (index into `psum`) that receives tile t's cross-tile fix-up, -1 = none
    uint8_t*  flags8 = nullptr;     // [num_tiles * 64] SpMV layout: segment-start flags of each lane's 8 entries
    float*    val = nullptr;        // [num_entries] or null (value-free)
    int32_t*  seg_row = nullptr;    // [num_segs] output row (new id) of every segment, -1 for sentinels (multi-seed layout; the
                                    // SpMV layout needs it at build time only)
    int       num_tiles = 0;
    int4*     tile = nullptr;       // [num_tiles] {entry_start, entry_count, seg_base, chain_first}
    int       tile_begin[kMaxBlocks + 1] = {0};  // tile range of every block
    double*   tail_carry = nullptr; // [num_tiles]
    double*   head_partial = nullptr;
    float*    part = nullptr;       // multi-seed layout only: per-tile head sums
    int32_t*  mm_close = nullptr;   // multi-seed layout only: closing row of every entry (k_mm_close_rows, pgh_spmm.hip)
    uint8_t*  mm_row_has = nullptr; // multi-seed layout only: 1 = the row of M^T holds entries (k_mm_mark_rows)
    // need lists of a partition slice (pgh_dist_need_counts): the cold image numbers the cold sources THIS slice references
    // compactly, block by block -- the exchange then moves those slots only, and a chunk of phase A holds referenced sources only
    bool      want_compact = false;
    uint32_t* need_idx = nullptr;   // [need_prefix[num_blocks]] slot - hot of every referenced cold slot, ascending, block-major
    int64_t   need_prefix[9] = {0}; // compact cold ids of block b: [need_prefix[b], need_prefix[b + 1]); all zero: the dense layout
    uint32_t* send_rows = nullptr;  // [send_total] local rows whose slots the peers asked for (pgh_dist_set_send_lists), destination-major
    int64_t   send_total = 0;
    unsigned long long send_stamp = 0;   // who registered the send lists: 0 = a caller (pgh_dist_set_send_lists), else the engine loop's set-up
    float*    mm_rowop = nullptr;   // multi-seed layout only: [n_out][4] {dst scale, gather scale s', 1 / s', row sum of M} per row: the epilogue's
                                    // row operands as ONE 16-byte word (k_mm_rowops, pgh_spmm.hip)
    int32_t*  mm_edge = nullptr;    // multi-seed layout, graph_dropout only: index of every stream entry in CSR(M^T) order (-1 = pad), the
                                    // argument of the dropout hash; copies of a multi-edge share one index (k_mm_edge_ids)
    // SpMV layout: block partial sums are stored COMPACTLY, one float per (block, row) segment in stream order (psum),
    // written sequentially by k_bsf_partial; the epilogue finds the segments of a row through one SegMeta word per
    // (block, 64 rows): bit r of mask = row 64 w + r has a segment in the block, base = index of the word's first segment
    float*    psum = nullptr;       // [num_segs + pad]
    double*   psum64 = nullptr;     // f64 image (pgh_bsf64.hip): the same compact partial sums in f64
    bool      want_meta = false;    // set before bsf_build: a multi-seed-style image (cold entries in the stream) that also gets `meta`
    // set before bsf_build (the f64 image, round 6): this multi-seed-style image moves its cold tail into a propagation-blocking image of
    // its own -- pb_hot sources of every block stay in the stream (the f64 hot cache), a chunk of phase A holds pb_chunk sources
    bool      pb64 = false;
    int       pb_hot = 0, pb_chunk = 0;
    SegMeta*  meta = nullptr;       // [B][meta_words]
    int64_t   meta_words = 0;       // ceil(n_out / 64)
    int32_t*  perm = nullptr;       // [n_src] new id -> old id, or null (identity)
    int32_t*  iperm = nullptr;      // [n_src] old id -> new id (square relabelled graphs: results leave by a gather)
    float*    src_scale = nullptr;  // [n_src_pad + 1] new space, or null
    float*    dst_scale = nullptr;  // [n_out] new space, or null
    float*    deg_int = nullptr;    // [n_out] row sums of M in the internal id space (square graphs; bsf_ensure_degrees, lazily)
    // square relabelled graphs: isolated ids (never referenced, empty row) sort last, slots [iso_begin[b], blk_size) of block b
    bool      has_iso = false;
    int64_t   live_nodes = -1;      // ids that are referenced or hold entries (they sort first), -1 = unknown
    int64_t   deal_head = pgh::kDealHeadAll;  // ranks below it were dealt to the blocks one by one, the others in runs (deal_new_id)
    int       iso_begin[8] = {0};
    int       iso_row_blocks = 0;   // row blocks the thresholds cover: num_blocks (square graphs) or the blocks of a rank's slice
    int32_t*  seed_list = nullptr;  // [2^16] original ids of the non-zeros of a run's operands (bsf_bring_pair), lazily allocated
    int*      seed_count = nullptr; // [2] how many there are (more than the list holds: the gather pass runs); the two words are used in turn
    int       seed_turn = 0;
    int*      iso_flag = nullptr;   // device word, set per run: 0 = the loop operands are zero on every isolated row, so those rows
                                    // stay zero and k_pb_finish / k_step_residual skip them; non-zero = they are processed
    int32_t*  live_dev = nullptr;   // [8] device copy target of `live`
    int       live[8] = {0};        // per block: 1 + the highest source slot any entry references (hot-first order puts
                                    // never-referenced sources last: they need not be exchanged or stored)
    int64_t   xg_base_cold[8] = {0}; // slot s >= hot of block b lives at xg_base_cold[b] + s (= xg_base[b] unless a partitioned run
                                    // keeps the hot prefixes and the cold parts of the blocks in two regions, pgh_graph_set_gather_bases_split)
    int64_t   xg_base[8] = {0};     // first element of every block's slice inside the gather vector (default b * blk_size;
                                    // a partitioned run lays the slices out as the trimmed all-gather delivers them)
    int32_t*  drop_edge = nullptr;  // index in CSR(M^T) order of every stream entry, laid out like `val` (graph_dropout; bsf_ensure_edge_ids)
    int       lg_live = 0, lg_hot = 0, lg_cold = -1;   // partitioned runs driven by the engine (pgh_dist.hip): the epilogue writes this rank's slice of the
                                    // next gather vector PACKED for the exchange -- [local block][lg_hot] | [local block][lg_live - lg_hot] --
                                    // instead of by row (0: by row; pgh_graph_set_gather_bases puts it back)
    int       xg_live = 0;          // > 0: the engine's own gather vector `xg` stores only the first xg_live slots of every
                                    // block (xg_base[b] = b * xg_live): a smaller cold region for the same gathers
    float*    xg = nullptr;         // [n_src_pad + 1] gather-source work buffer (new space)
    float*    tmp_out = nullptr;    // [n_out] work buffer (new space) for the single-step entry points
    int64_t   device_bytes = 0;
};

struct pgh_graph_s {
    int64_t n_rows = 0;      // rows of M   (= columns of the stored M^T)
    int64_t n_cols = 0;      // columns of M (= rows of the stored M^T = length of conv output)
    int64_t nnz = 0;
    // CSR of M^T
    int32_t* rowptr = nullptr;   // [n_cols + 1]
    int32_t* col = nullptr;      // [nnz]  (row index of M)
    float*   val = nullptr;      // [nnz]
    float*   degrees = nullptr;  // [n_rows] row sums of M
    // merge-path tile table: start coordinate (row, nnz) of every tile, plus the end sentinel
    int32_t  items_per_tile = 0;
    int32_t  num_tiles = 0;
    int2*    tile_coord = nullptr;   // [num_tiles + 1]
    int32_t* chain_first = nullptr;  // [num_tiles] first tile of the carry chain ending in tile t, or -1
    // per-tile carries of rows that span tiles (f64)
    double*  tail_carry = nullptr;   // [num_tiles]
    double*  head_partial = nullptr; // [num_tiles]
    int64_t  device_bytes = 0;
    BsfFormat bsf;
    // multi-seed (SpMM) layout: one column block, built on first use from the factors kept below
    BsfFormat bsf_mm;
    // f64 image of the "chebyshev" recurrence (pgh_bsf64.hip): 8 XCD-affine blocks, cold entries in the stream, built on first use
    BsfFormat bsf64;
    // reference count of every source (multiplicities included on value-free graphs): the relabelling key of every image.  Handed in
    // by whoever already has it (the generator: its out-degrees; an upload: row sums over the caller's CSR(M)), else counted once
    // (k_source_counts) and kept for the images built later (f64, multi-seed, dropout index words).  [n_rows] or null.
    unsigned int* src_counts = nullptr;
    bool          src_counts_weighted = false;   // counted with multiplicities (value-free images) or per entry (valued images)
    int32_t*  keep_mult = nullptr;   // [nnz] edge multiplicities of the value-free factorisation (generator graphs)
    float*    keep_src = nullptr;    // [n_rows] source scale, caller id space
    float*    keep_dst = nullptr;    // [n_cols] output scale
    // row-partitioned graphs: ids are globally relabelled, this graph holds rows [row_begin, row_begin + n_cols)
    int32_t* part_perm = nullptr;    // [n_rows] new id -> original id (same on every rank), or null
    int64_t  part_live_nodes = -1;   // generated partitions: ids with any edge (they sort first; -1 = unknown)
    int64_t  row_begin = 0;
};

// ---------------------------------------------------------------- device helpers
namespace pgh {

__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_reduce_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// Block reduction for blockDim.x == 256 (4 wavefronts).  kind: 0 sum, 1 max, 2 min.  Result valid in thread 0.
template <int KIND>
__device__ __forceinline__ double block_reduce_256(double v, double* s_scratch /* >= 4 doubles */) {
    if (KIND == 0) v = wave_reduce_sum(v);
    if (KIND == 1) v = wave_reduce_max(v);
    if (KIND == 2) v = wave_reduce_min(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_scratch[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = s_scratch[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            if (KIND == 0) r += s_scratch[w];
            if (KIND == 1) r = fmax(r, s_scratch[w]);
            if (KIND == 2) r = fmin(r, s_scratch[w]);
        }
        v = r;
    }
    return v;
}

}  // namespace pgh
