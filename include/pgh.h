/*
 * pgh.h -- C-ABI of the MI355X-native propagation engine for pygrank-style graph filters.
 *
 * This is the drop-in boundary underneath pygrank's backend-module interface
 * (reference: pygrank/core/backend/specification.py:5-118, the 29 backend functions; loader
 * pygrank/core/backend/__init__.py:40-84).  Every entry point below is what a Python (ctypes) backend
 * module -- pygrank_amd/backend/hip.py, or a `pygrank/core/backend/hip.py` a maintainer adds upstream
 *  -- binds.  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; pgh_last_error() gives the message
 *     (the reference raises plain `Exception`, e.g. convergence.py:90 -- the Python side converts).
 *   - vectors are dense f32 arrays resident in HBM, addressed through opaque handles; reductions
 *     accumulate in f64 and return f64 scalars to the host (reference vectors are fp64 numpy arrays,
 *     numpy.py:34-46; fp32 device precedent: pytorch.py:63-65,113-114).
 *   - a graph handle stores CSR(M^T) (f32 values, int32 columns) because the propagation multiplies by
 *     the transpose: conv(x, M) = x @ M = M^T x (numpy.py:64-65).
 *   - all work is enqueued on one HIP stream per process (pgh_set_stream adopts an external one, e.g.
 *     torch's current stream for the RCCL row-partitioned path).  Calls that return scalars synchronise.
 */
#ifndef PGH_H
#define PGH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgh_vec_s*   pgh_vec_t;     /* dense f32 vector in HBM                                  */
typedef struct pgh_mat_s*   pgh_mat_t;     /* dense row-major f32 [n, b] slab in HBM (multi-seed batch) */
typedef struct pgh_graph_s* pgh_graph_t;   /* CSR(M^T) + merge-path tile table in HBM                  */
typedef struct pgh_timer_s* pgh_timer_t;   /* pair of HIP events on the engine stream                  */

/* ---------------------------------------------------------------- runtime ------------------------- */
/* replaces backend_init(), specification.py:9-10 */
int         pgh_init(int device_ordinal);
int         pgh_shutdown(void);
const char* pgh_last_error(void);
/* "hip:gfx950" for the product library; the host test double under oracle/ reports "host-oracle". */
const char* pgh_runtime_name(void);
int         pgh_device_count(int* count);
int         pgh_device_name(char* buf, int buflen);
int         pgh_mem_info(int64_t* free_bytes, int64_t* total_bytes);
int         pgh_set_stream(void* hip_stream);      /* NULL -> the engine's own stream */
int         pgh_sync(void);

int pgh_timer_create(pgh_timer_t* out);
int pgh_timer_destroy(pgh_timer_t t);
int pgh_timer_start(pgh_timer_t t);                /* hipEventRecord on the engine stream */
int pgh_timer_stop(pgh_timer_t t);
int pgh_timer_elapsed_ms(pgh_timer_t t, double* ms);   /* synchronises on the stop event */

/* per-kernel HIP-event profiling of the propagation kernels (bench.py roofline leg) */
enum { PGH_K_SPMV = 0, PGH_K_FIXUP = 1, PGH_K_RESIDUAL = 2, PGH_K_FINAL = 3, PGH_K_SPMM = 4, PGH_K_COMBINE = 5,
       PGH_K_PB_GATHER = 6, PGH_K_PB_ACCUM = 7,     /* propagation-blocking passes of the cold tail (pgh_pb.hip) */
       PGH_K_PACK = 8,                              /* per-destination packing of a partition's exchange (pgh_dist_pack) */
       PGH_K_COUNT = 9 };
int pgh_profile_enable(int on);
int pgh_profile_reset(void);
int pgh_profile_read(int kernel_id, int64_t* launches, double* total_ms);

/* ---------------------------------------------------------------- vectors ------------------------- */
/* to_array / repeat / copy / length, specification.py:70-95; numpy.py:34-46 */
int     pgh_vec_alloc(int64_t n, pgh_vec_t* out);
int     pgh_vec_wrap(void* device_ptr, int64_t n, pgh_vec_t* out);   /* non-owning view (torch buffer) */
int     pgh_vec_free(pgh_vec_t v);
int64_t pgh_vec_len(pgh_vec_t v);
void*   pgh_vec_ptr(pgh_vec_t v);
int     pgh_vec_h2d_f32(pgh_vec_t v, const float* host, int64_t n);
int     pgh_vec_h2d_f64(pgh_vec_t v, const double* host, int64_t n);
int     pgh_vec_d2h_f32(pgh_vec_t v, float* host, int64_t n);
int     pgh_vec_d2h_f64(pgh_vec_t v, double* host, int64_t n);
int     pgh_vec_fill(pgh_vec_t v, double value);                     /* repeat(), specification.py:86 */
int     pgh_vec_copy(pgh_vec_t dst, pgh_vec_t src);                  /* copy(),   specification.py:66 */
int     pgh_vec_get(pgh_vec_t v, int64_t i, double* out);            /* float(x[i]), signals.py:89-90 */
int     pgh_vec_set(pgh_vec_t v, int64_t i, double value);           /* x[i] = v,    signals.py:92-96 */
int     pgh_vec_scatter_set(pgh_vec_t v, const int64_t* idx, const double* val, int64_t count);

/* elementwise operator protocol of the primitive type (signals.py:114-178) */
enum { PGH_ADD = 0, PGH_SUB = 1, PGH_MUL = 2, PGH_DIV = 3, PGH_POW = 4, PGH_MAXOP = 5, PGH_MINOP = 6,
       PGH_GT = 7, PGH_GE = 8, PGH_LT = 9, PGH_LE = 10, PGH_EQ = 11, PGH_NE = 12 };
enum { PGH_ABS = 0, PGH_EXP = 1, PGH_LOG = 2, PGH_NEG = 3, PGH_SQRT = 4, PGH_SAFE_INV = 5 };
int pgh_ewise_vv(int op, pgh_vec_t a, pgh_vec_t b, pgh_vec_t out);
int pgh_ewise_vs(int op, pgh_vec_t a, double scalar, int scalar_on_left, pgh_vec_t out);
int pgh_ewise_unary(int op, pgh_vec_t a, pgh_vec_t out);            /* abs/exp/log, specification.py:25-47 */
int pgh_axpby(double a, pgh_vec_t x, double b, pgh_vec_t y, pgh_vec_t out);   /* out = a*x + b*y */
/* filter_out(x, exclude) = x[exclude == 0], specification.py:113; out must hold len(x); count returned */
int pgh_filter_out(pgh_vec_t x, pgh_vec_t exclude, pgh_vec_t out, int64_t* out_len);

/* Ordinals / Top (algorithms/postprocess/postprocess.py:163-195,246-290): out[i] = 1 + the number of entries
 * that sort before x[i] in descending order of value (ties: lower index first, as python's stable sorted(reverse=True));
 * value of the k-th largest entry (k >= 1).  One device radix sort each. */
int pgh_vec_ordinals(pgh_vec_t x, pgh_vec_t out);
/* AUC of scores against binary labels (non-zero = positive), ties at their mid-rank: what sklearn.metrics.roc_curve + auc
 * compute in the reference (measures/supervised.py:255-263), with one device sort.  *num_positive receives the number of
 * positives (the caller raises when all labels agree, :259-260). */
int pgh_auc(pgh_vec_t labels, pgh_vec_t scores, double* auc, int64_t* num_positive);
/* Threshold("gap") (algorithms/postprocess/postprocess.py:328-343): the score after the first largest relative drop of the
 * descending order; 0 when there is none. */
int pgh_vec_gap_threshold(pgh_vec_t x, double* threshold);
int pgh_vec_kth_largest(pgh_vec_t x, int64_t k, double* value);

/* reductions: sum/min/max/mean/dot, specification.py:29-43,109; f64 accumulation */
enum { PGH_SUM = 0, PGH_ABSSUM = 1, PGH_MAX = 2, PGH_MIN = 3 };
int pgh_reduce(int kind, pgh_vec_t x, double* out);
int pgh_dot(pgh_vec_t x, pgh_vec_t y, double* out);
/* convergence residuals: Mabs / L1 / MaxDifference, measures/supervised.py:93-106,133-138 */
enum { PGH_ERR_MABS = 0, PGH_ERR_L1 = 1, PGH_ERR_LINF = 2, PGH_ERR_ITERS = 3 };
int pgh_residual(int kind, pgh_vec_t a, pgh_vec_t b, double* out);

/* ---------------------------------------------------------------- dense [n, b] slabs -------------- */
/* separate_cols / combine_cols, specification.py:17-22 (NodeRanking.propagate, signals.py:225-226) */
int     pgh_mat_alloc(int64_t n, int32_t b, pgh_mat_t* out);
int     pgh_mat_free(pgh_mat_t m);
int     pgh_mat_shape(pgh_mat_t m, int64_t* n, int32_t* b);
void*   pgh_mat_ptr(pgh_mat_t m);
int     pgh_mat_h2d_f64(pgh_mat_t m, const double* host_row_major);
int     pgh_mat_d2h_f64(pgh_mat_t m, double* host_row_major);
int     pgh_mat_set_col(pgh_mat_t m, int32_t col, pgh_vec_t v);     /* combine_cols */
int     pgh_mat_get_col(pgh_mat_t m, int32_t col, pgh_vec_t v);     /* separate_cols */
/* whole-slab forms of GraphFilter.rank's prologue for a batch of columns (abstract_filters.py:52-55 per column):
 * out[j] = sum_i |m[i, j]| (host doubles), and out[i, j] = m[i, j] / (float)divisors[j] (a zero divisor copies). */
int     pgh_mat_col_abssum(pgh_mat_t m, double* out_host /* [b] */);
int     pgh_mat_div_cols(pgh_mat_t m, const double* divisors_host /* [b] */, pgh_mat_t out);
/* out[i] = sum_{j < count} m[i, j] * coeffs[j] (f64 accumulation): a closed-form filter evaluated from the stored powers
 * {(M^T)^k p} of its personalization -- the device form of the optimisation dict (abstract_filters.py:232-246), so that a
 * tuner probe (autotune/parameterized.py:135-145) costs one pass over an [n, K] slab instead of K SpMVs. */
int     pgh_mat_gemv(pgh_mat_t m, const double* coeffs_host, int32_t count, pgh_vec_t out);
/* The same for P coefficient vectors at once: out[i, q] (+)= sum_{j < count} m[i, j] * coeffs[j * probes + q] (f64
 * accumulation; probes <= out.b <= 64; accumulate != 0 adds to out instead of overwriting it: slabs of more than 64
 * powers are folded chunk by chunk).  MANY probes of an optimiser (autotune/parameterized.py:94-145: every probe is a
 * coefficient vector over the same powers) cost ONE pass over the slab. */
int     pgh_mat_gemm(pgh_mat_t m, const double* coeffs_host, int32_t count, int32_t probes, int32_t accumulate, pgh_mat_t out);
/* out[:, 0:count] = m[:, first:first+count]  /  m[:, first:first+src.b] = src  (batches wider than 64 columns) */
int     pgh_mat_get_cols(pgh_mat_t m, int32_t first, pgh_mat_t out);
int     pgh_mat_set_cols(pgh_mat_t m, int32_t first, pgh_mat_t src);

/* ---------------------------------------------------------------- graph --------------------------- */
/* scipy_sparse_to_backend(M), specification.py:70-71 (called at core/utils/preprocessing.py:144).
 * Input: host CSR of the preprocessor's normalised matrix M (n_rows x n_cols, fp64 values, as scipy
 * holds it).  The engine uploads it, computes degrees(M) (row sums, numpy.py:76-77), transposes on the
 * device into CSR(M^T) and builds the merge-path tile table.  */
enum { PGH_GRAPH_DEFAULT = 0 };
int pgh_graph_from_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                       const int32_t* indices, const double* data, int flags, pgh_graph_t* out);
/* Factored upload: M = diag(left) * W * diag(right) with W the (weighted / multi-edge) adjacency in CSR.  This is what
 * the preprocessor's "col" (left = 1 / rowsum, right = null, preprocessing.py:109-113), "symmetric" (left = rowsum^-1/2,
 * right = colsum^-1/2, :131-138) and "both" (:123-130) normalisations produce; the values of M are evaluated on the
 * device in the reference's order ((left * w) * right, fp64, one rounding to f32).  When every weight of W is a small
 * positive integer the engine stores the value-free blocked layout (4 B/edge).  Same results as pgh_graph_from_csr(M). */
int pgh_graph_from_factored_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                                const int32_t* indices, const double* weights, const double* left, const double* right,
                                int flags, pgh_graph_t* out);
/* The preprocessor's normalisation evaluated on the device: W is the raw (weighted / multi-edge)
 * adjacency as graph_to_scipy delivers it (preprocessing.py:103; weights == NULL means all ones: 4 B/edge over PCIe),
 * the engine forms the degree reductions (row sums; column sums for "symmetric" / "both"), their (square-root)
 * inverses with zero degrees left zero (preprocessing.py:109-138) and M = diag(left) W diag(right) in fp64, in the
 * reference's evaluation order.  Bit-identical to the host route for integer weights (the sums are exact); for real
 * weights the row sums are accumulated in a different order than scipy's, i.e. values agree to 1 ulp of f32. */
enum { PGH_NORM_COL = 0, PGH_NORM_SYMMETRIC = 1, PGH_NORM_NONE = 2, PGH_NORM_BOTH = 3, PGH_NORM_LAPLACIAN = 4 };
int pgh_graph_from_adjacency(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                             const double* weights, int32_t normalization, int flags, pgh_graph_t* out);
/* ... with the two steps of to_sparse_matrix that change the STRUCTURE, on the device too (SURVEY.md 8f-1 names the self-loop term):
 *   self_loops != 0   the renormalisation trick, W <- W + self_loops * I BEFORE the degree reductions (preprocessing.py:107-108);
 *   PGH_NORM_LAPLACIAN  M = I - Dl^-1/2 W Dr^-1/2 (preprocessing.py:114-122).
 * Every row of the uploaded CSR gains its diagonal entries at its end; an entry the caller's row already holds on the diagonal stays a
 * second entry of the same position (products and row sums add them; pgh_graph_download returns both).  Square adjacencies only. */
int pgh_graph_from_adjacency_ex(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                                const double* weights, int32_t normalization, double self_loops, int flags, pgh_graph_t* out);
int pgh_graph_destroy(pgh_graph_t g);
int pgh_graph_info(pgh_graph_t g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz, int64_t* device_bytes);
/* human-readable description of the layout the propagation kernels stream (bench.py reports it) */
int pgh_graph_format(pgh_graph_t g, char* buf, int buflen);
/* Where the LAST graph build of this process spent its time: "phase=ms;phase=ms;..." (wall time per phase, the engine's stream drained
 * behind each).  The reference re-normalises and re-uploads on every rank() unless assume_immutability is set
 * (pygrank/core/utils/preprocessing.py:233-287): time-to-first-rank is format build. */
int pgh_last_build_profile(char* buf, int buflen);
/* degrees(M): row sums of the un-transposed M, specification.py:105; numpy.py:76-77 */
int pgh_graph_degrees(pgh_graph_t g, pgh_vec_t out);
/* degrees of graph_dropout(M, rate): the row sums of M under the same (seed, entry) mask pgh_spmv_dropout applies -- the
 * torch backends take degrees() of the dropped matrix (pytorch.py:34-38,100-104), so AbsorbingWalks and
 * SymmetricAbsorbingRandomWalks with graph_dropout > 0 see degrees that match their convolutions. */
int pgh_graph_degrees_dropout(pgh_graph_t g, double rate, uint64_t seed, pgh_vec_t out);
/* download the stored CSR(M^T) (verification / CPU baseline hand-off) */
int pgh_graph_download(pgh_graph_t g, int64_t* indptr_t, int32_t* indices_t, float* data_t);

/* conv(signal, M) = M^T x, specification.py:97-98; numpy.py:64-65.  Pure: y is a different buffer. */
int pgh_spmv(pgh_graph_t g, pgh_vec_t x, pgh_vec_t y);

/* conv(signal, graph_dropout(M, rate)) (specification.py:13; pytorch.py:34-38, torch dropout on the edge values): entry e of
 * CSR(M^T) (the order of pgh_graph_download) survives when the high 32 bits of splitmix64(seed ^ e * 0xD6E8FEB86659FD93) are
 * >= floor(rate * 2^32) and is scaled by 1 / (1 - rate).  The mask is never materialised. */
int pgh_spmv_dropout(pgh_graph_t g, pgh_vec_t x, pgh_vec_t y, double rate, uint64_t seed);

/* ---------------------------------------------------------------- fused propagation steps --------- */
/* PageRank._formula (adhoc.py:34-36): y = alpha * x_scale * (M^T x) + (1 - alpha) * p.
 * x_scale carries the lazily applied L1 quotient of RecursiveGraphFilter._step
 * (abstract_filters.py:133-134).  sum_y (nullable) receives sum(y) and synchronises. */
int pgh_ppr_step(pgh_graph_t g, pgh_vec_t x, double x_scale, pgh_vec_t p, double alpha, pgh_vec_t y,
                 double* sum_y);
/* AbsorbingWalks._formula (adhoc.py:166-169): y = ((M^T x) * x_scale * deg + p * lam) / (lam + deg). */
int pgh_absorb_step(pgh_graph_t g, pgh_vec_t x, double x_scale, pgh_vec_t p, pgh_vec_t deg, pgh_vec_t lam,
                    pgh_vec_t y, double* sum_y);
/* ClosedFormGraphFilter._step (abstract_filters.py:215-230,248-256), one polynomial term:
 *   term_out = a * (M^T term) + b * term;  result += c * term_out;
 *   delta = sum|result_new - result_old| (kind L1/MABS) or max|.| (LINF).
 * (a, b) = (1, 0) Taylor; (2, -1) the reference's "chebyshev" recurrence for iteration > 2. */
int pgh_poly_step(pgh_graph_t g, pgh_vec_t term, pgh_vec_t term_out, double a, double b, pgh_vec_t result,
                  double c, int err_kind, double* delta);
/* quotient + residual of one recursive step (abstract_filters.py:133-134 + convergence.py:96-101):
 *   err = residual(kind, y * y_scale, x * x_scale) without materialising the scaled vectors. */
int pgh_scaled_residual(int kind, pgh_vec_t y, double y_scale, pgh_vec_t x, double x_scale, double* err);

/* ---------------------------------------------------------------- resident iterates ---------------- */
/* The backend-primitive route (pygrank/core/backend/__init__.py:59-80: the reference's filters call conv, sum, abs, -, * ... one
 * primitive at a time; PageRank._formula adhoc.py:34-36, RecursiveGraphFilter._step abstract_filters.py:126-136, the residual of
 * ConvergenceManager convergence.py:96-101) without a way in and out of the engine's relabelled id space per conv.  A resident
 * iterate is a pair of plain vectors in that id space: x_int [n_int] (the iterate; padding slots are zero) and its gather form
 * xg [n_gather] (what a step gathers from: x_int times the source scale in the image's own layout; n_gather == 0: steps gather from
 * x_int itself and every xg / yg argument below is NULL).  Elementwise arithmetic between resident vectors of one graph and their
 * reductions are the ordinary pgh_axpby / pgh_ewise_* / pgh_reduce / pgh_scaled_residual calls on x_int.
 * n_int == 0: this image has no resident form (row-major, rectangular and partitioned images) -- conv stays pgh_spmv. */
int pgh_graph_resident_len(pgh_graph_t g, int64_t* n_int, int64_t* n_gather);
/* caller ids -> the id space (one pass; + the gather form); padding slots get `hole` (0 for iterates; the absorption of mode 2 below
 * wants 1 there, so that (0 * 0 + 0 * 1) / (1 + 0) is a zero and not 0 / 0) */
int pgh_resident_in(pgh_graph_t g, pgh_vec_t x, double hole, pgh_vec_t x_int, pgh_vec_t xg);
/* the gather form of a resident vector that elementwise arithmetic produced (n_gather > 0 only) */
int pgh_resident_gather(pgh_graph_t g, pgh_vec_t x_int, pgh_vec_t xg);
/* the id space -> caller ids: y = y_int[new id of .] * factor (to_array / np.asarray of a lazy vector) */
int pgh_resident_out(pgh_graph_t g, pgh_vec_t y_int, double factor, pgh_vec_t y);
/* mode 0: y = a * M^T x (conv, numpy.py:64-65);  mode 1: y = a * M^T x + b * v (PageRank._formula with the lazily applied L1 quotient
 * folded into a);  mode 2: y = (a * M^T x * deg + v * lam) / (lam + deg) (AbsorbingWalks._formula, adhoc.py:166-169; deg_int / lam_int
 * resident, NULL in the other modes).  Writes y_int and its gather form yg; pure (no output aliases an input).  sum_y (nullable)
 * receives sum(y) and synchronises (backend.sum of the step's outcome, abstract_filters.py:133-134). */
int pgh_resident_step(pgh_graph_t g, int32_t mode, pgh_vec_t x_int, pgh_vec_t xg, double a, pgh_vec_t v_int, double b,
                      pgh_vec_t deg_int, pgh_vec_t lam_int, pgh_vec_t y_int, pgh_vec_t yg, double* sum_y);

/* ---------------------------------------------------------------- whole loops on the device ------- */
/* GraphFilter.rank's hot loop (abstract_filters.py:58-62) with ConvergenceManager semantics
 * (convergence.py:77-101) evaluated on the device: kernels of an iteration become no-ops once the
 * convergence flag is set, so the host enqueues iterations in batches without a sync per iteration. */
typedef struct {
    double  alpha;          /* PageRank / AbsorbingWalks alpha                                   */
    int32_t use_quotient;   /* RecursiveGraphFilter use_quotient (bool form)                      */
    int32_t err_kind;       /* PGH_ERR_*; ITERS = stop at max_iters without raising               */
    double  tol;            /* already max(tol, epsilon()) (convergence.py:101); tol=None -> 0    */
    int32_t max_iters;
    int32_t end_modulo;
    double  out_scale;      /* preserve_norm factor applied to the final ranks (abstract_filters.py:63-64) */
    /* GraphFilter.rank's prologue folded into the loop's first pass over the operands (recursive runs only):
     * the personalization the loop uses is p / in_norm (abstract_filters.py:55; 0 means 1), and with start_from_p != 0
     * the starting vector is that same p / in_norm (abstract_filters.py:56 without warm_start) -- `ranks` is then
     * output only. */
    double  in_norm;        /* < 0: the engine computes sum |p| itself (abstract_filters.py:52; pgh_loop_result::in_norm reports it --
                             * 0 there means "all zeros": `ranks` is then untouched and the caller hands the personalization back,
                             * abstract_filters.py:53-54) -- and out_scale < 0 stands for "times that norm" (preserve_norm)   */
    int32_t start_from_p;
    int32_t reserved;
} pgh_loop_cfg;

typedef struct {
    int32_t iterations;     /* value of ConvergenceManager.iteration at loop exit                 */
    int32_t converged;      /* 1 = tolerance met; 0 = max_iters reached (caller raises unless ITERS) */
    int32_t spmv_count;     /* SpMV launches that contributed to the result                        */
    int32_t flags;          /* bit 0: the in-kernel residual paused once and the run went on with the separate residual kernel;
                               bit 1: the run evaluated its residual inside the finish kernel; bit 2: from the first step on */
    double  last_error;     /* residual of the last executed check                                */
    double  loop_ms;        /* HIP-event time of the loop on the engine stream (the way out of the engine's id space that follows
                               is not in it and may still be RUNNING when the call returns: engine calls are ordered on the engine's
                               stream and transfers to the host synchronise; pgh_sync() before a raw pointer is used elsewhere) */
    double  in_norm;        /* the L1 norm of the personalization when the run computed it (cfg in_norm < 0), else 0 */
} pgh_loop_result;

/* pgh_ppr_run on graph_dropout(M, rate) with a fresh mask per step (abstract_filters.py:59-62; pytorch.py:34-38) as ONE device loop:
 * step k multiplies by the matrix masked with seed seed0 + k - 1 -- the mask of pgh_spmv_dropout, evaluated inside the step's
 * kernels on whichever layout the graph carries (blocked stream + cold image: one index word per entry, built on first use). */
int pgh_ppr_run_dropout(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, double rate, uint64_t seed0,
                        pgh_loop_result* res);
/* pgh_ppr_run with f64 STORAGE (iterates, sums, quotient, residual in f64 on the blocked f64 image; p and ranks stay f32 vectors): the
 * reference's numpy backend is fp64 (epsilon() = finfo(float64).eps, pygrank/core/backend/numpy.py:84-86) and its tests run tol = 1e-9
 * (tests/test_filters.py:189,194), which the f32 engine clamps at fp32 eps.  cfg->tol is used as given (the caller passes max(tol, fp64 eps));
 * the L1 / Mabs / max rules and "iters"; an exactness mode (one host look per step), not a fast one.  Square graphs with the blocked layout. */
int pgh_ppr_run_f64(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res);
/* ... and the other recursive filters with f64 iterates, sums, quotient and residual on the same f64 image (round 6): AbsorbingWalks
 * (adhoc.py:157-169; the reference's default alpha = 1 - 1e-6 with tol = 1e-9 needs them) and SymmetricAbsorbingRandomWalks
 * (adhoc.py:348-364).  The row weights of their formulas are formed in f64 from the graph's f32 degrees.  pgh_poly_run(chebyshev = 2)
 * is the taylor form of the closed-form filters with f64 terms and accumulator. */
int pgh_absorb_run_f64(pgh_graph_t g, pgh_vec_t p, pgh_vec_t lam, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res);
int pgh_sarw_run_f64(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res);
/* ranks: in = starting vector (copy of p or warm_start), out = final ranks. */
int pgh_ppr_run(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res);
int pgh_absorb_run(pgh_graph_t g, pgh_vec_t p, pgh_vec_t lam, pgh_vec_t ranks, const pgh_loop_cfg* cfg,
                   pgh_loop_result* res);
/* SymmetricAbsorbingRandomWalks (adhoc.py:317-369): ranks <- conv(ranks / a, M) * deg / (a + deg) + p * a / (a + deg) with
 * deg = degrees(M) and a = (1 + sqrt(1 + 4 deg)) / 2 (adhoc.py:348-353,362-364), then the L1 quotient of
 * RecursiveGraphFilter._step; ranks in/out like pgh_ppr_run.  Graphs with the blocked layout (every uploaded / generated
 * graph unless PGH_FORMAT=csr). */
int pgh_sarw_run(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res);
/* closed-form run with host-computed coefficient schedule c_1..c_K (adhoc.py:83-84,113-116;
 * low_pass.py:23-26): iteration it uses coeffs[it-1], 0 beyond K.  chebyshev != 0 selects the
 * reference's recurrence (abstract_filters.py:216-224). */
int pgh_poly_run(pgh_graph_t g, pgh_vec_t p, const double* coeffs, int32_t num_coeffs, int32_t chebyshev,
                 pgh_vec_t result, const pgh_loop_cfg* cfg, pgh_loop_result* res);

/* The terms of the reference's "chebyshev" recurrence (abstract_filters.py:216-224) as f32 columns of a slab:
 * out[:, first_col + j] = T_{skip + 1 + j} for j < count, with T_1 = p, T_2 = M^T p, T_k = 2 M^T T_{k-1} - T_{k-1}, evaluated in f64
 * from T_1 on.  What optimization_dict (abstract_filters.py:232-246) stores for this form: a filter is then one pgh_mat_gemv /
 * pgh_mat_gemm over the slab.  chebyshev must be non-zero (the taylor form's terms are plain powers: pgh_spmv). */
int pgh_poly_terms(pgh_graph_t g, pgh_vec_t p, int32_t chebyshev, int32_t skip, int32_t count, pgh_mat_t out, int32_t first_col);

/* ---------------------------------------------------------------- multi-seed batch (SpMM) --------- */
/* Y = M^T X for a row-major [n, b] slab (b <= 64): the b conv() calls of NodeRanking.propagate
 * (signals.py:225-226) / tuner probes / sweeps in ONE pass over the adjacency. */
int pgh_spmm(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y);
/* b independent PageRank runs (same alpha and ConvergenceManager settings; personalizations = columns of p, already
 * L1-normalised; ranks in/out like pgh_ppr_run, output only with cfg->start_from_p).  Column j keeps its own quotient,
 * residual and stopping iteration:
 * results[j] is what pgh_ppr_run would report for seed j.  out_scales (nullable): per-column preserve_norm factor. */
int pgh_ppr_run_batch(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales,
                      pgh_loop_result* per_column_results);
/* The same two with graph_dropout(M, rate) (specification.py:13; pytorch.py:34-38) evaluated inside the batch kernel: the mask is
 * the one pgh_spmv_dropout applies -- a hash of (seed, index of the entry in CSR(M^T) order), so a multi-edge is kept or dropped
 * as a whole -- and the filters draw a fresh mask for every step (abstract_filters.py:61): step k of the batch run uses
 * seed0 + k - 1.  rate == 0 is the plain call. */
int pgh_spmm_dropout(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y, double rate, uint64_t seed);
int pgh_ppr_run_batch_dropout(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales,
                              double rate, uint64_t seed0, pgh_loop_result* per_column_results);

/* ---------------------------------------------------------------- row-partitioned step ---- */
/* The path shards with one exchange per iteration: every rank holds a contiguous slice of the rows of M^T in a
 * globally relabelled id space and a full-length gather vector; after each step the slices are all-gathered
 * (RCCL over xGMI; torch.distributed drives it on the engine stream, see pygrank_amd/distributed.py).
 * pgh_ppr_step_dist = PageRank._formula (adhoc.py:34-36) on the slice; it also writes this rank's slice of the
 * next gather vector (y * source scale), so the all-gather needs no extra pass.  No reference counterpart. */
int pgh_ppr_step_dist(pgh_graph_t g, pgh_vec_t xg_full, double x_scale, pgh_vec_t p_local, double alpha,
                      pgh_vec_t y_local, pgh_vec_t xg_local_out, double* sum_y);
int pgh_dist_prescale(pgh_graph_t g, pgh_vec_t x_local, pgh_vec_t xg_local_out);

/* Layout of the gather vector.  The source id space is cut into num_blocks column blocks of blk_size slots, hottest
 * sources first; live[b] = 1 + the highest slot of block b that an entry of THIS graph references (sources nobody
 * points at sort last: on RMAT graphs that is half of the id space or more).  A partitioned run exchanges only the
 * first L = max over ranks and blocks of live[] slots of every block and tells the engine where each block's slice
 * starts inside the (shorter) gather vector it passes to the steps below.  Defaults: bases[b] = b * blk_size. */
int pgh_graph_gather_layout(pgh_graph_t g, int32_t* num_blocks, int64_t* blk_size, int32_t* live /* [8] */);
int pgh_graph_set_gather_bases(pgh_graph_t g, const int64_t* bases /* [num_blocks] */);

/* Need lists ("grouped ncclSend/ncclRecv for exact uneven slices"; pygrank has no distributed counterpart,
 * documentation/tips.md:5-7).  A slice whose cold entries all live in the propagation-blocking image numbers its cold sources
 * COMPACTLY: block b of the gathered vector holds, from cold_bases[b] on (pgh_graph_set_gather_bases_split), the values of the
 * counts[b] cold slots (slot >= hot prefix) THIS slice references, in ascending slot order -- a rank of an 8-way partition references
 * ~43 % of the live slots.  counts all zero: the slice keeps the dense layout (small slices, PGH_DIST_NEED_LISTS=0) and
 * pgh_graph_set_gather_bases applies.  pgh_dist_need_list copies block b's slots (slot - hot) to the host; the caller sends them to
 * the block's owner once per graph.  The owner registers what its peers asked for (pgh_dist_set_send_lists: `segments` stretches,
 * destination-major, stretch k = cold slots of its local block local_block[k]) and packs its slice of the next gather vector for
 * all of them with ONE launch per step (pgh_dist_pack); the stretches travel point to point (ncclSend / ncclRecv, all_to_all).
 * pgh_dist_compact_from_dense applies a slice's own list to a dense copy of a block's cold part (communicators without
 * point-to-point transfers; single-process probes). */
int pgh_dist_need_counts(pgh_graph_t g, int64_t* counts /* [num_blocks] */);
int pgh_dist_need_list(pgh_graph_t g, int32_t block, uint32_t* out_host /* [counts[block]] */);
int pgh_dist_set_send_lists(pgh_graph_t g, const uint32_t* slots_host, const int32_t* local_block /* [segments] */,
                            const int64_t* seg_offsets /* [segments + 1] */, int32_t segments);
int pgh_dist_pack(pgh_graph_t g, pgh_vec_t xg_local, pgh_vec_t send_buf);
int pgh_dist_compact_from_dense(pgh_graph_t g, int32_t block, pgh_vec_t dense, int64_t dense_base, pgh_vec_t compact_out, int64_t out_base);

/* Device-driven partitioned loop: the scalars of the iteration stay in DEVICE memory, the collectives (RCCL
 * all-reduce, issued by the caller on the engine stream) act on them in place, and no call below synchronises with
 * the host.  `state` = 64 bytes of device memory owned by the caller, viewed as 8 doubles:
 *   d[0] scale (lazily applied L1 quotient of the current iterate)   d[1] err   d[2] sum
 *   d[3] = {int32 done, int32 steps}   d[4] = {int32 converged, int32 pad}   d[5] scale of the previous iterate
 *   d[6] residual as evaluated by the stopping rule   d[7] reserved
 * One iteration:  pgh_dist_partial -> pgh_dist_combine (d[2] = this rank's sum(y)) -> all_reduce(d[2], SUM) ->
 * pgh_dist_close_sum -> all-gather of the gather slices -> pgh_dist_residual (d[1] = this rank's residual) ->
 * all_reduce(d[1], SUM | MAX) -> pgh_dist_close_err (ConvergenceManager._has_converged, convergence.py:96-101).
 * Once `done` is set every call is a no-op on the device, so the partial sums of the next iteration can be enqueued
 * before the host has seen the flag. */
int pgh_dist_state_init(double* state);
int pgh_dist_partial(pgh_graph_t g, pgh_vec_t xg_full, const double* state);
/* pgh_dist_partial in two stages, so that the exchange of the gather vector can overlap the step: stage 1 = the block partial
 * sums, stage 2 = the cold image's phase A and the cross-tile fix-ups (stage 0 = both = pgh_dist_partial).  When
 * pgh_graph_hot_prefix reports hot_slots > 0, stage 1 reads only the first hot_slots slots of every block of the gather
 * vector (the LDS hot cache): a caller that all-gathers those first can start stage 1 while the rest is still in flight.
 * hot_slots == 0: stage 1 needs the whole gather vector. */
int pgh_dist_partial_stage(pgh_graph_t g, pgh_vec_t xg_full, const double* state, int32_t stage);
int pgh_graph_hot_prefix(pgh_graph_t g, int32_t* hot_slots);
int pgh_dist_combine(pgh_graph_t g, pgh_vec_t p_local, double alpha, pgh_vec_t y_local, pgh_vec_t xg_local_out, double* state);
/* pgh_dist_combine with the AbsorbingWalks formula (adhoc.py:157-169) on the slice's rows: deg_local / lam_local = this rank's slice of
 * degrees(M) and of absorption * (1 - alpha) / alpha. */
int pgh_dist_combine_absorb(pgh_graph_t g, pgh_vec_t p_local, pgh_vec_t deg_local, pgh_vec_t lam_local, pgh_vec_t y_local,
                            pgh_vec_t xg_local_out, double* state);
/* ... and with the step of the closed-form filters (abstract_filters.py:215-230, taylor form): term_out = a * (M^T term) + b * term,
 * result += c * term_out on the slice's rows; state[1] receives this rank's share of |result_new - result_old| (sum, or max with
 * err_linf) for the caller's all-reduce; pgh_dist_close_sum(state, 0) then counts the step, pgh_dist_close_err decides. */
int pgh_dist_combine_poly(pgh_graph_t g, pgh_vec_t term_local, pgh_vec_t term_out_local, double a, double b, pgh_vec_t result_local,
                          double c, int32_t err_linf, pgh_vec_t xg_local_out, double* state);
int pgh_dist_close_sum(double* state, int32_t use_quotient);
/* Isolated rows of a rank's slice (ids without any edge sort last in every block of a generated partition).  Between these two
 * calls the loop passes over them as long as p_local and the start iterate are zero there (checked on the device by the first
 * call); the second call restores "every row is processed".  Without them every row is always processed. */
int pgh_dist_watch_isolated(pgh_graph_t g, pgh_vec_t p_local, pgh_vec_t y_start);
int pgh_dist_release_isolated(pgh_graph_t g);
int pgh_dist_residual(int32_t kind, pgh_vec_t y_new, pgh_vec_t y_old, double* state);
int pgh_dist_close_err(double* state, int32_t kind, double tol, int64_t n_global);
/* The two parts of a block's slice in two regions of the gather vector: slots [0, hot) of block b start at hot_bases[b]
 * (read by the block partial sums: hot = what pgh_graph_hot_prefix reports), slots [hot, live) at cold_bases[b] (read by the
 * cold image's phase A).  With the regions laid out [j][rank][hot] and [j][rank][live - hot] both halves of the exchange
 * are plain all-gathers on contiguous memory.  Only for graphs whose stream is hot-only (hot prefix > 0). */
int pgh_graph_set_gather_bases_split(pgh_graph_t g, const int64_t* hot_bases, const int64_t* cold_bases /* [num_blocks] each */);

/* ---- the whole row-partitioned PageRank run behind ONE call: the engine drives RCCL itself (csrc/pgh_dist.hip) -------------
 * One process per GPU.  A communicator wraps two RCCL communicators (gather-vector exchange / scalar reductions; one when
 * num_ids == 1), three HIP streams (compute / exchange / scalars; one with PGH_DIST_SINGLE_STREAM=1) and the run's buffers.
 * Rank 0 draws num_ids ids with pgh_comm_unique_id and hands the bytes to every rank by any means (pygrank_amd/distributed.py
 * broadcasts them with torch.distributed); every rank then calls pgh_comm_create (collective).
 * pgh_dist_ppr_run = GraphFilter.rank + RecursiveGraphFilter._step + ConvergenceManager (abstract_filters.py:44-65,126-136;
 * convergence.py:77-101) for PageRank(alpha) on this rank's slice of a partitioned graph (pgh_graph_rmat_part /
 * pgh_graph_from_csr_part): p_local in, ranks_local out (new id space, this rank's rows).  Collective; host waits are bounded
 * by PGH_DIST_TIMEOUT_S (default 600 s) and end in an error return.  No reference counterpart. */
#define PGH_COMM_ID_BYTES 128
typedef struct pgh_comm_s* pgh_comm_t;
typedef struct {
    double  alpha;
    double  tol;            /* already max(tol, epsilon), convergence.py:101                                  */
    int64_t n_global;       /* Mabs divides by the global number of nodes                                      */
    int32_t err_kind;       /* PGH_ERR_*                                                                       */
    int32_t max_iters;
    int32_t end_modulo;
    int32_t use_quotient;
    int32_t preserve_norm;
    int32_t every_row;      /* 1: no row of the slice is passed over (an absorption of 0 on an isolated node is 0 / 0 in the reference) */
    /* AbsorbingWalks (adhoc.py:157-169) instead of PageRank when both are set: this rank's slice of degrees(M) and of
     * absorption * (1 - alpha) / alpha; the step is pgh_dist_combine_absorb */
    pgh_vec_t deg_local;
    pgh_vec_t lam_local;
} pgh_dist_cfg;
typedef struct {
    int32_t iterations;     /* ConvergenceManager.iteration at loop exit (0: the personalization is all zeros)  */
    int32_t spmv_count;
    int32_t converged;
    int32_t column_blocks;
    int32_t split_regions;  /* 1 = hot prefixes and cold parts are exchanged as two contiguous regions          */
    int32_t flags;          /* bit 1: the residual was evaluated inside the finish kernel (one 4-scalar all-reduce per
                             * iteration); bit 0: ... and one step had to be re-evaluated by the separate kernel; bit 2: the
                             * finish kernel ran as two launches (exchanged rows first: the exchange starts behind the first);
                             * bit 3: the cold parts travelled by need lists (point to point), bit 4: by all-gather with this slice
                             * copying its referenced slots out of it (neither: the dense layout and the all-gather alone)     */
    double  last_error;
    double  loop_ms;        /* HIP-event time of the loop on the compute stream                                 */
    int64_t exchange_bytes; /* received per rank and iteration                                                  */
    int64_t gather_slots;
} pgh_dist_result;
int pgh_comm_unique_id(uint8_t* id /* [PGH_COMM_ID_BYTES] */);
int pgh_comm_create(const uint8_t* ids /* [num_ids * PGH_COMM_ID_BYTES] */, int32_t num_ids, int32_t world, int32_t rank, pgh_comm_t* out);
/* The same communicator with the collectives performed by the HOST (MPI, gloo, ...) through two callbacks: the engine still drives the
 * loop, its streams and its events.  A callback gets device pointers and the HIP stream the exchange is ordered on and must have completed
 * the exchange, in stream order on that stream, when it returns; dtype: 0 f32, 1 f64, 2 i32; op: 0 sum, 1 max; all-gather: `count` f32
 * elements per rank, received in rank order; all-reduce: in place.  Return non-zero to abort the run. */
typedef int (*pgh_allgather_fn)(void* user, const void* send_dev, void* recv_dev, int64_t count, int32_t dtype, void* hip_stream);
typedef int (*pgh_allreduce_fn)(void* user, void* buf_dev, int64_t count, int32_t dtype, int32_t op, void* hip_stream);
int pgh_comm_create_external(int32_t world, int32_t rank, pgh_allgather_fn all_gather, pgh_allreduce_fn all_reduce, void* user, pgh_comm_t* out);
/* ... and, optionally, the point-to-point exchange of the need lists (pgh_dist_need_counts): 4-byte elements, rank r's stretch of `send_dev`
 * starts at send_offs[r] and holds send_counts[r] elements, what rank r sent lands at recv_offs[r] (recv_counts[r] elements); the stretch
 * a rank sends to itself is included.  Without it a host-collective communicator keeps the all-gather and compact slices copy their
 * slots out of it (pgh_dist_compact_from_dense). */
typedef int (*pgh_alltoallv_fn)(void* user, const void* send_dev, const int64_t* send_counts, const int64_t* send_offs, void* recv_dev,
                                const int64_t* recv_counts, const int64_t* recv_offs, void* hip_stream);
int pgh_comm_set_alltoallv(pgh_comm_t c, pgh_alltoallv_fn all_to_all_v);
int pgh_comm_destroy(pgh_comm_t comm);
int pgh_dist_ppr_run(pgh_graph_t g, pgh_comm_t comm, pgh_vec_t p_local, pgh_vec_t ranks_local, const pgh_dist_cfg* cfg,
                     pgh_dist_result* res);
/* ClosedFormGraphFilter (abstract_filters.py:152-270, taylor form: HeatKernel / PageRankClosed / GenericGraphFilter) on a partition,
 * the whole run behind one call: result = sum_k coeffs[k - 1] (M^T)^(k-1) p / |p|_1 on this rank's rows, stopped by
 * ConvergenceManager on the change of the result (convergence.py:77-101; cfg->alpha / use_quotient / every_row / deg / lam unused).
 * One all-gather of the term's gather slice and one 8-byte all-reduce of the change per term; the stopping rule runs on the device. */
int pgh_dist_poly_run(pgh_graph_t g, pgh_comm_t comm, pgh_vec_t p_local, const double* coeffs, int32_t num_coeffs, pgh_vec_t result_local,
                      const pgh_dist_cfg* cfg, pgh_dist_result* res);
/* Upper bound, in seconds, of every host wait on a collective from now on (<= 0: back to PGH_DIST_TIMEOUT_S, default 600 s): a first
 * run on new hardware can be probed with a short one (pygrank_amd/distributed.py does, before it trusts the engine's loop). */
int pgh_dist_set_timeout(double seconds);

/* new id -> original id of a relabelled (partitioned) graph, and the first row this graph holds */
int pgh_graph_perm(pgh_graph_t g, int32_t* new_to_old, int64_t* row_begin);

/* ---------------------------------------------------------------- synthetic workload -------------- */
/* Graph500-style RMAT generator + normalisation on the device (build-side addition, SURVEY.md 8d: the
 * reference has no generator).  Produces exactly the edges of oracle/rmat_np.py (integer-exact hash), sums
 * duplicate edges into weights and keeps self-loops (fastgraph.py:77-78 coo->csr semantics), then applies the
 * preprocessor's normalisation on the GPU: 0 = "col" (preprocessing.py:109-113), 1 = "symmetric"
 * (preprocessing.py:131-138), 2 = "none".  symmetrize != 0 builds A + A^T.  Only rows [row_begin, row_end) of
 * M^T are kept (1-D row partition; row_end <= 0 means all): the graph then maps a full-length
 * vector to the slice's outputs. */
int pgh_graph_rmat(int32_t scale, int32_t edge_factor, double a, double b, double c, uint64_t seed,
                   int32_t normalization, int32_t symmetrize, int64_t row_begin, int64_t row_end, pgh_graph_t* out);
/* Row-partitioned variant: ids are relabelled by descending source count into max(part_count, auto) hot-first
 * blocks (every rank derives the same permutation from the same edge stream); rank part_rank keeps the contiguous
 * slice [part_rank * n / part_count, (part_rank + 1) * n / part_count) of the new ids, so slices are equal-sized
 * (no padding in the all-gather) and statistically nnz-balanced.  Vectors of such a graph live in the new id space
 * (pgh_graph_perm maps back). */
int pgh_graph_rmat_part(int32_t scale, int32_t edge_factor, double a, double b, double c, uint64_t seed,
                        int32_t normalization, int32_t symmetrize, int32_t part_rank, int32_t part_count, pgh_graph_t* out);

/* Row-partitioned upload of a CALLER's graph (the counterpart of pgh_graph_rmat_part for matrices that enter through
 * scipy_sparse_to_backend, specification.py:70-71): every rank relabels the ids with the same permutation (new id ->
 * original id in `perm`, -1 for padding ids; pygrank_amd/distributed.py partition_scipy derives it from the source counts,
 * hot-first, dealt round-robin to num_blocks column blocks) and passes the columns [row_begin, row_begin + n_cols_local) of
 * the relabelled, normalised M -- its rows of M^T -- as host CSR (n_rows x n_cols_local, fp64 values).  n_rows is the padded
 * id space (a multiple of num_blocks); slices of all ranks are equal-sized.  Vectors of such a graph live in the new id
 * space (pgh_graph_perm maps back); it is driven through pgh_dist_* like the generated partitions. */
int pgh_graph_from_csr_part(int64_t n_rows, int64_t n_cols_local, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                            const double* data, int64_t row_begin, int32_t num_blocks, const int32_t* perm, pgh_graph_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PGH_H */
