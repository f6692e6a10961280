// TEST INFRASTRUCTURE ONLY -- host restatement of the include/pgh.h C-ABI ("test double").
//
// Purpose: lets tests/ exercise the host-side Python of pygrank_amd (backend module, signals, filters,
// convergence bookkeeping, the gloo row-partition path) in a container without a GPU, and gives the GPU
// parity tests an independent f32-storage / f64-accumulate implementation of every entry point.  It is
// never loaded by the product: pygrank_amd/_lib.py binds only csrc/libpgh_hip.so, and only
// tests call _lib._install_test_double().  pgh_runtime_name() reports "host-oracle".
//
// Each function restates the reference semantics it stands for (citations: path:line under /root/reference):
//   conv                      pygrank/core/backend/numpy.py:64-65        y = x @ M = M^T x
//   degrees                   pygrank/core/backend/numpy.py:76-77        row sums of M
//   PageRank step             pygrank/algorithms/filters/adhoc.py:34-36
//   AbsorbingWalks step       pygrank/algorithms/filters/adhoc.py:166-169
//   closed-form step          pygrank/algorithms/filters/abstract_filters.py:215-230,248-256
//   L1 quotient               pygrank/algorithms/filters/abstract_filters.py:133-134
//   stopping rule             pygrank/algorithms/convergence.py:77-101
//   residuals                 pygrank/measures/supervised.py:93-106,133-138
// Build: make -C oracle  (g++ -O2 -shared -fPIC) -> oracle/_build/libpgh_host_oracle.so
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "pgh.h"

struct pgh_vec_s {
    float* data;
    int64_t n;
    bool owns;
};
struct pgh_mat_s {
    std::vector<float> data;
    int64_t n;
    int32_t b;
};
struct pgh_timer_s {
    std::chrono::steady_clock::time_point a, b;
};
struct pgh_graph_s {
    int64_t n_rows, n_cols, nnz;
    std::vector<int64_t> rowptr;   // CSR of M^T
    std::vector<int32_t> col;
    std::vector<float> val;
    std::vector<float> degrees;
    std::vector<int32_t> part_perm;   // new id -> old id for row-partitioned graphs
    int64_t row_begin = 0;
    // gather-vector layout of partitioned graphs (pgh_graph_gather_layout / pgh_graph_set_gather_bases)
    int32_t gather_blocks = 1;
    int64_t gather_blk = 0;
    int64_t gather_base[8] = {0};
    const float* pending_xg = nullptr;
    std::vector<float> dense_x;       // gather vector expanded to the full id space
};

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return 1;
}
#define CHECK(c, m) \
    do {            \
        if (!(c)) return fail(m); \
    } while (0)

extern "C" {

int pgh_init(int) { return 0; }
int pgh_shutdown(void) { return 0; }
const char* pgh_last_error(void) { return g_err.c_str(); }
const char* pgh_runtime_name(void) { return "host-oracle"; }
int pgh_device_count(int* c) {
    *c = 0;
    return 0;
}
int pgh_device_name(char* buf, int len) {
    snprintf(buf, len, "host-oracle (test double)");
    return 0;
}
int pgh_mem_info(int64_t* f, int64_t* t) {
    *f = *t = 0;
    return 0;
}
int pgh_set_stream(void*) { return 0; }
int pgh_sync(void) { return 0; }

int pgh_timer_create(pgh_timer_t* out) {
    *out = new pgh_timer_s();
    return 0;
}
int pgh_timer_destroy(pgh_timer_t t) {
    delete t;
    return 0;
}
int pgh_timer_start(pgh_timer_t t) {
    t->a = std::chrono::steady_clock::now();
    return 0;
}
int pgh_timer_stop(pgh_timer_t t) {
    t->b = std::chrono::steady_clock::now();
    return 0;
}
int pgh_timer_elapsed_ms(pgh_timer_t t, double* ms) {
    *ms = std::chrono::duration<double, std::milli>(t->b - t->a).count();
    return 0;
}
static bool g_prof_on = false;
static int64_t g_prof_count[PGH_K_COUNT] = {0};
static double g_prof_ms[PGH_K_COUNT] = {0};
int pgh_profile_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}
int pgh_profile_reset(void) {
    for (int i = 0; i < PGH_K_COUNT; ++i) g_prof_count[i] = 0, g_prof_ms[i] = 0;
    return 0;
}
int pgh_profile_read(int id, int64_t* launches, double* ms) {
    CHECK(id >= 0 && id < PGH_K_COUNT, "pgh_profile_read: bad kernel id");
    *launches = g_prof_count[id];
    *ms = g_prof_ms[id];
    return 0;
}

// ------------------------------------------------------------------------------------------ vectors
int pgh_vec_alloc(int64_t n, pgh_vec_t* out) {
    CHECK(n >= 0, "pgh_vec_alloc: negative length");
    pgh_vec_s* v = new pgh_vec_s();
    v->data = new float[n > 0 ? n : 1]();
    v->n = n;
    v->owns = true;
    *out = v;
    return 0;
}
int pgh_vec_wrap(void* p, int64_t n, pgh_vec_t* out) {
    pgh_vec_s* v = new pgh_vec_s();
    v->data = (float*)p;
    v->n = n;
    v->owns = false;
    *out = v;
    return 0;
}
int pgh_vec_free(pgh_vec_t v) {
    if (!v) return 0;
    if (v->owns) delete[] v->data;
    delete v;
    return 0;
}
int64_t pgh_vec_len(pgh_vec_t v) { return v ? v->n : -1; }
void* pgh_vec_ptr(pgh_vec_t v) { return v ? v->data : nullptr; }
int pgh_vec_h2d_f32(pgh_vec_t v, const float* h, int64_t n) {
    CHECK(v && n == v->n, "pgh_vec_h2d_f32: length mismatch");
    std::copy(h, h + n, v->data);
    return 0;
}
int pgh_vec_h2d_f64(pgh_vec_t v, const double* h, int64_t n) {
    CHECK(v && n == v->n, "pgh_vec_h2d_f64: length mismatch");
    for (int64_t i = 0; i < n; ++i) v->data[i] = (float)h[i];
    return 0;
}
int pgh_vec_d2h_f32(pgh_vec_t v, float* h, int64_t n) {
    CHECK(v && n == v->n, "pgh_vec_d2h_f32: length mismatch");
    std::copy(v->data, v->data + n, h);
    return 0;
}
int pgh_vec_d2h_f64(pgh_vec_t v, double* h, int64_t n) {
    CHECK(v && n == v->n, "pgh_vec_d2h_f64: length mismatch");
    for (int64_t i = 0; i < n; ++i) h[i] = (double)v->data[i];
    return 0;
}
int pgh_vec_fill(pgh_vec_t v, double value) {
    std::fill(v->data, v->data + v->n, (float)value);
    return 0;
}
int pgh_vec_copy(pgh_vec_t d, pgh_vec_t s) {
    CHECK(d && s && d->n == s->n, "pgh_vec_copy: length mismatch");
    std::copy(s->data, s->data + s->n, d->data);
    return 0;
}
int pgh_vec_get(pgh_vec_t v, int64_t i, double* out) {
    CHECK(v && i >= 0 && i < v->n, "pgh_vec_get: index out of range");
    *out = v->data[i];
    return 0;
}
int pgh_vec_set(pgh_vec_t v, int64_t i, double value) {
    CHECK(v && i >= 0 && i < v->n, "pgh_vec_set: index out of range");
    v->data[i] = (float)value;
    return 0;
}
int pgh_vec_scatter_set(pgh_vec_t v, const int64_t* idx, const double* val, int64_t count) {
    for (int64_t k = 0; k < count; ++k) {
        CHECK(idx[k] >= 0 && idx[k] < v->n, "pgh_vec_scatter_set: index out of range");
        v->data[idx[k]] = (float)val[k];
    }
    return 0;
}

static float bin(int op, float a, float b) {
    switch (op) {
        case PGH_ADD: return a + b;
        case PGH_SUB: return a - b;
        case PGH_MUL: return a * b;
        case PGH_DIV: return a / b;
        case PGH_POW: return std::pow(a, b);
        case PGH_MAXOP: return std::fmax(a, b);
        case PGH_MINOP: return std::fmin(a, b);
        case PGH_GT: return a > b;
        case PGH_GE: return a >= b;
        case PGH_LT: return a < b;
        case PGH_LE: return a <= b;
        case PGH_EQ: return a == b;
        default: return a != b;
    }
}
int pgh_ewise_vv(int op, pgh_vec_t a, pgh_vec_t b, pgh_vec_t out) {
    CHECK(a && b && out && a->n == b->n && a->n == out->n, "pgh_ewise_vv: length mismatch");
    CHECK(op >= 0 && op <= PGH_NE, "unknown binary operator");
    for (int64_t i = 0; i < a->n; ++i) out->data[i] = bin(op, a->data[i], b->data[i]);
    return 0;
}
int pgh_ewise_vs(int op, pgh_vec_t a, double s, int left, pgh_vec_t out) {
    CHECK(a && out && a->n == out->n, "pgh_ewise_vs: length mismatch");
    CHECK(op >= 0 && op <= PGH_NE, "unknown binary operator");
    const float f = (float)s;
    for (int64_t i = 0; i < a->n; ++i) out->data[i] = left ? bin(op, f, a->data[i]) : bin(op, a->data[i], f);
    return 0;
}
int pgh_ewise_unary(int op, pgh_vec_t a, pgh_vec_t out) {
    CHECK(a && out && a->n == out->n, "pgh_ewise_unary: length mismatch");
    for (int64_t i = 0; i < a->n; ++i) {
        const float x = a->data[i];
        float r;
        switch (op) {
            case PGH_ABS: r = std::fabs(x); break;
            case PGH_EXP: r = std::exp(x); break;
            case PGH_LOG: r = std::log(x); break;
            case PGH_NEG: r = -x; break;
            case PGH_SQRT: r = std::sqrt(x); break;
            case PGH_SAFE_INV: r = x != 0.f ? 1.f / x : 0.f; break;
            default: return fail("unknown unary operator");
        }
        out->data[i] = r;
    }
    return 0;
}
int pgh_axpby(double a, pgh_vec_t x, double b, pgh_vec_t y, pgh_vec_t out) {
    CHECK(x && y && out && x->n == y->n && x->n == out->n, "pgh_axpby: length mismatch");
    for (int64_t i = 0; i < x->n; ++i) out->data[i] = (float)a * x->data[i] + (float)b * y->data[i];
    return 0;
}
int pgh_filter_out(pgh_vec_t x, pgh_vec_t ex, pgh_vec_t out, int64_t* out_len) {
    CHECK(x && ex && out && x->n == ex->n && out->n >= x->n, "pgh_filter_out: length mismatch");
    int64_t k = 0;
    for (int64_t i = 0; i < x->n; ++i)
        if (ex->data[i] == 0.f) out->data[k++] = x->data[i];
    *out_len = k;
    return 0;
}
int pgh_reduce(int kind, pgh_vec_t x, double* out) {
    CHECK(x && out, "pgh_reduce: null argument");
    if (x->n == 0) {
        CHECK(kind == PGH_SUM || kind == PGH_ABSSUM, "pgh_reduce: max/min of an empty vector");
        *out = 0;
        return 0;
    }
    double acc = kind == PGH_MAX ? -INFINITY : (kind == PGH_MIN ? INFINITY : 0.0);
    for (int64_t i = 0; i < x->n; ++i) {
        const double v = x->data[i];
        if (kind == PGH_SUM) acc += v;
        else if (kind == PGH_ABSSUM) acc += std::fabs(v);
        else if (kind == PGH_MAX) acc = std::fmax(acc, v);
        else if (kind == PGH_MIN) acc = std::fmin(acc, v);
        else return fail("pgh_reduce: unknown kind");
    }
    *out = acc;
    return 0;
}
int pgh_dot(pgh_vec_t x, pgh_vec_t y, double* out) {
    CHECK(x && y && x->n == y->n, "pgh_dot: length mismatch");
    double acc = 0;
    for (int64_t i = 0; i < x->n; ++i) acc += (double)x->data[i] * (double)y->data[i];
    *out = acc;
    return 0;
}
static double scaled_res(int kind, const float* y, double ys, const float* x, double xs, int64_t n) {
    double acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double d = std::fabs((double)y[i] * ys - (double)x[i] * xs);
        acc = (kind == PGH_ERR_LINF) ? std::fmax(acc, d) : acc + d;
    }
    if (kind == PGH_ERR_MABS && n > 0) acc /= (double)n;
    return acc;
}
int pgh_scaled_residual(int kind, pgh_vec_t y, double ys, pgh_vec_t x, double xs, double* err) {
    CHECK(y && x && x->n == y->n, "pgh_scaled_residual: length mismatch");
    CHECK(kind == PGH_ERR_MABS || kind == PGH_ERR_L1 || kind == PGH_ERR_LINF, "pgh_scaled_residual: unknown kind");
    *err = scaled_res(kind, y->data, ys, x->data, xs, x->n);
    return 0;
}
int pgh_residual(int kind, pgh_vec_t a, pgh_vec_t b, double* out) { return pgh_scaled_residual(kind, a, 1, b, 1, out); }

// ------------------------------------------------------------------------------------------ slabs
int pgh_mat_alloc(int64_t n, int32_t b, pgh_mat_t* out) {
    CHECK(n >= 0 && b >= 1, "pgh_mat_alloc: bad shape");
    pgh_mat_s* m = new pgh_mat_s();
    m->n = n;
    m->b = b;
    m->data.assign((size_t)n * b, 0.f);
    *out = m;
    return 0;
}
int pgh_mat_free(pgh_mat_t m) {
    delete m;
    return 0;
}
int pgh_mat_shape(pgh_mat_t m, int64_t* n, int32_t* b) {
    *n = m->n;
    *b = m->b;
    return 0;
}
void* pgh_mat_ptr(pgh_mat_t m) { return m ? m->data.data() : nullptr; }
int pgh_mat_h2d_f64(pgh_mat_t m, const double* h) {
    for (size_t i = 0; i < m->data.size(); ++i) m->data[i] = (float)h[i];
    return 0;
}
int pgh_mat_d2h_f64(pgh_mat_t m, double* h) {
    for (size_t i = 0; i < m->data.size(); ++i) h[i] = m->data[i];
    return 0;
}
int pgh_mat_set_col(pgh_mat_t m, int32_t c, pgh_vec_t v) {
    CHECK(m && v && v->n == m->n && c >= 0 && c < m->b, "pgh_mat_set_col: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i) m->data[i * m->b + c] = v->data[i];
    return 0;
}
int pgh_mat_get_col(pgh_mat_t m, int32_t c, pgh_vec_t v) {
    CHECK(m && v && v->n == m->n && c >= 0 && c < m->b, "pgh_mat_get_col: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i) v->data[i] = m->data[i * m->b + c];
    return 0;
}

int pgh_mat_col_abssum(pgh_mat_t m, double* out) {
    CHECK(m && out, "pgh_mat_col_abssum: null argument");
    for (int j = 0; j < m->b; ++j) out[j] = 0.0;
    for (int64_t i = 0; i < m->n; ++i)
        for (int j = 0; j < m->b; ++j) out[j] += std::fabs((double)m->data[i * m->b + j]);
    return 0;
}
int pgh_mat_div_cols(pgh_mat_t m, const double* div, pgh_mat_t out) {
    CHECK(m && out && div && m->n == out->n && m->b == out->b, "pgh_mat_div_cols: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i)
        for (int j = 0; j < m->b; ++j) {
            const float d = (float)div[j];
            out->data[i * m->b + j] = d != 0.f ? m->data[i * m->b + j] / d : m->data[i * m->b + j];
        }
    return 0;
}
static std::vector<int64_t> order_desc(pgh_vec_t x) {
    std::vector<int64_t> idx(x->n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return x->data[a] > x->data[b]; });
    return idx;
}
int pgh_vec_ordinals(pgh_vec_t x, pgh_vec_t out) {
    CHECK(x && out && x->n == out->n && x->data != out->data, "pgh_vec_ordinals: bad arguments");
    const std::vector<int64_t> idx = order_desc(x);
    for (int64_t k = 0; k < x->n; ++k) out->data[idx[k]] = (float)(k + 1);
    return 0;
}
int pgh_vec_kth_largest(pgh_vec_t x, int64_t k, double* value) {
    CHECK(x && value && k >= 1 && k <= x->n, "pgh_vec_kth_largest: k outside [1, n]");
    *value = x->data[order_desc(x)[k - 1]];
    return 0;
}
// AUC as the pair statistic it is (supervised.py:255-263): P(score of a positive > score of a negative) + P(equal) / 2,
// counted from the descending order with ties grouped -- written differently from the engine's mid-rank sum on purpose
int pgh_auc(pgh_vec_t labels, pgh_vec_t scores, double* auc, int64_t* num_positive) {
    CHECK(labels && scores && auc && labels->n == scores->n, "pgh_auc: bad arguments");
    const std::vector<int64_t> order = order_desc(scores);
    const int64_t n = scores->n;
    double pos_total = 0;
    for (int64_t i = 0; i < n; ++i) pos_total += labels->data[i] != 0.f ? 1.0 : 0.0;
    double wins = 0, pos_seen = 0;                      // positives strictly above the current tie group
    for (int64_t k = 0; k < n;) {
        int64_t e = k;
        double gp = 0, gn = 0;
        while (e < n && scores->data[order[e]] == scores->data[order[k]]) {
            if (labels->data[order[e]] != 0.f) gp += 1; else gn += 1;
            ++e;
        }
        wins += gn * pos_seen + 0.5 * gn * gp;
        pos_seen += gp;
        k = e;
    }
    const double neg_total = (double)n - pos_total;
    if (num_positive) *num_positive = (int64_t)pos_total;
    *auc = (pos_total > 0 && neg_total > 0) ? wins / (pos_total * neg_total) : 0.0;
    return 0;
}
int pgh_vec_gap_threshold(pgh_vec_t x, double* threshold) {              // postprocess.py:328-343, the reference's loop
    CHECK(x && threshold, "pgh_vec_gap_threshold: null argument");
    const std::vector<int64_t> order = order_desc(x);
    double max_diff = 0, thr = 0, prev = 0;
    for (int64_t k = 0; k < x->n; ++k) {
        const double v = (double)x->data[order[k]];
        if (prev > 0) {
            const double diff = (prev - v) / prev;
            if (diff > max_diff) {
                max_diff = diff;
                thr = v;
            }
        }
        prev = v;
    }
    *threshold = thr;
    return 0;
}
int pgh_mat_gemv(pgh_mat_t m, const double* c, int32_t count, pgh_vec_t out) {
    CHECK(m && out && (c || count == 0) && count >= 0 && count <= m->b && out->n == m->n, "pgh_mat_gemv: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i) {
        double acc = 0;
        for (int32_t j = 0; j < count; ++j) acc += (double)m->data[i * m->b + j] * c[j];
        out->data[i] = (float)acc;
    }
    return 0;
}
int pgh_mat_gemm(pgh_mat_t m, const double* c, int32_t count, int32_t probes, int32_t accumulate, pgh_mat_t out) {
    CHECK(m && out && (c || count == 0) && count >= 0 && count <= m->b && count <= 64 && probes >= 1 && probes <= out->b && probes <= 64 &&
              out->n == m->n, "pgh_mat_gemm: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i)
        for (int32_t q = 0; q < probes; ++q) {
            double acc = 0;
            for (int32_t j = 0; j < count; ++j) acc += (double)m->data[i * m->b + j] * c[(int64_t)j * probes + q];
            float& o = out->data[i * out->b + q];
            o = accumulate ? (float)((double)o + acc) : (float)acc;
        }
    return 0;
}
int pgh_mat_get_cols(pgh_mat_t m, int32_t first, pgh_mat_t out) {
    CHECK(m && out && m->n == out->n && first >= 0 && first + out->b <= m->b, "pgh_mat_get_cols: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i)
        for (int j = 0; j < out->b; ++j) out->data[i * out->b + j] = m->data[i * m->b + first + j];
    return 0;
}
int pgh_mat_set_cols(pgh_mat_t m, int32_t first, pgh_mat_t src) {
    CHECK(m && src && m->n == src->n && first >= 0 && first + src->b <= m->b, "pgh_mat_set_cols: shape mismatch");
    for (int64_t i = 0; i < m->n; ++i)
        for (int j = 0; j < src->b; ++j) m->data[i * m->b + first + j] = src->data[i * src->b + j];
    return 0;
}

// ------------------------------------------------------------------------------------------ graph
int pgh_graph_from_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                       const double* data, int, pgh_graph_t* out) {
    CHECK(indptr && indptr[0] == 0 && indptr[n_rows] == nnz, "pgh_graph_from_csr: indptr does not match nnz");
    for (int64_t r = 0; r < n_rows; ++r)
        CHECK(indptr[r] <= indptr[r + 1] && indptr[r] >= 0 && indptr[r + 1] <= nnz, "pgh_graph_from_csr: indptr is not non-decreasing within [0, nnz]");
    pgh_graph_s* g = new pgh_graph_s();
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->degrees.assign(n_rows, 0.f);
    g->rowptr.assign(n_cols + 1, 0);
    g->col.resize(nnz);
    g->val.resize(nnz);
    for (int64_t r = 0; r < n_rows; ++r) {                      // numpy.py:76-77
        double acc = 0;
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) acc += data[k];
        g->degrees[r] = (float)acc;
    }
    for (int64_t k = 0; k < nnz; ++k) {
        if (indices[k] < 0 || indices[k] >= n_cols) {
            delete g;
            return fail("pgh_graph_from_csr: column index out of range");
        }
        g->rowptr[indices[k] + 1]++;
    }
    for (int64_t c = 0; c < n_cols; ++c) g->rowptr[c + 1] += g->rowptr[c];
    std::vector<int64_t> cursor(g->rowptr.begin(), g->rowptr.end() - 1);
    for (int64_t r = 0; r < n_rows; ++r)                         // stable: rows ascending inside every M^T row
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
            const int64_t pos = cursor[indices[k]]++;
            g->col[pos] = (int32_t)r;
            g->val[pos] = (float)data[k];
        }
    *out = g;
    return 0;
}
int pgh_graph_from_factored_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                                const double* w, const double* left, const double* right, int flags, pgh_graph_t* out) {
    std::vector<double> data(nnz);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
            data[k] = ((left ? left[r] : 1.0) * w[k]) * (right ? right[indices[k]] : 1.0);      // preprocessing.py:113,138
    return pgh_graph_from_csr(n_rows, n_cols, nnz, indptr, indices, data.data(), flags, out);
}
int pgh_graph_from_adjacency(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                             const double* w, int32_t normalization, int flags, pgh_graph_t* out) {
    CHECK(normalization >= 0 && normalization <= 3, "pgh_graph_from_adjacency: unknown normalization");
    CHECK(normalization == PGH_NORM_NONE || normalization == PGH_NORM_COL || n_rows == n_cols,
          "pgh_graph_from_adjacency: symmetric / both normalisation needs a square adjacency");
    // preprocessing.py:109-138: degree reductions, (square-root) inverses with zero degrees left zero
    std::vector<double> weights(nnz, 1.0), left(n_rows, 0.0), right(n_cols, 0.0);
    if (w) weights.assign(w, w + nnz);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
            left[r] += weights[k];
            right[indices[k]] += weights[k];
        }
    const bool sq = normalization == PGH_NORM_SYMMETRIC;
    auto inv = [&](std::vector<double>& v) {
        for (double& x : v) {
            if (sq) x = std::sqrt(x);
            if (x != 0.0) x = 1.0 / x;
        }
    };
    inv(left);
    inv(right);
    const bool use_left = normalization != PGH_NORM_NONE;
    const bool use_right = normalization == PGH_NORM_SYMMETRIC || normalization == PGH_NORM_BOTH;
    return pgh_graph_from_factored_csr(n_rows, n_cols, nnz, indptr, indices, weights.data(), use_left ? left.data() : nullptr,
                                       use_right ? right.data() : nullptr, flags, out);
}
int pgh_graph_from_adjacency_ex(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                                const double* w, int32_t normalization, double self_loops, int flags, pgh_graph_t* out) {
    const bool laplacian = normalization == PGH_NORM_LAPLACIAN;
    if (self_loops == 0.0 && !laplacian) return pgh_graph_from_adjacency(n_rows, n_cols, nnz, indptr, indices, w, normalization, flags, out);
    CHECK(normalization >= 0 && normalization <= 4 && n_rows == n_cols, "pgh_graph_from_adjacency_ex: self-loops / the laplacian need a square adjacency");
    // preprocessing.py:107-108: W + self_loops * I (a diagonal entry at the end of every row), then the reductions of :109-138
    const int extra = (self_loops != 0.0 ? 1 : 0) + (laplacian ? 1 : 0);
    std::vector<int64_t> ip(n_rows + 1, 0);
    std::vector<int32_t> idx((size_t)(nnz + extra * n_rows));
    std::vector<double> weights(idx.size());
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t at = indptr[r] + extra * r;
        ip[r] = at;
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k, ++at) {
            idx[at] = indices[k];
            weights[at] = w ? w[k] : 1.0;
        }
        if (self_loops != 0.0) {
            idx[at] = (int32_t)r;
            weights[at++] = self_loops;
        }
        if (laplacian) {
            idx[at] = (int32_t)r;
            weights[at++] = 0.0;                             // takes the +1 of the identity below
        }
    }
    ip[n_rows] = nnz + extra * n_rows;
    if (!laplacian) return pgh_graph_from_adjacency(n_rows, n_cols, ip[n_rows], ip.data(), idx.data(), weights.data(), normalization, flags, out);
    std::vector<double> left(n_rows, 0.0), right(n_cols, 0.0);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t k = ip[r]; k < ip[r + 1]; ++k) {
            left[r] += weights[k];
            right[idx[k]] += weights[k];
        }
    for (auto* v : {&left, &right})
        for (double& x : *v) {
            x = std::sqrt(x);
            if (x != 0.0) x = 1.0 / x;
        }
    std::vector<double> data(weights.size());
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int64_t k = ip[r]; k < ip[r + 1]; ++k) data[k] = -((left[r] * weights[k]) * right[idx[k]]);      // preprocessing.py:121-122
        data[ip[r + 1] - 1] = 1.0;
    }
    return pgh_graph_from_csr(n_rows, n_cols, ip[n_rows], ip.data(), idx.data(), data.data(), flags, out);
}
int pgh_graph_destroy(pgh_graph_t g) {
    delete g;
    return 0;
}
int pgh_graph_info(pgh_graph_t g, int64_t* a, int64_t* b, int64_t* c, int64_t* d) {
    if (a) *a = g->n_rows;
    if (b) *b = g->n_cols;
    if (c) *c = g->nnz;
    if (d) *d = 0;
    return 0;
}
int pgh_graph_format(pgh_graph_t, char* buf, int len) {
    snprintf(buf, len, "host-oracle csr");
    return 0;
}
int pgh_graph_degrees(pgh_graph_t g, pgh_vec_t out) {
    CHECK(g && out && out->n == g->n_rows, "pgh_graph_degrees: length mismatch");
    std::copy(g->degrees.begin(), g->degrees.end(), out->data);
    return 0;
}
int pgh_graph_download(pgh_graph_t g, int64_t* ip, int32_t* idx, float* d) {
    std::copy(g->rowptr.begin(), g->rowptr.end(), ip);
    if (idx) std::copy(g->col.begin(), g->col.end(), idx);
    if (d) std::copy(g->val.begin(), g->val.end(), d);
    return 0;
}

// row sum of M^T x in f64 from f32 products (mirrors the kernel: f32 multiply, f64 accumulate, f32 store)
static inline float row_dot(const pgh_graph_s* g, const float* x, int64_t row) {
    double acc = 0;
    for (int64_t k = g->rowptr[row]; k < g->rowptr[row + 1]; ++k) acc += (double)(g->val[k] * x[g->col[k]]);
    return (float)acc;
}
static int check_gv(pgh_graph_t g, pgh_vec_t x, pgh_vec_t y, const char* who) {
    CHECK(g && x && y, std::string(who) + ": null argument");
    CHECK(x->n == g->n_rows, std::string(who) + ": input length must equal the number of rows of M");
    CHECK(y->n == g->n_cols, std::string(who) + ": output length must equal the number of columns of M");
    CHECK(x->data != y->data, std::string(who) + ": conv must be pure (output aliases input)");
    return 0;
}
int pgh_spmv(pgh_graph_t g, pgh_vec_t x, pgh_vec_t y) {
    if (check_gv(g, x, y, "pgh_spmv")) return 1;
    for (int64_t r = 0; r < g->n_cols; ++r) y->data[r] = row_dot(g, x->data, r);
    return 0;
}

int pgh_spmv_dropout(pgh_graph_t g, pgh_vec_t x, pgh_vec_t y, double rate, uint64_t seed) {
    if (check_gv(g, x, y, "pgh_spmv_dropout")) return 1;
    CHECK(rate >= 0.0 && rate < 1.0, "pgh_spmv_dropout: rate must lie in [0, 1)");
    const uint32_t threshold = (uint32_t)std::floor(rate * 4294967296.0);
    const float keep = (float)(1.0 / (1.0 - rate));
    for (int64_t r = 0; r < g->n_cols; ++r) {
        double acc = 0;
        for (int64_t k = g->rowptr[r]; k < g->rowptr[r + 1]; ++k) {
            uint64_t z = (seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z = z ^ (z >> 31);
            const float f = (uint32_t)(z >> 32) >= threshold ? keep : 0.f;
            acc += (double)((g->val[k] * f) * x->data[g->col[k]]);
        }
        y->data[r] = (float)acc;
    }
    return 0;
}

int pgh_graph_degrees_dropout(pgh_graph_t g, double rate, uint64_t seed, pgh_vec_t out) {
    CHECK(g && out && out->n == g->n_rows && rate >= 0.0 && rate < 1.0, "pgh_graph_degrees_dropout: bad argument");
    const uint32_t threshold = (uint32_t)std::floor(rate * 4294967296.0);
    const float keep = (float)(1.0 / (1.0 - rate));
    std::vector<double> acc((size_t)g->n_rows, 0.0);
    for (int64_t k = 0; k < g->nnz; ++k) {
        uint64_t z = (seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        const float f = (uint32_t)(z >> 32) >= threshold ? keep : 0.f;
        if (f != 0.f) acc[(size_t)g->col[k]] += (double)(g->val[k] * f);
    }
    for (int64_t i = 0; i < g->n_rows; ++i) out->data[i] = (float)acc[(size_t)i];
    return 0;
}

static double ppr_step(const pgh_graph_s* g, const float* x, double xs, const float* p, double alpha, float* y) {
    const float a = (float)(alpha * xs), b = (float)(1.0 - alpha);
    double sum = 0;
    for (int64_t r = 0; r < g->n_cols; ++r) {
        const float v = a * row_dot(g, x, r) + b * p[r];        // adhoc.py:36
        y[r] = v;
        sum += v;
    }
    return sum;
}
static double absorb_step(const pgh_graph_s* g, const float* x, double xs, const float* p, const float* deg,
                          const float* lam, float* y) {
    const float a = (float)xs;
    double sum = 0;
    for (int64_t r = 0; r < g->n_cols; ++r) {
        const float v = (a * row_dot(g, x, r) * deg[r] + p[r] * lam[r]) / (lam[r] + deg[r]);   // adhoc.py:167-168
        y[r] = v;
        sum += v;
    }
    return sum;
}
int pgh_ppr_step(pgh_graph_t g, pgh_vec_t x, double xs, pgh_vec_t p, double alpha, pgh_vec_t y, double* sum_y) {
    if (check_gv(g, x, y, "pgh_ppr_step")) return 1;
    CHECK(p && p->n == g->n_cols, "pgh_ppr_step: personalization length mismatch");
    const double s = ppr_step(g, x->data, xs, p->data, alpha, y->data);
    if (sum_y) *sum_y = s;
    return 0;
}
// ---- resident iterates (include/pgh.h): the double's "id space" is deliberately NOT the caller's -- ids reversed, three zero padding
// slots behind them, a gather form that holds twice the iterate -- so that host logic that mixes the two spaces, forgets the padding or
// gathers from the wrong form fails on the CPU
static const int64_t kResidentPad = 3;
static bool resident_ok(const pgh_graph_s* g) {
    const char* e = getenv("PGH_RESIDENT");
    return g && g->n_rows == g->n_cols && g->gather_blk == 0 && g->n_cols > 0 && !(e != nullptr && atoi(e) == 0);
}
int pgh_last_build_profile(char* buf, int buflen) {
    CHECK(buf && buflen > 0, "pgh_last_build_profile: null buffer");
    buf[0] = 0;                                             // the double builds nothing worth timing
    return 0;
}
int pgh_graph_resident_len(pgh_graph_t g, int64_t* n_int, int64_t* n_gather) {
    CHECK(g && n_int && n_gather, "pgh_graph_resident_len: null argument");
    *n_int = resident_ok(g) ? g->n_cols + kResidentPad : 0;
    *n_gather = resident_ok(g) ? g->n_cols + kResidentPad + 1 : 0;
    return 0;
}
int pgh_resident_in(pgh_graph_t g, pgh_vec_t x, double hole, pgh_vec_t x_int, pgh_vec_t xg) {
    CHECK(resident_ok(g), "pgh_resident_in: this graph's image has no resident form");
    const int64_t n = g->n_cols;
    CHECK(x && x_int && x->n == n && x_int->n == n + kResidentPad && (xg == nullptr || xg->n == n + kResidentPad + 1), "pgh_resident_in: vector length mismatch");
    for (int64_t i = 0; i < n; ++i) x_int->data[i] = x->data[n - 1 - i];
    for (int64_t i = n; i < n + kResidentPad; ++i) x_int->data[i] = (float)hole;
    if (xg != nullptr)
        for (int64_t i = 0; i < n + kResidentPad; ++i) xg->data[i] = 2.f * x_int->data[i];
    return 0;
}
int pgh_resident_gather(pgh_graph_t g, pgh_vec_t x_int, pgh_vec_t xg) {
    CHECK(resident_ok(g), "pgh_resident_gather: this graph's image has no resident form");
    const int64_t n = g->n_cols;
    CHECK(x_int && xg && x_int->n == n + kResidentPad && xg->n == n + kResidentPad + 1, "pgh_resident_gather: vector length mismatch");
    for (int64_t i = 0; i < n + kResidentPad; ++i) xg->data[i] = 2.f * x_int->data[i];
    return 0;
}
int pgh_resident_out(pgh_graph_t g, pgh_vec_t y_int, double factor, pgh_vec_t y) {
    CHECK(resident_ok(g), "pgh_resident_out: this graph's image has no resident form");
    const int64_t n = g->n_cols;
    CHECK(y && y_int && y->n == n && y_int->n == n + kResidentPad, "pgh_resident_out: vector length mismatch");
    for (int64_t i = 0; i < n; ++i) y->data[n - 1 - i] = y_int->data[i] * (float)factor;
    return 0;
}
int pgh_resident_step(pgh_graph_t g, int32_t mode, pgh_vec_t x_int, pgh_vec_t xg, double a, pgh_vec_t v_int, double b, pgh_vec_t deg_int,
                      pgh_vec_t lam_int, pgh_vec_t y_int, pgh_vec_t yg, double* sum_y) {
    CHECK(resident_ok(g), "pgh_resident_step: this graph's image has no resident form");
    const int64_t n = g->n_cols;
    CHECK(mode >= 0 && mode <= 2, "pgh_resident_step: mode 0, 1 or 2");
    CHECK(mode != 2 || (deg_int && lam_int && deg_int->n == n + kResidentPad && lam_int->n == n + kResidentPad), "pgh_resident_step: mode 2 needs the resident degrees and absorption");
    CHECK(x_int && xg && y_int && yg && x_int->n == n + kResidentPad && y_int->n == n + kResidentPad && xg->n == n + kResidentPad + 1 &&
              yg->n == n + kResidentPad + 1 && x_int->data != y_int->data && xg->data != yg->data,
          "pgh_resident_step: iterate length mismatch / aliasing");
    CHECK(mode == 0 || (v_int && v_int->n == n + kResidentPad), "pgh_resident_step: modes 1 and 2 need the resident second operand");
    std::vector<float> x((size_t)n);
    for (int64_t i = 0; i < n; ++i) x[(size_t)(n - 1 - i)] = 0.5f * xg->data[i];        // the step gathers from the gather form
    double sum = 0;
    const float fa = (float)a, fb = (float)b;
    for (int64_t r = 0; r < n; ++r) {
        float v = fa * row_dot(g, x.data(), r);
        if (mode == 1) v += fb * v_int->data[n - 1 - r];
        if (mode == 2) {
            const float d = deg_int->data[n - 1 - r], l = lam_int->data[n - 1 - r];
            v = (v * d + v_int->data[n - 1 - r] * l) / (l + d);                                 // adhoc.py:167-168
        }
        y_int->data[n - 1 - r] = v;
        sum += v;
    }
    for (int64_t i = n; i < n + kResidentPad; ++i) y_int->data[i] = 0.f;
    for (int64_t i = 0; i < n + kResidentPad; ++i) yg->data[i] = 2.f * y_int->data[i];
    if (sum_y) *sum_y = sum;
    return 0;
}
int pgh_absorb_step(pgh_graph_t g, pgh_vec_t x, double xs, pgh_vec_t p, pgh_vec_t deg, pgh_vec_t lam, pgh_vec_t y,
                    double* sum_y) {
    if (check_gv(g, x, y, "pgh_absorb_step")) return 1;
    CHECK(p && deg && lam && p->n == g->n_cols && deg->n == g->n_cols && lam->n == g->n_cols,
          "pgh_absorb_step: vector length mismatch");
    const double s = absorb_step(g, x->data, xs, p->data, deg->data, lam->data, y->data);
    if (sum_y) *sum_y = s;
    return 0;
}
static double poly_step(const pgh_graph_s* g, const float* term, float* term_out, double a, double b, float* result,
                        double c, int linf) {
    double delta = 0;
    for (int64_t r = 0; r < g->n_cols; ++r) {
        float t = (float)a * row_dot(g, term, r);
        if (b != 0.0) t += (float)b * term[r];
        term_out[r] = t;
        const float r_old = result[r], r_new = r_old + (float)c * t;
        result[r] = r_new;
        const double d = std::fabs((double)r_new - (double)r_old);
        delta = linf ? std::fmax(delta, d) : delta + d;
    }
    return delta;
}
int pgh_poly_step(pgh_graph_t g, pgh_vec_t term, pgh_vec_t term_out, double a, double b, pgh_vec_t result, double c,
                  int err_kind, double* delta) {
    if (check_gv(g, term, term_out, "pgh_poly_step")) return 1;
    CHECK(result && result->n == g->n_cols, "pgh_poly_step: result length mismatch");
    CHECK(b == 0.0 || term->n == g->n_cols, "pgh_poly_step: b != 0 needs a square matrix");
    double d = poly_step(g, term->data, term_out->data, a, b, result->data, c, err_kind == PGH_ERR_LINF);
    if (err_kind == PGH_ERR_MABS && g->n_cols > 0) d /= (double)g->n_cols;
    if (delta) *delta = d;
    return 0;
}

// ------------------------------------------------------------------------------------------ loops
// Direct restatement of GraphFilter.rank's loop (abstract_filters.py:58-62) with ConvergenceManager
// (convergence.py:77-101): has_converged runs BEFORE every step and increments `iteration` first.
extern "C++" {
struct Conv {
    const pgh_loop_cfg* cfg;
    int iteration = 0;
    bool have_last = false;
    bool converged = false;
    double last_err = 0;
    // returns true when the loop must stop; err_fn evaluates the residual between the last two iterates
    template <class F>
    bool has_converged(F err_fn) {
        ++iteration;
        if (iteration >= cfg->max_iters) return true;                      // caller raises unless ITERS
        bool done = false;
        if (have_last && cfg->err_kind != PGH_ERR_ITERS && iteration % cfg->end_modulo == 0) {
            last_err = err_fn();
            done = last_err <= cfg->tol;
            converged = done;
        }
        have_last = true;
        return done;
    }
};

template <class StepFn>
static int recursive_run(pgh_graph_t g, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res, const float* start,
                         StepFn step) {
    CHECK(g->n_rows == g->n_cols, "recursive filters need a square matrix");
    CHECK(ranks && ranks->n == g->n_cols, "ranks length mismatch");
    CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    const int64_t n = g->n_cols;
    std::vector<float> cur(ranks->data, ranks->data + n), prev(n), next(n);
    if (cfg->start_from_p) cur.assign(start, start + n);                    // abstract_filters.py:56 (no warm start)
    double cur_scale = 1.0, prev_scale = 1.0;
    Conv cm{cfg};
    int steps = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (!cm.has_converged([&] { return scaled_res(cfg->err_kind, cur.data(), cur_scale, prev.data(), prev_scale, n); })) {
        const auto ts = std::chrono::steady_clock::now();
        const double s = step(cur.data(), cur_scale, next.data());
        if (g_prof_on) {
            g_prof_count[PGH_K_SPMV] += 1;
            g_prof_ms[PGH_K_SPMV] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
        }
        prev.swap(cur);
        prev_scale = cur_scale;
        cur.swap(next);
        cur_scale = cfg->use_quotient ? (s != 0.0 ? 1.0 / s : 0.0) : 1.0;   // abstract_filters.py:133-134
        ++steps;
    }
    const float f = (float)(cur_scale * cfg->out_scale);                     // abstract_filters.py:63-64
    for (int64_t i = 0; i < n; ++i) ranks->data[i] = cur[i] * f;
    memset(res, 0, sizeof(*res));
    res->iterations = cm.iteration;
    res->converged = cm.converged ? 1 : 0;
    res->spmv_count = steps;
    res->last_error = cm.last_err;
    res->loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
}  // extern "C++"

// personalization / in_norm in f32 (abstract_filters.py:55)
static std::vector<float> normalised(pgh_vec_t p, const pgh_loop_cfg* cfg) {
    std::vector<float> out(p->data, p->data + p->n);
    if (cfg->in_norm != 0.0 && cfg->in_norm != 1.0) {
        const float d = (float)cfg->in_norm;
        for (float& v : out) v = v / d;
    }
    return out;
}
// pgh_loop_cfg::in_norm < 0: the engine computes sum |p| itself (abstract_filters.py:52), out_scale < 0 = "times that norm".
// Returns the configuration with both resolved; *norm = 0 means "hand the personalization back" (the caller returns at once).
static pgh_loop_cfg resolve_norm(pgh_vec_t p, const pgh_loop_cfg* cfg, double* norm, bool* wanted) {
    pgh_loop_cfg c = *cfg;
    *wanted = cfg->in_norm < 0.0;
    *norm = cfg->in_norm;
    if (*wanted) {
        double s = 0.0;
        for (int64_t i = 0; i < p->n; ++i) s += std::fabs((double)p->data[i]);
        *norm = s;
        c.in_norm = s;
    }
    if (cfg->out_scale < 0.0) c.out_scale = *norm;
    return c;
}
#define PGH_RESOLVE_NORM(P, CFG, RES)                              \
    double norm_ = 0.0;                                            \
    bool norm_wanted_ = false;                                     \
    const pgh_loop_cfg cfg_ = resolve_norm(P, CFG, &norm_, &norm_wanted_); \
    if (norm_wanted_ && norm_ == 0.0) {                            \
        memset(RES, 0, sizeof(*RES));                              \
        (RES)->iterations = 1;                                     \
        return 0;                                                  \
    }                                                              \
    const pgh_loop_cfg* cfg = &cfg_;
#define PGH_REPORT_NORM(RC, RES)                                   \
    if ((RC) == 0 && norm_wanted_) (RES)->in_norm = norm_;

int pgh_ppr_run(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg_in, pgh_loop_result* res) {
    CHECK(g && p && ranks && cfg_in && res, "pgh_ppr_run: null argument");
    CHECK(p->n == g->n_cols, "pgh_ppr_run: personalization length mismatch");
    PGH_RESOLVE_NORM(p, cfg_in, res)
    const std::vector<float> pn = normalised(p, cfg);
    const int rc = recursive_run(g, ranks, cfg, res, pn.data(), [&](const float* x, double xs, float* y) {
        return ppr_step(g, x, xs, pn.data(), cfg->alpha, y);
    });
    PGH_REPORT_NORM(rc, res)
    return rc;
}
// PageRank with f64 storage (include/pgh.h pgh_ppr_run_f64): iterates, sums, quotient and residual in double over the stored f32 matrix
static int recursive_run_f64(pgh_graph_t g, int mode, pgh_vec_t p, pgh_vec_t lam, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res);
int pgh_ppr_run_f64(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res) {
    return recursive_run_f64(g, 0, p, nullptr, ranks, cfg, res);
}
int pgh_absorb_run_f64(pgh_graph_t g, pgh_vec_t p, pgh_vec_t lam, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res) {
    CHECK(lam && g && lam->n == g->n_cols, "pgh_absorb_run_f64: absorption length mismatch");
    return recursive_run_f64(g, 1, p, lam, ranks, cfg, res);
}
int pgh_sarw_run_f64(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res) {
    return recursive_run_f64(g, 2, p, nullptr, ranks, cfg, res);
}
// mode 0 PageRank (adhoc.py:34-36), 1 AbsorbingWalks (adhoc.py:157-169), 2 SymmetricAbsorbingRandomWalks (adhoc.py:348-364); the walks take
// the graph's f32 degrees, like the engine
static int recursive_run_f64(pgh_graph_t g, int mode, pgh_vec_t p, pgh_vec_t lam, pgh_vec_t ranks, const pgh_loop_cfg* cfg, pgh_loop_result* res) {
    CHECK(g && p && ranks && cfg && res, "pgh_ppr_run_f64: null argument");
    CHECK(g->n_rows == g->n_cols && p->n == g->n_cols && ranks->n == g->n_cols, "pgh_ppr_run_f64: shape mismatch");
    CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    memset(res, 0, sizeof(*res));
    const int64_t n = g->n_cols;
    double norm = cfg->in_norm;
    if (norm < 0.0) {
        norm = 0.0;
        for (int64_t i = 0; i < n; ++i) norm += std::fabs((double)p->data[i]);
        res->in_norm = norm;
        if (norm == 0.0) return 0;
    }
    if (norm == 0.0) norm = 1.0;
    std::vector<double> pn(n), cur(n), next(n);
    for (int64_t i = 0; i < n; ++i) pn[i] = (double)p->data[i] / norm;
    for (int64_t i = 0; i < n; ++i) cur[i] = cfg->start_from_p ? pn[i] : (double)ranks->data[i];
    double scale = 1.0, err = 0.0;
    int it = 1, spmv = 0;
    bool converged = false;
    const auto t0 = std::chrono::steady_clock::now();
    while (it < cfg->max_iters) {
        double S = 0.0;
        for (int64_t r = 0; r < n; ++r) {
            double acc = 0.0;
            if (mode == 2) {
                for (int64_t k = g->rowptr[r]; k < g->rowptr[r + 1]; ++k) {
                    const double dc = (double)g->degrees[g->col[k]];
                    acc += (double)g->val[k] * (cur[g->col[k]] / ((std::sqrt(dc * 4.0 + 1.0) + 1.0) / 2.0));      // conv(ranks * pre, M)
                }
            } else {
                for (int64_t k = g->rowptr[r]; k < g->rowptr[r + 1]; ++k) acc += (double)g->val[k] * cur[g->col[k]];
            }
            const double d = (double)g->degrees[r];
            if (mode == 0) {
                next[r] = cfg->alpha * scale * acc + (1.0 - cfg->alpha) * pn[r];
            } else if (mode == 1) {
                const double l = (double)lam->data[r];
                next[r] = (scale * acc * d + pn[r] * l) / (l + d);                          // adhoc.py:167-168
            } else {
                const double a = (std::sqrt(d * 4.0 + 1.0) + 1.0) / 2.0;
                next[r] = scale * acc * (d / (a + d)) + pn[r] * (a / (a + d));              // adhoc.py:363-364
            }
            S += next[r];
        }
        const double scale_new = cfg->use_quotient ? (S != 0.0 ? 1.0 / S : 0.0) : 1.0;
        ++spmv;
        ++it;
        const bool check = cfg->err_kind != PGH_ERR_ITERS && it < cfg->max_iters && it % cfg->end_modulo == 0;
        if (check) {
            err = 0.0;
            for (int64_t i = 0; i < n; ++i) {
                const double d = std::fabs(next[i] * scale_new - cur[i] * scale);
                err = cfg->err_kind == PGH_ERR_LINF ? std::max(err, d) : err + d;
            }
            if (cfg->err_kind == PGH_ERR_MABS) err /= (double)n;
        }
        cur.swap(next);
        scale = scale_new;
        if (check && err <= cfg->tol) {
            converged = true;
            break;
        }
    }
    const double f = scale * (cfg->out_scale < 0.0 ? norm : cfg->out_scale);
    for (int64_t i = 0; i < n; ++i) ranks->data[i] = (float)(cur[i] * f);
    res->iterations = it;
    res->converged = converged ? 1 : 0;
    res->spmv_count = spmv;
    res->last_error = err;
    res->loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
// PageRank on graph_dropout(M, rate) with the mask of step k = pgh_spmv_dropout's for seed seed0 + k - 1
int pgh_ppr_run_dropout(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg_in, double rate, uint64_t seed0,
                        pgh_loop_result* res) {
    CHECK(g && p && ranks && cfg_in && res, "pgh_ppr_run_dropout: null argument");
    CHECK(p->n == g->n_cols, "pgh_ppr_run_dropout: personalization length mismatch");
    CHECK(rate >= 0.0 && rate < 1.0, "pgh_ppr_run_dropout: rate must lie in [0, 1)");
    PGH_RESOLVE_NORM(p, cfg_in, res)
    const std::vector<float> pn = normalised(p, cfg);
    const uint32_t threshold = (uint32_t)std::floor(rate * 4294967296.0);
    const float keep = (float)(1.0 / (1.0 - rate));
    uint64_t step_no = 0;
    const int rc = recursive_run(g, ranks, cfg, res, pn.data(), [&](const float* x, double xs, float* y) {
        const uint64_t seed = seed0 + step_no++;
        const float a = (float)(cfg->alpha * xs), b = (float)(1.0 - cfg->alpha);
        double sum = 0;
        for (int64_t r = 0; r < g->n_cols; ++r) {
            double acc = 0;
            for (int64_t k = g->rowptr[r]; k < g->rowptr[r + 1]; ++k) {
                uint64_t z = (seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                z = z ^ (z >> 31);
                const float f = (uint32_t)(z >> 32) >= threshold ? keep : 0.f;
                acc += (double)((g->val[k] * f) * x[g->col[k]]);
            }
            const float v = a * (float)acc + b * pn[r];
            y[r] = v;
            sum += v;
        }
        return sum;
    });
    PGH_REPORT_NORM(rc, res)
    return rc;
}

int pgh_absorb_run(pgh_graph_t g, pgh_vec_t p, pgh_vec_t lam, pgh_vec_t ranks, const pgh_loop_cfg* cfg_in,
                   pgh_loop_result* res) {
    CHECK(g && p && lam && ranks && cfg_in && res, "pgh_absorb_run: null argument");
    CHECK(p->n == g->n_cols && lam->n == g->n_cols, "pgh_absorb_run: vector length mismatch");
    PGH_RESOLVE_NORM(p, cfg_in, res)
    const std::vector<float> pn = normalised(p, cfg);
    const int rc = recursive_run(g, ranks, cfg, res, pn.data(), [&](const float* x, double xs, float* y) {
        return absorb_step(g, x, xs, pn.data(), g->degrees.data(), lam->data, y);
    });
    PGH_REPORT_NORM(rc, res)
    return rc;
}

// SymmetricAbsorbingRandomWalks (adhoc.py:348-364) written the way the reference writes it: precomputed skews, pre-scaled iterate
int pgh_sarw_run(pgh_graph_t g, pgh_vec_t p, pgh_vec_t ranks, const pgh_loop_cfg* cfg_in, pgh_loop_result* res) {
    CHECK(g && p && ranks && cfg_in && res, "pgh_sarw_run: null argument");
    CHECK(p->n == g->n_cols && g->n_rows == g->n_cols, "pgh_sarw_run: shape mismatch");
    PGH_RESOLVE_NORM(p, cfg_in, res)
    const int64_t n = g->n_cols;
    const std::vector<float> pn = normalised(p, cfg);
    std::vector<float> left(n), post(n), skew(n), xs_buf(n);
    for (int64_t i = 0; i < n; ++i) {
        const double deg = (double)g->degrees[i];
        const double a = (1.0 + std::sqrt(1.0 + 4.0 * deg)) / 2.0;
        left[i] = (float)(1.0 / a);
        post[i] = (float)(deg / (a + deg));
        skew[i] = (float)(a / (a + deg));
    }
    const int rc = recursive_run(g, ranks, cfg, res, pn.data(), [&](const float* x, double xs, float* y) {
        for (int64_t i = 0; i < n; ++i) xs_buf[i] = x[i] * left[i];
        const float s = (float)xs;
        double sum = 0;
        for (int64_t r = 0; r < n; ++r) {
            const float v = s * row_dot(g, xs_buf.data(), r) * post[r] + pn[r] * skew[r];
            y[r] = v;
            sum += v;
        }
        return sum;
    });
    PGH_REPORT_NORM(rc, res)
    return rc;
}

int pgh_poly_run(pgh_graph_t g, pgh_vec_t p, const double* coeffs, int32_t num_coeffs, int32_t chebyshev,
                 pgh_vec_t result, const pgh_loop_cfg* cfg, pgh_loop_result* res) {
    CHECK(g && p && result && cfg && res, "pgh_poly_run: null argument");
    const int64_t n = g->n_cols;
    CHECK(g->n_rows == n && p->n == n && result->n == n, "pgh_poly_run: shape mismatch");
    CHECK(cfg->end_modulo >= 1, "end_modulo must be >= 1");
    auto coeff = [&](int it) { return (it >= 1 && it <= num_coeffs) ? coeffs[it - 1] : 0.0; };
    if (chebyshev) {
        // The engine evaluates the reference's "chebyshev" recurrence in f64 (it amplifies rounding noise: an f32 evaluation
        // cannot hold 1e-6); so does the double: f64 vectors, f64 row sums over the f32 matrix values.
        std::vector<double> resv(n, 0.0), prev_res(n, 0.0), term(n), prev_term, tmp(n);
        for (int64_t i = 0; i < n; ++i) term[i] = (double)p->data[i];
        Conv cm{cfg};
        int spmv = 0;
        const auto t0 = std::chrono::steady_clock::now();
        auto resid = [&] {
            double acc = 0;
            for (int64_t i = 0; i < n; ++i) {
                const double d = std::fabs(resv[i] - prev_res[i]);
                acc = cfg->err_kind == PGH_ERR_LINF ? std::max(acc, d) : acc + d;
            }
            return cfg->err_kind == PGH_ERR_MABS && n > 0 ? acc / (double)n : acc;
        };
        while (!cm.has_converged(resid)) {
            const int it = cm.iteration;
            const double c = coeff(it);
            prev_res = resv;
            if (chebyshev == 1 && it == 2) prev_term = term;                 // abstract_filters.py:216-224 (chebyshev == 2: taylor in f64)
            if (chebyshev == 1 && it > 2) {
                for (int64_t i = 0; i < n; ++i) term[i] = 2.0 * term[i] - prev_term[i];
                prev_term = term;
            }
            for (int64_t i = 0; i < n; ++i) resv[i] = resv[i] + c * term[i];
            for (int64_t r = 0; r < n; ++r) {
                double acc = 0;
                for (int64_t k = g->rowptr[r]; k < g->rowptr[r + 1]; ++k) acc += (double)g->val[k] * term[g->col[k]];
                tmp[r] = acc;
            }
            term.swap(tmp);
            ++spmv;
        }
        for (int64_t i = 0; i < n; ++i) result->data[i] = (float)(resv[i] * cfg->out_scale);
        memset(res, 0, sizeof(*res));
        res->iterations = cm.iteration;
        res->converged = cm.converged ? 1 : 0;
        res->spmv_count = spmv;
        res->last_error = cm.last_err;
        res->loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    // reference order (abstract_filters.py:248-256): accumulate the current term, THEN advance the power.
    std::vector<float> resv(n, 0.f), prev_res(n, 0.f), term(p->data, p->data + n), prev_term, tmp(n);
    Conv cm{cfg};
    int spmv = 0;
    const int linf = cfg->err_kind == PGH_ERR_LINF;
    const auto t0 = std::chrono::steady_clock::now();
    while (!cm.has_converged([&] { return scaled_res(cfg->err_kind, resv.data(), 1.0, prev_res.data(), 1.0, n); })) {
        const int it = cm.iteration;
        const double c = coeff(it);
        prev_res = resv;
        if (chebyshev) {                                                     // abstract_filters.py:216-224
            if (it == 2) prev_term = term;
            if (it > 2) {
                for (int64_t i = 0; i < n; ++i) term[i] = 2.f * term[i] - prev_term[i];
                prev_term = term;
            }
        }
        for (int64_t i = 0; i < n; ++i) resv[i] = resv[i] + (float)c * term[i];
        for (int64_t r = 0; r < n; ++r) tmp[r] = row_dot(g, term.data(), r);  // abstract_filters.py:256
        term.swap(tmp);
        ++spmv;
        (void)linf;
    }
    const float f = (float)cfg->out_scale;
    for (int64_t i = 0; i < n; ++i) result->data[i] = resv[i] * f;
    memset(res, 0, sizeof(*res));
    res->iterations = cm.iteration;
    res->converged = cm.converged ? 1 : 0;
    res->spmv_count = spmv;
    res->last_error = cm.last_err;
    res->loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}


// ------------------------------------------------------------------------------------------ synthetic
// host restatement of pgh_graph_rmat (same splitmix64 hash and thresholds as oracle/rmat_np.py)
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// multi-seed entry points: column-by-column restatement (NodeRanking.propagate, signals.py:225-226)
// terms of the "chebyshev" recurrence in f64 over the f32 matrix values (abstract_filters.py:216-224), rounded once per column
int pgh_poly_terms(pgh_graph_t g, pgh_vec_t p, int32_t chebyshev, int32_t skip, int32_t count, pgh_mat_t out, int32_t first_col) {
    CHECK(g && p && out && chebyshev != 0 && g->n_rows == g->n_cols && p->n == g->n_cols && out->n == g->n_cols, "pgh_poly_terms: bad arguments");
    CHECK(skip >= 0 && count >= 1 && first_col >= 0 && first_col + count <= out->b, "pgh_poly_terms: column range outside the slab");
    const int64_t n = g->n_cols;
    std::vector<double> term(n), next(n);
    for (int64_t i = 0; i < n; ++i) term[i] = (double)p->data[i];
    for (int k = 1; k <= skip + count; ++k) {
        if (k > 1) {
            for (int64_t r = 0; r < n; ++r) {
                double acc = 0;
                for (int64_t e = g->rowptr[r]; e < g->rowptr[r + 1]; ++e) acc += (double)g->val[e] * term[g->col[e]];
                next[r] = k > 2 ? 2.0 * acc - term[r] : acc;
            }
            term.swap(next);
        }
        if (k > skip)
            for (int64_t i = 0; i < n; ++i) out->data[i * out->b + first_col + (k - 1 - skip)] = (float)term[i];
    }
    return 0;
}
int pgh_spmm(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y) {
    CHECK(g && x && y && x->n == g->n_rows && y->n == g->n_cols && x->b == y->b, "pgh_spmm: shape mismatch");
    CHECK(x->b >= 1 && x->b <= 64, "pgh_spmm: the batch width must be in [1, 64]");
    std::vector<float> col(g->n_rows);
    for (int32_t j = 0; j < x->b; ++j) {
        for (int64_t i = 0; i < g->n_rows; ++i) col[i] = x->data[i * x->b + j];
        for (int64_t r = 0; r < g->n_cols; ++r) y->data[r * y->b + j] = row_dot(g, col.data(), r);
    }
    return 0;
}
int pgh_ppr_run_batch(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales,
                      pgh_loop_result* results) {
    CHECK(g && p && ranks && cfg && results && p->n == g->n_cols && ranks->n == g->n_cols && p->b == ranks->b,
          "pgh_ppr_run_batch: shape mismatch");
    CHECK(p->b >= 1 && p->b <= 64, "pgh_ppr_run_batch: the batch width must be in [1, 64]");
    const int64_t n = g->n_cols;
    for (int32_t j = 0; j < p->b; ++j) {
        pgh_vec_t vp, vr;
        pgh_vec_alloc(n, &vp);
        pgh_vec_alloc(n, &vr);
        for (int64_t i = 0; i < n; ++i) {
            vp->data[i] = p->data[i * p->b + j];
            vr->data[i] = ranks->data[i * p->b + j];
        }
        pgh_loop_cfg c = *cfg;
        if (out_scales) c.out_scale = out_scales[j];
        const int rc = pgh_ppr_run(g, vp, vr, &c, &results[j]);
        for (int64_t i = 0; i < n; ++i) ranks->data[i * p->b + j] = vr->data[i];
        pgh_vec_free(vp);
        pgh_vec_free(vr);
        if (rc) return rc;
    }
    return 0;
}

// the batch entry points with graph_dropout: the mask of pgh_spmv_dropout (a hash of (seed, index of the entry in CSR(M^T) order))
static inline float dropped_row_dot(const pgh_graph_s* g, const float* x, int64_t row, uint64_t seed, uint32_t threshold, float keep) {
    double acc = 0;
    for (int64_t k = g->rowptr[row]; k < g->rowptr[row + 1]; ++k) {
        uint64_t z = (seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        const float f = (uint32_t)(z >> 32) >= threshold ? keep : 0.f;
        acc += (double)((g->val[k] * f) * x[g->col[k]]);
    }
    return (float)acc;
}
int pgh_spmm_dropout(pgh_graph_t g, pgh_mat_t x, pgh_mat_t y, double rate, uint64_t seed) {
    if (rate == 0.0) return pgh_spmm(g, x, y);
    CHECK(g && x && y && x->n == g->n_rows && y->n == g->n_cols && x->b == y->b, "pgh_spmm_dropout: shape mismatch");
    CHECK(x->b >= 1 && x->b <= 64 && rate > 0.0 && rate < 1.0, "pgh_spmm_dropout: batch width in [1, 64], rate in [0, 1)");
    const uint32_t threshold = (uint32_t)std::floor(rate * 4294967296.0);
    const float keep = (float)(1.0 / (1.0 - rate));
    std::vector<float> col(g->n_rows);
    for (int32_t j = 0; j < x->b; ++j) {
        for (int64_t i = 0; i < g->n_rows; ++i) col[i] = x->data[i * x->b + j];
        for (int64_t r = 0; r < g->n_cols; ++r) y->data[r * y->b + j] = dropped_row_dot(g, col.data(), r, seed, threshold, keep);
    }
    return 0;
}
int pgh_ppr_run_batch_dropout(pgh_graph_t g, pgh_mat_t p, pgh_mat_t ranks, const pgh_loop_cfg* cfg, const double* out_scales, double rate,
                              uint64_t seed0, pgh_loop_result* results) {
    if (rate == 0.0) return pgh_ppr_run_batch(g, p, ranks, cfg, out_scales, results);
    CHECK(g && p && ranks && cfg && results && p->n == g->n_cols && ranks->n == g->n_cols && p->b == ranks->b,
          "pgh_ppr_run_batch_dropout: shape mismatch");
    CHECK(p->b >= 1 && p->b <= 64 && rate > 0.0 && rate < 1.0, "pgh_ppr_run_batch_dropout: batch width in [1, 64], rate in [0, 1)");
    const int64_t n = g->n_cols;
    const uint32_t threshold = (uint32_t)std::floor(rate * 4294967296.0);
    const float keep = (float)(1.0 / (1.0 - rate));
    for (int32_t j = 0; j < p->b; ++j) {
        pgh_vec_t vp, vr;
        pgh_vec_alloc(n, &vp);
        pgh_vec_alloc(n, &vr);
        for (int64_t i = 0; i < n; ++i) {
            vp->data[i] = p->data[i * p->b + j];
            vr->data[i] = ranks->data[i * p->b + j];
        }
        pgh_loop_cfg c = *cfg;
        if (out_scales) c.out_scale = out_scales[j];
        const std::vector<float> pn = normalised(vp, &c);
        uint64_t step = 0;                                   // step k of every column uses the mask of seed0 + k - 1
        const int rc = recursive_run(g, vr, &c, &results[j], pn.data(), [&](const float* x, double xs, float* y) {
            const float a = (float)(c.alpha * xs), b = (float)(1.0 - c.alpha);
            const uint64_t seed = seed0 + step++;
            double sum = 0;
            for (int64_t r = 0; r < n; ++r) {
                const float v = a * dropped_row_dot(g, x, r, seed, threshold, keep) + b * pn[r];
                y[r] = v;
                sum += v;
            }
            return sum;
        });
        for (int64_t i = 0; i < n; ++i) ranks->data[i * p->b + j] = vr->data[i];
        pgh_vec_free(vp);
        pgh_vec_free(vr);
        if (rc) return rc;
    }
    return 0;
}

static int rmat_build(int32_t scale, int32_t ef, double a, double b, double c, uint64_t seed, int32_t normalization,
                      int32_t symmetrize, int64_t row_begin, int64_t row_end, int32_t part_rank, int32_t part_count,
                      pgh_graph_t* out) {
    CHECK(scale >= 1 && scale <= 30 && ef >= 1, "pgh_graph_rmat: scale must be in [1, 30]");
    CHECK(normalization >= 0 && normalization <= 2, "pgh_graph_rmat: normalization must be 0, 1 or 2");
    const int64_t n = 1LL << scale, E = n * ef;
    const uint32_t ta = (uint32_t)std::floor(a * 4294967296.0), tb = (uint32_t)std::floor((a + b) * 4294967296.0),
                   tc = (uint32_t)std::floor((a + b + c) * 4294967296.0);
    std::vector<uint32_t> src(E), dst(E);
    for (int64_t e = 0; e < E; ++e) {
        const uint64_t emix = (uint64_t)e * 0xD6E8FEB86659FD93ULL;
        uint32_t s = 0, d = 0;
        for (int pair = 0; pair < (scale + 1) / 2; ++pair) {
            const uint64_t h = splitmix64(splitmix64(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(pair + 1)) ^ emix);
            for (int half = 0; half < 2; ++half) {
                const int level = 2 * pair + half;
                if (level >= scale) break;
                const uint32_t u = half == 0 ? (uint32_t)(h >> 32) : (uint32_t)(h & 0xffffffffu);
                s |= (uint32_t)(u >= tb) << (scale - 1 - level);
                d |= (uint32_t)(((u >= ta) && (u < tb)) || (u >= tc)) << (scale - 1 - level);
            }
        }
        src[e] = s;
        dst[e] = d;
    }
    pgh_graph_s* g = new pgh_graph_s();
    if (part_count > 0) {
        CHECK((part_count & (part_count - 1)) == 0 && part_count <= n && part_rank >= 0 && part_rank < part_count,
              "pgh_graph_rmat_part: the number of partitions must be a power of two and the rank inside it");
        // global relabelling by descending source count (stable), dealt round-robin to B hot-first blocks
        std::vector<uint32_t> cnt(n, 0);
        for (int64_t e = 0; e < E; ++e) {
            cnt[src[e]]++;
            if (symmetrize) cnt[dst[e]]++;
        }
        int B = 1;
        while (B < 8 && n * 4 > (int64_t)B * (8 << 20)) B <<= 1;
        if (B < part_count) B = part_count;
        CHECK(B <= 8, "pgh_graph_rmat_part: at most 8 partitions per node");
        const int64_t blk = n / B;
        std::vector<int32_t> ids(n);
        std::iota(ids.begin(), ids.end(), 0);
        std::stable_sort(ids.begin(), ids.end(), [&](int32_t x, int32_t y) { return cnt[x] > cnt[y]; });
        std::vector<int32_t> iperm(n);
        g->part_perm.assign(n, -1);
        for (int64_t r = 0; r < n; ++r) {
            const int32_t nw = (int32_t)((r % B) * blk + r / B);
            iperm[ids[r]] = nw;
            g->part_perm[nw] = ids[r];
        }
        for (int64_t e = 0; e < E; ++e) {
            src[e] = (uint32_t)iperm[src[e]];
            dst[e] = (uint32_t)iperm[dst[e]];
        }
        row_begin = (int64_t)part_rank * (n / part_count);
        row_end = row_begin + n / part_count;
        g->gather_blocks = B;
        g->gather_blk = blk;
        for (int b = 0; b < 8; ++b) g->gather_base[b] = (int64_t)b * blk;
    }
    if (row_end <= 0) row_end = n;
    CHECK(row_begin >= 0 && row_begin <= row_end && row_end <= n, "pgh_graph_rmat: bad row range");
    std::vector<double> outdeg(n, 0.0), indeg(row_end - row_begin, 0.0);
    std::vector<uint64_t> keys;
    for (int64_t e = 0; e < E; ++e) {
        const uint32_t s = src[e], d = dst[e];
        outdeg[s] += 1;
        if (d >= row_begin && d < row_end) keys.push_back(((uint64_t)(d - row_begin) << 32) | s);
        if (symmetrize) {
            outdeg[d] += 1;
            if (s >= row_begin && s < row_end) keys.push_back(((uint64_t)(s - row_begin) << 32) | d);
        }
    }
    std::sort(keys.begin(), keys.end());
    g->n_rows = n;
    g->n_cols = row_end - row_begin;
    g->row_begin = row_begin;
    std::vector<double> w;
    std::vector<uint64_t> uk;
    for (size_t k = 0; k < keys.size(); ++k) {
        if (k > 0 && keys[k] == keys[k - 1]) w.back() += 1; else { uk.push_back(keys[k]); w.push_back(1); }
    }
    for (size_t k = 0; k < uk.size(); ++k) indeg[uk[k] >> 32] += w[k];
    g->nnz = (int64_t)uk.size();
    g->rowptr.assign(g->n_cols + 1, 0);
    g->col.resize(uk.size());
    g->val.resize(uk.size());
    std::vector<double> deg(n, 0.0);
    for (size_t k = 0; k < uk.size(); ++k) {
        const int64_t row = (int64_t)(uk[k] >> 32);
        const uint32_t s = (uint32_t)(uk[k] & 0xffffffffu);
        double v;
        if (normalization == 0) v = (outdeg[s] != 0 ? 1.0 / outdeg[s] : 0.0) * w[k];
        else if (normalization == 1) {
            const double l = std::sqrt(outdeg[s]), r = std::sqrt(indeg[row]);
            v = ((l != 0 ? 1.0 / l : 0.0) * w[k]) * (r != 0 ? 1.0 / r : 0.0);
        } else v = w[k];
        g->col[k] = (int32_t)s;
        g->val[k] = (float)v;
        deg[s] += v;
        g->rowptr[row + 1]++;
    }
    for (int64_t r = 0; r < g->n_cols; ++r) g->rowptr[r + 1] += g->rowptr[r];
    g->degrees.resize(n);
    for (int64_t i = 0; i < n; ++i) g->degrees[i] = (float)deg[i];
    *out = g;
    return 0;
}

int pgh_graph_rmat(int32_t scale, int32_t ef, double a, double b, double c, uint64_t seed, int32_t normalization,
                   int32_t symmetrize, int64_t row_begin, int64_t row_end, pgh_graph_t* out) {
    return rmat_build(scale, ef, a, b, c, seed, normalization, symmetrize, row_begin, row_end, 0, 0, out);
}
int pgh_graph_rmat_part(int32_t scale, int32_t ef, double a, double b, double c, uint64_t seed, int32_t normalization,
                        int32_t symmetrize, int32_t part_rank, int32_t part_count, pgh_graph_t* out) {
    CHECK(part_count >= 1, "pgh_graph_rmat_part: part_count must be >= 1");
    return rmat_build(scale, ef, a, b, c, seed, normalization, symmetrize, 0, 0, part_rank, part_count, out);
}
int pgh_graph_from_csr_part(int64_t n_rows, int64_t n_cols_local, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                            const double* data, int64_t row_begin, int32_t num_blocks, const int32_t* perm, pgh_graph_t* out) {
    CHECK(perm && num_blocks >= 1 && num_blocks <= 8 && n_rows % num_blocks == 0, "pgh_graph_from_csr_part: bad layout");
    CHECK(row_begin >= 0 && row_begin + n_cols_local <= n_rows, "pgh_graph_from_csr_part: slice outside the id space");
    if (pgh_graph_from_csr(n_rows, n_cols_local, nnz, indptr, indices, data, 0, out)) return 1;
    pgh_graph_s* g = *out;
    g->row_begin = row_begin;
    g->part_perm.assign(perm, perm + n_rows);
    g->gather_blocks = num_blocks;
    g->gather_blk = n_rows / num_blocks;
    for (int b = 0; b < 8; ++b) g->gather_base[b] = (int64_t)b * g->gather_blk;
    return 0;
}
int pgh_graph_perm(pgh_graph_t g, int32_t* new_to_old, int64_t* row_begin) {
    if (row_begin) *row_begin = g->row_begin;
    for (int64_t i = 0; i < g->n_rows; ++i) new_to_old[i] = g->part_perm.empty() ? (int32_t)i : g->part_perm[i];
    return 0;
}
// gather vector in the caller's (possibly trimmed) layout -> dense vector over the whole id space
static const float* expand_gather(pgh_graph_s* g, const float* xg, int64_t xg_len) {
    if (g->gather_blk <= 0) return xg;
    bool identity = true;
    for (int b = 0; b < g->gather_blocks; ++b) identity = identity && g->gather_base[b] == (int64_t)b * g->gather_blk;
    if (identity) return xg;
    g->dense_x.assign((size_t)g->n_rows, 0.f);
    for (int b = 0; b < g->gather_blocks; ++b)
        for (int64_t i = 0; i < g->gather_blk && g->gather_base[b] + i < xg_len; ++i)
            g->dense_x[(size_t)(b * g->gather_blk + i)] = xg[g->gather_base[b] + i];
    return g->dense_x.data();
}
int pgh_graph_gather_layout(pgh_graph_t g, int32_t* num_blocks, int64_t* blk_size, int32_t* live) {
    CHECK(g && g->gather_blk > 0, "pgh_graph_gather_layout: not a partitioned graph");
    if (num_blocks) *num_blocks = g->gather_blocks;
    if (blk_size) *blk_size = g->gather_blk;
    if (live) {
        for (int b = 0; b < 8; ++b) live[b] = 0;
        for (int64_t k = 0; k < g->nnz; ++k) {
            const int64_t c = g->col[k];
            const int b = (int)(c / g->gather_blk);
            const int32_t top = (int32_t)(c % g->gather_blk) + 1;
            if (top > live[b]) live[b] = top;
        }
        for (int b = 0; b < g->gather_blocks; ++b)
            if (live[b] < 1) live[b] = 1;
    }
    return 0;
}
int pgh_graph_set_gather_bases(pgh_graph_t g, const int64_t* bases) {
    CHECK(g && bases && g->gather_blk > 0, "pgh_graph_set_gather_bases: not a partitioned graph");
    for (int b = 0; b < g->gather_blocks; ++b) g->gather_base[b] = bases[b];
    return 0;
}
// device-driven partitioned loop: same state layout as the engine (include/pgh.h)
struct DistState {
    double scale, err, sum;
    int32_t done, steps, converged, pad;
    double prev_scale, evaluated, reserved;
};
static_assert(sizeof(DistState) == 64, "pgh_dist state is 8 doubles");
int pgh_dist_state_init(double* state) {
    DistState* st = reinterpret_cast<DistState*>(state);
    *st = DistState{};
    st->scale = st->prev_scale = 1.0;
    return 0;
}
int pgh_dist_partial(pgh_graph_t g, pgh_vec_t xg_full, const double* state) {
    CHECK(g && xg_full && state, "pgh_dist_partial: null argument");
    if (reinterpret_cast<const DistState*>(state)->done) return 0;
    g->pending_xg = expand_gather(g, xg_full->data, xg_full->n);
    return 0;
}
// the double has no hot cache: stage 1 does nothing, stage 2 (and 0) expand the gather vector; hot_slots = 0 tells the
// caller that the whole vector must be there before stage 1
int pgh_dist_partial_stage(pgh_graph_t g, pgh_vec_t xg_full, const double* state, int32_t stage) {
    CHECK(g && xg_full && state && stage >= 0 && stage <= 2, "pgh_dist_partial_stage: bad argument");
    if (stage == 1) return 0;
    return pgh_dist_partial(g, xg_full, state);
}
int pgh_graph_hot_prefix(pgh_graph_t g, int32_t* hot_slots) {
    CHECK(g && hot_slots && g->gather_blk > 0, "pgh_graph_hot_prefix: not a partitioned graph");
    *hot_slots = 0;
    return 0;
}
static double ppr_step(const pgh_graph_s* g, const float* x, double xs, const float* p, double alpha, float* y);
int pgh_dist_combine(pgh_graph_t g, pgh_vec_t p, double alpha, pgh_vec_t y, pgh_vec_t xg_local, double* state) {
    CHECK(g && p && y && xg_local && state, "pgh_dist_combine: null argument");
    CHECK(p->n == g->n_cols && y->n == g->n_cols && xg_local->n == g->n_cols, "pgh_dist_combine: local vector length mismatch");
    DistState* st = reinterpret_cast<DistState*>(state);
    if (st->done) return 0;
    CHECK(g->pending_xg != nullptr, "pgh_dist_combine: no pgh_dist_partial before it");
    st->sum = ppr_step(g, g->pending_xg, st->scale, p->data, alpha, y->data);
    std::copy(y->data, y->data + y->n, xg_local->data);
    return 0;
}
static double absorb_step(const pgh_graph_s* g, const float* x, double xs, const float* p, const float* deg, const float* lam, float* y);
int pgh_dist_combine_absorb(pgh_graph_t g, pgh_vec_t p, pgh_vec_t deg, pgh_vec_t lam, pgh_vec_t y, pgh_vec_t xg_local, double* state) {
    CHECK(g && p && deg && lam && y && xg_local && state, "pgh_dist_combine_absorb: null argument");
    CHECK(p->n == g->n_cols && deg->n == g->n_cols && lam->n == g->n_cols && y->n == g->n_cols && xg_local->n == g->n_cols,
          "pgh_dist_combine_absorb: local vector length mismatch");
    DistState* st = reinterpret_cast<DistState*>(state);
    if (st->done) return 0;
    CHECK(g->pending_xg != nullptr, "pgh_dist_combine_absorb: no pgh_dist_partial before it");
    st->sum = absorb_step(g, g->pending_xg, st->scale, p->data, deg->data, lam->data, y->data);
    std::copy(y->data, y->data + y->n, xg_local->data);
    return 0;
}
int pgh_dist_combine_poly(pgh_graph_t g, pgh_vec_t term, pgh_vec_t term_out, double a, double b, pgh_vec_t result, double c, int32_t err_linf,
                          pgh_vec_t xg_local, double* state) {
    CHECK(g && term && term_out && result && xg_local && state, "pgh_dist_combine_poly: null argument");
    CHECK(term->n == g->n_cols && term_out->n == g->n_cols && result->n == g->n_cols && xg_local->n == g->n_cols,
          "pgh_dist_combine_poly: local vector length mismatch");
    DistState* st = reinterpret_cast<DistState*>(state);
    if (st->done) return 0;
    CHECK(g->pending_xg != nullptr, "pgh_dist_combine_poly: no pgh_dist_partial before it");
    double delta = 0;
    for (int64_t r = 0; r < g->n_cols; ++r) {
        float y = (float)a * row_dot(g, g->pending_xg, r);
        if (b != 0.0) y += (float)b * term->data[r];
        term_out->data[r] = y;
        const float r_old = result->data[r], r_new = r_old + (float)c * y;
        result->data[r] = r_new;
        const double d = std::fabs((double)r_new - (double)r_old);
        delta = err_linf ? std::max(delta, d) : delta + d;
    }
    st->err = delta;
    std::copy(term_out->data, term_out->data + term_out->n, xg_local->data);
    return 0;
}
// the host restatement processes every row in every step: nothing to watch
int pgh_dist_watch_isolated(pgh_graph_t g, pgh_vec_t p_local, pgh_vec_t y_start) {
    CHECK(g && p_local && y_start, "pgh_dist_watch_isolated: null argument");
    return 0;
}
int pgh_dist_release_isolated(pgh_graph_t) { return 0; }
int pgh_dist_close_sum(double* state, int32_t use_quotient) {
    DistState* st = reinterpret_cast<DistState*>(state);
    if (st->done) return 0;
    st->prev_scale = st->scale;
    st->scale = use_quotient ? (st->sum != 0.0 ? 1.0 / st->sum : 0.0) : 1.0;
    st->steps += 1;
    return 0;
}
// need lists: the host restatement keeps the dense gather layout (it has no hot / cold split at all): every count is zero, which the
// callers read as "this slice exchanges by all-gather"
int pgh_dist_need_counts(pgh_graph_t g, int64_t* counts) {
    CHECK(g && g->gather_blk > 0 && counts, "pgh_dist_need_counts: not a partitioned graph");
    for (int b = 0; b < g->gather_blocks; ++b) counts[b] = 0;
    return 0;
}
int pgh_dist_need_list(pgh_graph_t, int32_t, uint32_t*) { return fail("pgh_dist_need_list: the host restatement keeps the dense layout"); }
int pgh_dist_set_send_lists(pgh_graph_t g, const uint32_t*, const int32_t*, const int64_t* seg_offsets, int32_t segments) {
    CHECK(g && (segments == 0 || (seg_offsets && seg_offsets[segments] == 0)), "pgh_dist_set_send_lists: the host restatement keeps the dense layout");
    return 0;
}
int pgh_dist_pack(pgh_graph_t g, pgh_vec_t xg_local, pgh_vec_t send_buf) {
    CHECK(g && xg_local && send_buf, "pgh_dist_pack: null argument");
    return 0;                                               // no lists: nothing to pack
}
int pgh_dist_compact_from_dense(pgh_graph_t, int32_t, pgh_vec_t, int64_t, pgh_vec_t, int64_t) {
    return fail("pgh_dist_compact_from_dense: the host restatement keeps the dense layout");
}
int pgh_dist_residual(int32_t kind, pgh_vec_t y_new, pgh_vec_t y_old, double* state) {
    CHECK(y_new && y_old && state && y_new->n == y_old->n, "pgh_dist_residual: bad arguments");
    DistState* st = reinterpret_cast<DistState*>(state);
    if (st->done) return 0;
    st->err = scaled_res(kind == PGH_ERR_LINF ? PGH_ERR_LINF : PGH_ERR_L1, y_new->data, st->scale, y_old->data, st->prev_scale, y_new->n);
    return 0;
}
// The RCCL-driven loop of csrc/pgh_dist.hip has no host twin: the multi-rank tests on CPU go through the staged pgh_dist_*
// calls above with gloo collectives (pygrank_amd/distributed.py falls back to them when these refuse).
int pgh_graph_set_gather_bases_split(pgh_graph_t, const int64_t*, const int64_t*) {
    return fail("pgh_graph_set_gather_bases_split: not available in the host double");
}
int pgh_comm_unique_id(uint8_t*) { return fail("pgh_comm_unique_id: not available in the host double"); }
int pgh_comm_create(const uint8_t*, int32_t, int32_t, int32_t, pgh_comm_t*) { return fail("pgh_comm_create: not available in the host double"); }
int pgh_comm_create_external(int32_t, int32_t, pgh_allgather_fn, pgh_allreduce_fn, void*, pgh_comm_t*) {
    return fail("pgh_comm_create_external: not available in the host double");
}
int pgh_comm_set_alltoallv(pgh_comm_t, pgh_alltoallv_fn) { return fail("pgh_comm_set_alltoallv: not available in the host double"); }
int pgh_comm_destroy(pgh_comm_t) { return 0; }
int pgh_dist_ppr_run(pgh_graph_t, pgh_comm_t, pgh_vec_t, pgh_vec_t, const pgh_dist_cfg*, pgh_dist_result*) {
    return fail("pgh_dist_ppr_run: not available in the host double");
}
int pgh_dist_poly_run(pgh_graph_t, pgh_comm_t, pgh_vec_t, const double*, int32_t, pgh_vec_t, const pgh_dist_cfg*, pgh_dist_result*) {
    return fail("pgh_dist_poly_run: not available in the host double");
}
int pgh_dist_set_timeout(double) { return 0; }

int pgh_dist_close_err(double* state, int32_t kind, double tol, int64_t n_global) {
    DistState* st = reinterpret_cast<DistState*>(state);
    if (st->done) return 0;
    double e = st->err;
    if (kind == PGH_ERR_MABS) e /= (double)n_global;
    st->evaluated = e;
    if (e <= tol) st->done = st->converged = 1;
    return 0;
}
// the double keeps the normalisation inside the values, so its gather vector is the iterate itself
int pgh_ppr_step_dist(pgh_graph_t g, pgh_vec_t xg_full, double xs, pgh_vec_t p, double alpha, pgh_vec_t y, pgh_vec_t xg_local,
                      double* sum_y) {
    CHECK(g && xg_full && p && y && xg_local, "pgh_ppr_step_dist: null argument");
    CHECK(p->n == g->n_cols && y->n == g->n_cols && xg_local->n == g->n_cols, "pgh_ppr_step_dist: vector length mismatch");
    const double s = ppr_step(g, expand_gather(g, xg_full->data, xg_full->n), xs, p->data, alpha, y->data);
    std::copy(y->data, y->data + y->n, xg_local->data);
    if (sum_y) *sum_y = s;
    return 0;
}
int pgh_dist_prescale(pgh_graph_t g, pgh_vec_t x, pgh_vec_t out) {
    CHECK(g && x && out && x->n == g->n_cols && out->n == g->n_cols, "pgh_dist_prescale: length mismatch");
    std::copy(x->data, x->data + x->n, out->data);
    return 0;
}

}  // extern "C"
