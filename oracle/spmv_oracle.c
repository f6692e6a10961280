/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the arithmetic behind the reference's hot op.
 *
 * pygrank/core/backend/numpy.py:64-65 evaluates conv(signal, M) as `signal @ M`; scipy dispatches that to
 * `M.T` (a CSC view of the CSR arrays) times a vector, i.e. scipy.sparse._sparsetools.csc_matvec
 * (scipy 1.15.3 here, un-pinned in the reference's setup.py:28-30).  Published algorithm of csc_matvec:
 *     for j in columns: for k in Ap[j]..Ap[j+1]: y[Ai[k]] += Ax[k] * x[j]
 * With the CSR arrays of M standing in as the CSC arrays of M^T, j runs over the ROWS of M: a single-threaded
 * fp64 scatter-add.  oracle_csc_matvec restates exactly that (used as cpu_baseline kind "port");
 * oracle_pull_spmv_omp is the all-core pull formulation over CSR(M^T) ("best-effort CPU").
 * Pinned by tests/test_oracle_c.py against scipy and the golden conv fixtures.
 */
#include <stdint.h>
#include <string.h>

void oracle_csc_matvec(int64_t n_rows, int64_t n_cols, const int32_t* indptr, const int32_t* indices,
                       const double* data, const double* x, double* y) {
    memset(y, 0, sizeof(double) * (size_t)n_cols);
    for (int64_t j = 0; j < n_rows; ++j) {
        const double xj = x[j];
        for (int32_t k = indptr[j]; k < indptr[j + 1]; ++k) y[indices[k]] += data[k] * xj;
    }
}

/* one PageRank step around it: adhoc.py:36 + abstract_filters.py:133-134 + supervised.py:101-106 */
double oracle_pagerank_step(int64_t n, const int32_t* indptr, const int32_t* indices, const double* data,
                            const double* x, const double* p, double alpha, int use_quotient, double* y,
                            double* l1_residual) {
    oracle_csc_matvec(n, n, indptr, indices, data, x, y);
    double sum = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        y[i] = y[i] * alpha + p[i] * (1.0 - alpha);
        sum += y[i];
    }
    double res = 0.0;
    const double inv = (use_quotient && sum != 0.0) ? 1.0 / sum : 1.0;
    for (int64_t i = 0; i < n; ++i) {
        y[i] *= inv;
        const double d = y[i] - x[i];
        res += d < 0 ? -d : d;
    }
    if (l1_residual) *l1_residual = res;
    return sum;
}

/* all-core pull SpMV over CSR(M^T): y[i] = sum_k valT[k] * x[colT[k]] */
void oracle_pull_spmv_omp(int64_t n, const int64_t* indptr_t, const int32_t* indices_t, const double* data_t,
                          const double* x, double* y) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int64_t k = indptr_t[i]; k < indptr_t[i + 1]; ++k) acc += data_t[k] * x[indices_t[k]];
        y[i] = acc;
    }
}
