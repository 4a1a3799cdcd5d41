/*
 * Oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py): C restatement of
 * the PRPACK solvers igraph dispatches to for the reference's only PPR call,
 *   src/hipporag/HippoRAG.py:1736-1743
 *   graph.personalized_pagerank(..., directed=False, weights='weight',
 *                               reset=reset_prob, implementation='prpack')
 *
 * PARITY UNPINNED.  PRPACK is vendored inside python_igraph==0.11.8
 * (requirements.txt:9), which is absent from /root/reference and not
 * installable here; what follows restates its published algorithm
 * (D. Gleich / D. Kurokawa, "prpack": prpack_solver::solve_via_gs and
 * solve_via_ge) from the description in SURVEY.md section 8a:
 *   - reset r >= 0 is normalised to v = r / sum(r) and used for BOTH the
 *     teleport vector v and the dangling distribution u;
 *   - weights are normalised per source vertex (the CSR handed in here is
 *     already P[i][j] = A[i][j] / colsum_j, row i listing in-neighbours j);
 *   - fewer than 128 vertices: dense Gaussian elimination of
 *         (I - alpha (P + u d^T)) x = (1 - alpha) v
 *     otherwise Gauss-Seidel sweeps from x = 0 with the running dangling mass
 *     "delta", stopping when the not-yet-distributed probability mass
 *         err = 1 - sum(x)          (Kahan-compensated)
 *     drops below tol (igraph passes 1e-10);
 *   - the result is scaled to sum 1.
 * Single-threaded, like PRPACK.  Used (a) by tests to cross-check the numpy
 * oracle and (b) as the PPR stage of bench.py's cpu_baseline ("port").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HRO_OK 0
#define HRO_EINVAL 1
#define HRO_ENOMEM 2
#define HRO_EZERO 3
#define HRO_ENOCONV 4

static int normalise_reset(int64_t n, const double *reset, double *v)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double r = reset[i];
        if (isnan(r) || r < 0.0) r = 0.0; /* HippoRAG.py:1735 */
        v[i] = r;
        s += r;
    }
    if (!(s > 0.0)) return HRO_EZERO;
    for (int64_t i = 0; i < n; ++i) v[i] /= s;
    return HRO_OK;
}

/* dangling[i] = 1 when vertex i has no outgoing weight (column i of P is empty).
 * For the symmetric adjacency of an undirected graph that is the same as row i
 * being empty, but compute it from the columns to stay general. */
static void find_dangling(int64_t n, const int64_t *rowptr, const int32_t *col, unsigned char *d)
{
    memset(d, 1, (size_t)n);
    for (int64_t e = 0; e < rowptr[n]; ++e) d[col[e]] = 0;
}

int hro_ppr_gs(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val,
               const double *reset, double alpha, double tol, int max_sweeps,
               double *x, int *sweeps_out)
{
    if (n <= 0 || !rowptr || !reset || !x) return HRO_EINVAL;
    double *v = (double *)malloc((size_t)n * sizeof(double));
    unsigned char *d = (unsigned char *)malloc((size_t)n);
    if (!v || !d) { free(v); free(d); return HRO_ENOMEM; }
    int rc = normalise_reset(n, reset, v);
    if (rc != HRO_OK) { free(v); free(d); return rc; }
    find_dangling(n, rowptr, col, d);

    for (int64_t i = 0; i < n; ++i) x[i] = 0.0;
    double delta = 0.0;          /* alpha * (mass currently sitting on dangling vertices) */
    double err = 1.0, comp = 0.0; /* 1 - sum(x), compensated */
    int sweeps = 0;
    do {
        for (int64_t i = 0; i < n; ++i) {
            double nv = 0.0;
            for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) nv += x[col[e]] * val[e];
            nv = alpha * nv + (1.0 - alpha) * v[i];
            if (d[i]) delta -= alpha * x[i];
            nv += delta * v[i];
            if (d[i]) nv /= 1.0 - alpha * v[i];
            if (d[i]) delta += alpha * nv;
            /* err -= (nv - x[i]) with Kahan compensation */
            double y = (x[i] - nv) - comp;
            double t = err + y;
            comp = (t - err) - y;
            err = t;
            x[i] = nv;
        }
        ++sweeps;
    } while (err >= tol && sweeps < max_sweeps);
    if (sweeps_out) *sweeps_out = sweeps;

    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += x[i];
    for (int64_t i = 0; i < n; ++i) x[i] /= s;
    free(v);
    free(d);
    return err < tol ? HRO_OK : HRO_ENOCONV;
}

int hro_ppr_ge(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val,
               const double *reset, double alpha, double *x)
{
    if (n <= 0 || n > 4096 || !rowptr || !reset || !x) return HRO_EINVAL;
    double *v = (double *)malloc((size_t)n * sizeof(double));
    unsigned char *d = (unsigned char *)malloc((size_t)n);
    double *m = (double *)calloc((size_t)n * (size_t)(n + 1), sizeof(double));
    if (!v || !d || !m) { free(v); free(d); free(m); return HRO_ENOMEM; }
    int rc = normalise_reset(n, reset, v);
    if (rc != HRO_OK) { free(v); free(d); free(m); return rc; }
    find_dangling(n, rowptr, col, d);
    const int64_t w = n + 1;
    for (int64_t i = 0; i < n; ++i) {
        m[i * w + i] = 1.0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) m[i * w + col[e]] -= alpha * val[e];
        for (int64_t j = 0; j < n; ++j)
            if (d[j]) m[i * w + j] -= alpha * v[i];
        m[i * w + n] = (1.0 - alpha) * v[i];
    }
    /* Gaussian elimination with partial pivoting */
    for (int64_t k = 0; k < n; ++k) {
        int64_t piv = k;
        double best = fabs(m[k * w + k]);
        for (int64_t i = k + 1; i < n; ++i)
            if (fabs(m[i * w + k]) > best) { best = fabs(m[i * w + k]); piv = i; }
        if (best == 0.0) { free(v); free(d); free(m); return HRO_EINVAL; }
        if (piv != k)
            for (int64_t j = k; j <= n; ++j) {
                double t = m[k * w + j]; m[k * w + j] = m[piv * w + j]; m[piv * w + j] = t;
            }
        for (int64_t i = k + 1; i < n; ++i) {
            double f = m[i * w + k] / m[k * w + k];
            if (f == 0.0) continue;
            for (int64_t j = k; j <= n; ++j) m[i * w + j] -= f * m[k * w + j];
        }
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double s = m[i * w + n];
        for (int64_t j = i + 1; j < n; ++j) s -= m[i * w + j] * x[j];
        x[i] = s / m[i * w + i];
    }
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += x[i];
    for (int64_t i = 0; i < n; ++i) x[i] /= s;
    free(v); free(d); free(m);
    return HRO_OK;
}

/* igraph's dispatch: dense GE below 128 vertices, Gauss-Seidel (tol 1e-10) otherwise. */
int hro_ppr_prpack(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val,
                   const double *reset, double alpha, double *x, int *sweeps_out)
{
    if (n < 128) {
        if (sweeps_out) *sweeps_out = 0;
        return hro_ppr_ge(n, rowptr, col, val, reset, alpha, x);
    }
    return hro_ppr_gs(n, rowptr, col, val, reset, alpha, 1e-10, 1000, x, sweeps_out);
}
