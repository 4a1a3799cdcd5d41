/* Multi-threaded fp32 SpMM for bench.py's "vectorised" CPU baseline leg.  TEST / MEASUREMENT
 * INFRASTRUCTURE ONLY (see oracle/__init__.py): the best a CPU deployment of the same algorithm would
 * reasonably do, so that the GPU / CPU ratio is not inflated by the reference's Python loops.
 *
 *   y[i, :] = alpha * sum_k val[k] * x[col[k], :] + beta * v[i, :]        (row i = in-neighbours of i)
 *
 * x, v, y: row-major [n, b] fp32 (the b queries of a vertex contiguous), rows spread over all cores
 * with OpenMP.  The restatement of HippoRAG.py:1736-1743 (PRPACK) stays oracle/prpack_port.c; this is
 * the fixed-sweep power iteration the GPU path runs, in fp32.
 * Build: gcc -O3 -fopenmp -shared -fPIC (oracle/cpu_baseline.py). */
#include <stdint.h>
#include <string.h>

#define HRO_MAX_B 256

__attribute__((target_clones("avx512f", "avx2,fma", "default")))
void hro_spmm_f32(int64_t n, const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                  const float *v, float alpha, float beta, int32_t b, float *y) {
#pragma omp parallel for schedule(dynamic, 512)
    for (int64_t i = 0; i < n; ++i) {
        float acc[HRO_MAX_B];
        memset(acc, 0, sizeof(float) * (size_t)b);
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const float w = val[k];
            const float *xr = x + (int64_t)col[k] * b;
            for (int32_t j = 0; j < b; ++j) acc[j] += w * xr[j];
        }
        const float *vr = v + i * b;
        float *yr = y + i * b;
        for (int32_t j = 0; j < b; ++j) yr[j] = alpha * acc[j] + beta * vr[j];
    }
}
