// Index-time entity KNN: the pieces retrieve_knn needs besides the similarity GEMM and
// the row top-k that the retrieval path already has.
//
// Replaces, in reference src/hipporag/utils/embed_utils.py:6-94 (called by add_synonymy_edges,
// HippoRAG.py:959-1020):  torch.nn.functional.normalize (:25,28), torch.mm (:53), torch.topk (:55,73).
// The reference multiplies in fp32.  To stay within ~1e-6 of that on the bf16 matrix cores every vector
// is split x = hi + lo (both bf16, lo = bf16(x - hi)) and the product is accumulated from three MFMA
// passes  lo.hi + hi.lo + hi.hi  (the dropped lo.lo term is <= 2^-16 |x||y|); hrag_sim_gemm's
// `accumulate` flag adds a pass into the fp32 score matrix.
#include "common.h"

namespace hrag {
namespace {

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// one wavefront per row: L2-normalise like F.normalize (x / max(||x||, 1e-12)), then split
__global__ __launch_bounds__(256) void normalize_split_kernel(const float *__restrict__ x, int64_t rows,
                                                              int32_t dim, int32_t normalize,
                                                              uint16_t *__restrict__ hi,
                                                              uint16_t *__restrict__ lo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * dim;
    float inv = 1.f;
    if (normalize) {
        double ss = 0.0;
        for (int k = lane; k < dim; k += 64) ss += (double)xr[k] * (double)xr[k];
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv = 1.f / fmaxf((float)sqrt(ss), 1e-12f);
    }
    for (int k = lane; k < dim; k += 64) {
        const float v = normalize ? xr[k] * inv : xr[k];
        const uint16_t h = f32_to_bf16_rne(v);
        hi[(size_t)row * dim + k] = h;
        if (lo) lo[(size_t)row * dim + k] = f32_to_bf16_rne(v - __uint_as_float((uint32_t)h << 16));
    }
}

// fp32-faithful similarity (HRAG_F32_SPLIT): a vector x = hi + lo, both IEEE fp16 (11 + 11 significant bits: x to
// 2^-22 |x|; components are <= 1 in magnitude, unit vectors), is laid out over 3 * dim fp16 elements so that ONE fp16
// MFMA dot product of a stored row with a query is  hi.qhi + lo.qhi + hi.qlo  -- every partial product is exact in
// the fp32 accumulator, the dropped lo.qlo term is <= 2^-22 |x||q| -- through every similarity kernel unchanged:
//   embedding row: [hi | lo | hi]        query: [qhi | qhi | qlo]
// (bf16 halves, 8 + 8 bits, were tried first: 1.5e-6 off the reference's fp32 scores, enough to move the prior of
// the lowest-ranked passages by 1e-3 relative.)  One wavefront per row.
__global__ __launch_bounds__(256) void split3_kernel(const float *__restrict__ x, int64_t rows, int32_t dim,
                                                     int32_t as_query, int32_t normalize, uint16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * dim;
    _Float16 *o = reinterpret_cast<_Float16 *>(out) + (size_t)row * dim * 3;
    float inv = 1.f;
    if (normalize) {      // F.normalize: x / max(||x||, 1e-12)  (embed_utils.py:25,28)
        double ss = 0.0;
        for (int k = lane; k < dim; k += 64) ss += (double)xr[k] * (double)xr[k];
        for (int o2 = 32; o2 > 0; o2 >>= 1) ss += __shfl_xor(ss, o2, 64);
        inv = 1.f / fmaxf((float)sqrt(ss), 1e-12f);
    }
    for (int k = lane; k < dim; k += 64) {
        const float v = normalize ? xr[k] * inv : xr[k];
        const _Float16 h = (_Float16)v;                 // round to nearest even
        const _Float16 l = (_Float16)(v - (float)h);
        o[k] = h;
        o[dim + k] = as_query ? h : l;
        o[2 * dim + k] = as_query ? l : h;
    }
}

}  // namespace

hrag_status launch_split3(const float *x, int64_t rows, int32_t dim, int32_t as_query, uint16_t *out, hipStream_t s,
                          int32_t normalize) {
    if (rows <= 0) return HRAG_OK;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, rows, dim, as_query,
                       normalize, out);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}
}  // namespace hrag

using namespace hrag;

extern "C" {

hrag_status hrag_normalize_split_bf16(const float *x_dev, int64_t rows, int32_t dim, int32_t normalize,
                                      uint16_t *hi_dev, uint16_t *lo_dev, hrag_stream stream) {
    HRAG_REQUIRE(x_dev && hi_dev && rows >= 0 && dim > 0, "bad argument");
    if (rows == 0) return HRAG_OK;
    hipLaunchKernelGGL(normalize_split_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0,
                       (hipStream_t)stream, x_dev, rows, dim, normalize, hi_dev, lo_dev);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status hrag_split_f32(const float *x_dev, int64_t rows, int32_t dim, int32_t as_query, int32_t normalize,
                           uint16_t *out_dev, hrag_stream stream) {
    HRAG_REQUIRE(x_dev && out_dev && rows >= 0 && dim > 0, "bad argument");
    return launch_split3(x_dev, rows, dim, as_query, out_dev, (hipStream_t)stream, normalize);
}

int64_t hrag_sim_topk_workspace_bytes(int64_t rows, int32_t batch) {
    if (rows < 1 || batch < 1) return 0;
    return (2 * sim_fused_tiles(rows) * (int64_t)batch + 2 * (int64_t)batch) * (int64_t)sizeof(float) +
           sim_fused_sel_ints(batch) * (int64_t)sizeof(int32_t);
}

hrag_status hrag_sim_topk(const uint16_t *emb_dev, int64_t rows, int32_t dim, const uint16_t *q_dev, int32_t batch,
                          int32_t k, int32_t dtype, void *workspace_dev, int64_t workspace_bytes, int32_t *idx_out_dev,
                          float *val_out_dev, hrag_stream stream) {
    HRAG_REQUIRE(emb_dev && q_dev && workspace_dev && idx_out_dev && val_out_dev && rows >= 1 && batch >= 1, "bad argument");
    HRAG_REQUIRE(dtype == HRAG_BF16 || dtype == HRAG_FP16, "dtype must be HRAG_BF16 or HRAG_FP16");
    HRAG_REQUIRE(workspace_bytes >= hrag_sim_topk_workspace_bytes(rows, batch), "workspace too small: %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)hrag_sim_topk_workspace_bytes(rows, batch));
    // the per-query records first: they are what has to stay zero between calls, and a call with a smaller batch then
    // uses a prefix of the same records (the float scratch behind them is rewritten by every call)
    int32_t *sel = static_cast<int32_t *>(workspace_dev);
    float *mn = reinterpret_cast<float *>(sel + sim_fused_sel_ints(batch)), *mx = mn + batch;
    float *ws = mx + batch;
    return launch_sim_topk_fused(emb_dev, rows, dim, q_dev, batch, k, 0, 0, ws, sel, mn, mx, idx_out_dev, val_out_dev,
                                 (hipStream_t)stream, dtype);
}

hrag_status hrag_sim_topk_min_score(const uint16_t *emb_dev, int64_t rows, int32_t dim, const uint16_t *q_dev, int32_t batch,
                                    int32_t k, int32_t dtype, int32_t approx_dim, float min_score, float margin,
                                    void *workspace_dev, int64_t workspace_bytes, int32_t *idx_out_dev, float *val_out_dev,
                                    int32_t *overflow_out_dev, hrag_stream stream) {
    HRAG_REQUIRE(emb_dev && q_dev && workspace_dev && idx_out_dev && val_out_dev && overflow_out_dev && rows >= 1 && batch >= 1,
                 "bad argument");
    HRAG_REQUIRE(dtype == HRAG_BF16 || dtype == HRAG_FP16, "dtype must be HRAG_BF16 or HRAG_FP16");
    HRAG_REQUIRE(margin >= 0.f && margin < 1.f && min_score == min_score, "margin %g must lie in [0, 1) and min_score be a number",
                 (double)margin);
    HRAG_REQUIRE(approx_dim == 0 || margin > 0.f, "a prefix first pass (approx_dim %d) needs the margin that bounds its error", approx_dim);
    HRAG_REQUIRE(workspace_bytes >= hrag_sim_topk_workspace_bytes(rows, batch), "workspace too small: %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)hrag_sim_topk_workspace_bytes(rows, batch));
    int32_t *sel = static_cast<int32_t *>(workspace_dev);
    float *mn = reinterpret_cast<float *>(sel + sim_fused_sel_ints(batch)), *mx = mn + batch;
    float *ws = mx + batch;
    return launch_sim_topk_fused(emb_dev, rows, dim, q_dev, batch, k, 0, 0, ws, sel, mn, mx, idx_out_dev, val_out_dev,
                                 (hipStream_t)stream, dtype, approx_dim, min_score - margin, overflow_out_dev);
}

hrag_status hrag_sim_gemm(const uint16_t *emb_dev, int64_t rows, int32_t dim, const uint16_t *q_dev,
                          int32_t batch, float *out_dev, int64_t ld, int32_t accumulate, int32_t dtype,
                          hrag_stream stream) {
    HRAG_REQUIRE(emb_dev && q_dev && out_dev && rows >= 0 && batch >= 0 && ld >= rows, "bad argument");
    HRAG_REQUIRE(dtype == HRAG_BF16 || dtype == HRAG_FP16, "dtype must be HRAG_BF16 or HRAG_FP16");
    return launch_sim_gemm(emb_dev, rows, dim, q_dev, batch, out_dev, ld, (hipStream_t)stream, accumulate, dtype);
}

}  // extern "C"
