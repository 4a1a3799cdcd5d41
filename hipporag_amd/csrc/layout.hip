// Layout changes between the caller-facing [B, n] row-major matrices (one row per query, as the
// reference hands vectors to run_ppr, HippoRAG.py:1709-1711) and the engine's PPR slab layout
// [n_slabs][n][BC], with the element-wise stages of the path fused into the copy:
//   rows_to_slab  kSanitize   : reset_prob NaN / negative -> 0              (HippoRAG.py:1735)
//                 kMinMaxScale: passage_weights = minmax(dpr) * passage_node_weight
//                               (utils/misc_utils.py:130-139, HippoRAG.py:1627-1633)
//   slab_to_rows              : doc_scores = pagerank[passage_node_idxs] with the final
//                               normalisation sum(x) = 1 (HippoRAG.py:1745), or the normalised
//                               DPR scores for queries on the DPR fallback (:467-469).
// Tiles go through LDS so that both the global reads and the global writes are coalesced.
#include <algorithm>

#include "common.h"

namespace hrag {
namespace {

constexpr int TI = 64;  // i-extent (passages / vertices) of a tile

__device__ __forceinline__ float minmax_norm(float s, float mn, float mx) {
    const float range = mx - mn;
    // all-equal -> ones (misc_utils.py:136-137); IEEE division like numpy, not a reciprocal
    return range == 0.f ? 1.f : __fdiv_rn(s - mn, range);
}

template <int G>
__global__ __launch_bounds__(256) void rows_to_slab_kernel(
    const float *__restrict__ rows, int64_t ld, int64_t n, int32_t batch, int mode,
    const float *__restrict__ mn, const float *__restrict__ mx, float scale,
    const int32_t *__restrict__ skip_flags, float *__restrict__ slab, int64_t slab_rows,
    const float *__restrict__ qscale) {
    constexpr int BC = 4 * G;
    __shared__ float tile[BC][TI + 1];
    const int tid = threadIdx.x;
    const int s = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * TI;
    // load: one query row segment of TI consecutive elements per 64 threads
    for (int qq = tid / TI; qq < BC; qq += 256 / TI) {
        const int ii = tid % TI;
        const int q = s * BC + qq;
        const int64_t i = i0 + ii;
        float v = 0.f;
        if (q < batch && i < n && !(skip_flags && (skip_flags[q] & 1))) {
            v = rows[(size_t)q * ld + i];
            if (mode == kSanitize) {
                v = (v != v || v < 0.f) ? 0.f : v;
            } else {
                v = minmax_norm(v, mn[q], mx[q]) * scale;
                if (qscale) v *= qscale[q];  // a power of two: exact
            }
        }
        tile[qq][ii] = v;
    }
    __syncthreads();
    float4 *out = reinterpret_cast<float4 *>(slab + (size_t)s * slab_rows * BC);
    for (int t = tid; t < TI * G; t += 256) {
        const int ii = t / G, gl = t % G;
        const int64_t i = i0 + ii;
        if (i < n) {
            float4 v;
            v.x = tile[4 * gl + 0][ii];
            v.y = tile[4 * gl + 1][ii];
            v.z = tile[4 * gl + 2][ii];
            v.w = tile[4 * gl + 3][ii];
            out[(size_t)i * G + gl] = v;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void slab_to_rows_kernel(
    const float *__restrict__ slab, int64_t slab_rows, const int32_t *__restrict__ gather, int64_t n,
    int32_t batch, const double *__restrict__ sums, float *__restrict__ out, int64_t ld,
    const float *__restrict__ alt, int64_t alt_ld, const float *__restrict__ mn,
    const float *__restrict__ mx, const int32_t *__restrict__ flags) {
    constexpr int BC = 4 * G;
    __shared__ float tile[BC][TI + 1];
    const int tid = threadIdx.x;
    const int s = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * TI;
    const float4 *xs = reinterpret_cast<const float4 *>(slab + (size_t)s * slab_rows * BC);
    for (int t = tid; t < TI * G; t += 256) {
        const int ii = t / G, gl = t % G;
        const int64_t i = i0 + ii;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const int64_t src = gather ? (int64_t)gather[i] : i;
            v = xs[(size_t)src * G + gl];
        }
        tile[4 * gl + 0][ii] = v.x;
        tile[4 * gl + 1][ii] = v.y;
        tile[4 * gl + 2][ii] = v.z;
        tile[4 * gl + 3][ii] = v.w;
    }
    __syncthreads();
    for (int qq = tid / TI; qq < BC; qq += 256 / TI) {
        const int ii = tid % TI;
        const int q = s * BC + qq;
        const int64_t i = i0 + ii;
        if (q >= batch || i >= n) continue;
        float r;
        if (alt && flags && (flags[q] & 1)) {
            r = minmax_norm(alt[(size_t)q * alt_ld + i], mn[q], mx[q]);
        } else {
            const double sm = sums[q];
            r = sm > 0.0 ? (float)((double)tile[qq][ii] / sm) : 0.f;
        }
        out[(size_t)q * ld + i] = r;
    }
}

// out[i][:] = src[i] >= 0 ? emb[src[i]][:] : fresh[-src[i] - 1][:]   -- rows of row_bytes (multiple of 16) bytes
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t *__restrict__ emb, const uint8_t *__restrict__ fresh,
                                                          const int32_t *__restrict__ src, int64_t n, int32_t row_bytes,
                                                          uint8_t *__restrict__ out) {
    const int chunks = row_bytes / 16;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = t / chunks;
    const int c = (int)(t % chunks);
    if (i >= n) return;
    const int32_t r = src[i];
    const uint8_t *from = r >= 0 ? emb + (size_t)r * row_bytes : fresh + (size_t)(-r - 1) * row_bytes;
    reinterpret_cast<int4 *>(out + (size_t)i * row_bytes)[c] = reinterpret_cast<const int4 *>(from)[c];
}

__global__ void fill_i32_kernel(int32_t *dst, int32_t value, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

// BlitList (common.h): operation blockIdx.y, grid-stride over its 16-byte (or, unaligned, 4-byte) units
__global__ __launch_bounds__(256) void blit_kernel(const BlitList l) {
    const BlitOp op = l.op[blockIdx.y];
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    const bool v16 = (((uintptr_t)op.dst | (uintptr_t)op.src | (uintptr_t)op.bytes | (uintptr_t)op.stride) & 15) == 0;
    if (v16) {
        const int64_t per = op.bytes >> 4, total = per * op.reps;
        for (int64_t u = t0; u < total; u += step) {
            const int64_t rep = u / per, off = u - rep * per;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (op.src) v = reinterpret_cast<const uint4 *>(op.src)[off];
            reinterpret_cast<uint4 *>(static_cast<char *>(op.dst) + rep * op.stride)[off] = v;
        }
    } else {
        const int64_t per = op.bytes >> 2, total = per * op.reps;
        for (int64_t u = t0; u < total; u += step) {
            const int64_t rep = u / per, off = u - rep * per;
            uint32_t v = 0u;
            if (op.src) v = reinterpret_cast<const uint32_t *>(op.src)[off];
            reinterpret_cast<uint32_t *>(static_cast<char *>(op.dst) + rep * op.stride)[off] = v;
        }
    }
}

__global__ void flag_zero_mass_kernel(const double *sums, int32_t batch, int32_t *flags, int32_t bit) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    if (!(sums[q] > 0.0) && !(flags[q] & 1)) flags[q] |= bit;  // fallback queries carry no PPR mass
}

}  // namespace

hrag_status launch_rows_to_slab(const float *rows, int64_t ld, int64_t n, int32_t batch,
                                ToSlabMode mode, const float *mn, const float *mx, float scale,
                                const int32_t *skip_flags, float *slab, SlabLayout lay,
                                hipStream_t s, int64_t slab_rows, const float *qscale) {
    if (n == 0) return HRAG_OK;
    if (slab_rows <= 0) slab_rows = n;
    dim3 grid((unsigned)ceil_div(n, TI), (unsigned)lay.n_slabs);
#define CALL(G)                                                                                    \
    hipLaunchKernelGGL(rows_to_slab_kernel<G>, grid, dim3(256), 0, s, rows, ld, n, batch, (int)mode, \
                       mn, mx, scale, skip_flags, slab, slab_rows, qscale)
    switch (lay.bc) {
        case 4: CALL(1); break;
        case 8: CALL(2); break;
        case 16: CALL(4); break;
        case 32: CALL(8); break;
        case 64: CALL(16); break;
        default: set_error("unsupported slab width %d", lay.bc); return HRAG_EINVAL;
    }
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_slab_to_rows(const float *slab, int64_t slab_rows, const int32_t *gather,
                                int64_t n, int32_t batch, const double *sums, float *out,
                                int64_t ld, const float *alt, int64_t alt_ld, const float *mn,
                                const float *mx, const int32_t *flags, SlabLayout lay,
                                hipStream_t s) {
    if (n == 0) return HRAG_OK;
    dim3 grid((unsigned)ceil_div(n, TI), (unsigned)lay.n_slabs);
#define CALL(G)                                                                                   \
    hipLaunchKernelGGL(slab_to_rows_kernel<G>, grid, dim3(256), 0, s, slab, slab_rows, gather, n, \
                       batch, sums, out, ld, alt, alt_ld, mn, mx, flags)
    switch (lay.bc) {
        case 4: CALL(1); break;
        case 8: CALL(2); break;
        case 16: CALL(4); break;
        case 32: CALL(8); break;
        case 64: CALL(16); break;
        default: set_error("unsupported slab width %d", lay.bc); return HRAG_EINVAL;
    }
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_gather_rows(const void *emb, const void *fresh, const int32_t *src, int64_t n, int32_t row_bytes,
                               void *out, hipStream_t s) {
    if (n <= 0) return HRAG_OK;
    const int64_t threads = n * (row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, s,
                       static_cast<const uint8_t *>(emb), static_cast<const uint8_t *>(fresh), src, n, row_bytes,
                       static_cast<uint8_t *>(out));
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_fill_i32(int32_t *dst, int32_t value, int64_t n, hipStream_t s) {
    if (n <= 0) return HRAG_OK;
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, dst, value, n);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_blits(const BlitList &l, hipStream_t s) {
    HRAG_REQUIRE(!l.overflow, "internal: more than 8 operations in one BlitList (a fill or copy would be dropped)");
    if (l.n <= 0) return HRAG_OK;
    int64_t most = 1;
    for (int i = 0; i < l.n; ++i) {
        HRAG_REQUIRE((l.op[i].bytes & 3) == 0 && (l.op[i].stride & 3) == 0, "blit sizes must be multiples of 4 bytes");
        most = std::max<int64_t>(most, l.op[i].bytes * l.op[i].reps / 16);
    }
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(most, 256 * 4), 2048);
    hipLaunchKernelGGL(blit_kernel, dim3(std::max(gx, 1u), (unsigned)l.n), dim3(256), 0, s, l);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_flag_zero_mass(const double *sums, int32_t batch, int32_t *flags, int32_t bit,
                                  hipStream_t s) {
    hipLaunchKernelGGL(flag_zero_mass_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, s,
                       sums, batch, flags, bit);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}


namespace {
// one thread per (passage, query): slab layout [n_slabs][slab_rows][bc]
__global__ __launch_bounds__(256) void passage_delta_kernel(const float *__restrict__ x, const float *__restrict__ xp,
                                                            int64_t slab_rows, const int32_t *__restrict__ pv,
                                                            int64_t n_passages, int32_t batch, int32_t bc, int32_t *est) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = t / batch;
    const int q = (int)(t - p * batch);
    if (p >= n_passages) return;
    const size_t at = ((size_t)(q / bc) * slab_rows + (size_t)pv[p]) * bc + (size_t)(q % bc);
    const float a = x[at];
    const float r = a > 0.f ? fabsf(a - xp[at]) / a : 0.f;
    const int bits = __float_as_int(r);
    if (bits > est_peek(est + q)) atomicMax(&est[q], bits);   // rare path (fp32 slab state): a coherent pre-check is fine
}
}  // namespace

hrag_status launch_passage_delta(const float *x, const float *x_prev, int64_t slab_rows, const int32_t *passage_vertex,
                                 int64_t n_passages, int32_t batch, SlabLayout lay, int32_t *est, hipStream_t s) {
    if (n_passages <= 0 || batch <= 0) return HRAG_OK;
    hipLaunchKernelGGL(passage_delta_kernel, dim3((unsigned)ceil_div(n_passages * batch, 256)), dim3(256), 0, s, x,
                       x_prev, slab_rows, passage_vertex, n_passages, batch, lay.bc, est);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
