// K1v -- cosine scores for latency-bound batches (B <= 8): S[b][m] = sum_k Q[b][k] * E[m][k].
//
// Same contract as sim_gemm.hip (np.dot(self.fact_embeddings, q.T) / np.dot(self.passage_embeddings,
// q.T), reference src/hipporag/HippoRAG.py:1459, :1496) for the B = 1 callers (get_fact_scores /
// dense_passage_retrieval seams, retrieve_ircot :526,539).  With so few queries the matrix cores have
// nothing to amortise: the job is streaming E from HBM once (F*D*2 bytes), so this is a GEMV:
//   * the queries live in registers (8 bf16 per lane per 512-element chunk of D);
//   * a wavefront streams R rows at a time, one 16-byte load per lane per chunk (a row of D = 768
//     is two fully coalesced 1-KiB / 512-B loads), all R*CH loads issued before the first use;
//   * products and sums by v_dot2c_f32_bf16 / v_dot2c_f32_f16 (fp32 accumulate);
//   * the R*BQ partial sums of a lane are reduced across the 64 lanes with a halving butterfly:
//     each step a lane hands half of its values to the lane `offset` away and keeps the other half
//     (31 shuffles for 32 values instead of 6 per value), then lanes write distinct outputs.
// Summation order differs from the MFMA kernel by design; both are within 3e-6 of fp64 (tests).
#include <algorithm>

#include "common.h"

namespace hrag {
namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <bool F16>
__device__ __forceinline__ float dot2(unsigned int e, unsigned int q, float acc) {
    if constexpr (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, e), __builtin_bit_cast(f16x2_t, q), acc, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, e), __builtin_bit_cast(bf16x2_t, q), acc, false);
}
template <bool F16>
__device__ __forceinline__ float dot8(const uint4 &e, const uint4 &q, float acc) {
    acc = dot2<F16>(e.x, q.x, acc);
    acc = dot2<F16>(e.y, q.y, acc);
    acc = dot2<F16>(e.z, q.z, acc);
    acc = dot2<F16>(e.w, q.w, acc);
    return acc;
}

template <int N>
struct Log2 {
    static constexpr int value = 1 + Log2<N / 2>::value;
};
template <>
struct Log2<1> {
    static constexpr int value = 0;
};

// vals[N] per lane -> after the call lane l holds in vals[0] the sum over all 64 lanes of value
// number (l >> (6 - log2 N)); N a power of two <= 64
template <int N>
__device__ __forceinline__ void halving_reduce(float (&vals)[N], int lane) {
    constexpr int LG = Log2<N>::value;
#pragma unroll
    for (int s = 0; s < LG; ++s) {
        const int offset = 32 >> s;
        const int half = N >> (s + 1);
        const bool upper = (lane & offset) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const float send = upper ? vals[j] : vals[j + half];
            const float keep = upper ? vals[j + half] : vals[j];
            vals[j] = keep + __shfl_xor(send, offset, 64);
        }
    }
#pragma unroll
    for (int offset = 32 >> LG; offset > 0; offset >>= 1) vals[0] += __shfl_xor(vals[0], offset, 64);
}

template <int BQ, int CH, int R, bool F16>
__global__ __launch_bounds__(256) void sim_gemv_kernel(const uint16_t *__restrict__ emb, int64_t rows,
                                                       int32_t dim, const uint16_t *__restrict__ q,
                                                       int32_t batch, float *__restrict__ out, int64_t ld) {
    constexpr int N = R * BQ;
    constexpr int LG = Log2<N>::value;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    uint4 qr[BQ][CH];
#pragma unroll
    for (int b = 0; b < BQ; ++b)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = c * 512 + lane * 8;
            qr[b][c] = make_uint4(0, 0, 0, 0);
            if (b < batch && k < dim) qr[b][c] = *reinterpret_cast<const uint4 *>(q + (size_t)b * dim + k);
        }
    const int64_t n_groups = (rows + R - 1) / R;
    for (int64_t g = wave; g < n_groups; g += n_waves) {
        const int64_t m0 = g * R;
        uint4 e[R][CH];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = c * 512 + lane * 8;
                e[r][c] = make_uint4(0, 0, 0, 0);
                if (m0 + r < rows && k < dim)
                {
                    const u32x4_t t = __builtin_nontemporal_load(
                        reinterpret_cast<const u32x4_t *>(emb + (size_t)(m0 + r) * dim + k));
                    e[r][c] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
        float vals[N];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) a = dot8<F16>(e[r][c], qr[b][c], a);
                vals[r * BQ + b] = a;
            }
        halving_reduce<N>(vals, lane);
        // lane l holds value (l >> (6 - LG)); one lane of each run of 2^(6-LG) lanes writes it
        if ((lane & ((64 >> LG) - 1)) == 0) {
            const int idx = lane >> (6 - LG);
            const int r = idx / BQ, b = idx % BQ;
            if (m0 + r < rows && b < batch) out[(size_t)b * ld + m0 + r] = vals[0];
        }
    }
}

template <int BQ, int CH, int R>
void launch_one(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q, int32_t batch, float *out,
                int64_t ld, hipStream_t s, bool f16) {
    const int64_t groups = ceil_div(rows, R);
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(groups, 4), 256 * 8);
    if (f16)
        hipLaunchKernelGGL((sim_gemv_kernel<BQ, CH, R, true>), dim3(blocks), dim3(256), 0, s, emb, rows, dim, q,
                           batch, out, ld);
    else
        hipLaunchKernelGGL((sim_gemv_kernel<BQ, CH, R, false>), dim3(blocks), dim3(256), 0, s, emb, rows, dim, q,
                           batch, out, ld);
}

}  // namespace

// returns false when no instantiation fits (the caller falls back to the MFMA kernel)
bool launch_sim_gemv(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q, int32_t batch,
                     float *out, int64_t ld, hipStream_t s, int32_t dtype) {
    const bool f16 = dtype == HRAG_FP16;
    if (batch < 1 || batch > 8 || dim % 8 != 0 || dim > 4096) return false;
    const int bq = batch <= 1 ? 1 : batch <= 2 ? 2 : batch <= 4 ? 4 : 8;
    const int ch = dim <= 512 ? 1 : dim <= 1024 ? 2 : dim <= 2048 ? 4 : 8;
#define GO(BQ, CH, R)                                                  \
    do {                                                               \
        launch_one<BQ, CH, R>(emb, rows, dim, q, batch, out, ld, s, f16); \
        return true;                                                   \
    } while (0)
    if (ch == 1) {
        if (bq == 1) GO(1, 1, 8);
        if (bq == 2) GO(2, 1, 8);
        if (bq == 4) GO(4, 1, 8);
        GO(8, 1, 4);
    }
    if (ch == 2) {
        if (bq == 1) GO(1, 2, 8);
        if (bq == 2) GO(2, 2, 8);
        if (bq == 4) GO(4, 2, 8);
        GO(8, 2, 4);
    }
    if (ch == 4) {
        if (bq == 1) GO(1, 4, 4);
        if (bq == 2) GO(2, 4, 4);
        if (bq == 4) GO(4, 4, 4);
        return false;
    }
    if (bq == 1) GO(1, 8, 2);
    if (bq == 2) GO(2, 8, 2);
#undef GO
    return false;
}

}  // namespace hrag
