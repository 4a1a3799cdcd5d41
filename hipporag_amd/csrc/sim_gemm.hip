// K1 -- cosine-similarity scores S[b][m] = sum_k Q[b][k] * E[m][k] on the bf16 matrix cores.
//
// Replaces np.dot(self.fact_embeddings, q.T) / np.dot(self.passage_embeddings, q.T)
// (reference src/hipporag/HippoRAG.py:1459, :1496; StandardRAG.py:422) for a batch of B queries.
// Embeddings are unit-norm rows rounded to bf16; products are exact in fp32, accumulation is fp32
// (v_mfma_f32_16x16x32_bf16), so the only difference to an fp64 dot product is summation order.
//
// Both operands are K-contiguous ("B^T" form): A = E tile (MFMA rows i <-> embedding rows m),
// B = Q tile (MFMA cols j <-> queries b).  Each lane feeds 8 consecutive k of one row to the MFMA
// for A and B alike, so the result does not depend on how the hardware numbers k inside a
// fragment.  C/D layout (cdna_hip_programming.md section 3): col = lane & 15, row = 4*(lane>>4)+reg,
// i.e. a lane owns 4 consecutive embedding rows of one query -> one float4 store into S[b][m..m+3].
//
// Tile: 128 embedding rows x BN queries per 256-thread workgroup, BK = 64 (one full 128-byte line
// of every row per step), global -> registers -> LDS with the next tile's loads issued before the
// MFMA block (T14 split), LDS rows padded to 144 bytes (conflict-free ds_read_b128 for 16 rows).
// The embedding matrix is streamed from HBM once: the BN-tiles of one row tile are adjacent in
// dispatch order so that they meet in the Infinity Cache.
#include "common.h"

namespace hrag {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int LDS_LD = BK + 8;  // bf16 elements per LDS row (144 bytes)

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256) void sim_gemm_kernel(const uint16_t *__restrict__ emb,
                                                       int64_t rows, int32_t dim,
                                                       const uint16_t *__restrict__ q,
                                                       int32_t batch, float *__restrict__ out,
                                                       int64_t ld, int32_t n_tiles_n, int32_t accumulate) {
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    constexpr int MI = BM / (WM * 16);
    constexpr int NJ = BN / (WN * 16);
    constexpr int A_PASSES = BM / 32;
    constexpr int B_PASSES = (BN + 31) / 32;
    __shared__ __attribute__((aligned(16))) uint16_t As[BM][LDS_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BN][LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t tile = blockIdx.x;
    const int nt = (int)(tile % n_tiles_n);
    const int64_t mt = tile / n_tiles_n;
    const int64_t m0 = mt * BM;
    const int b0 = nt * BN;

    const int ld_row = tid >> 3;        // 0..31
    const int ld_chunk = (tid & 7) * 8; // element offset inside the BK slice

    uint4 ra[A_PASSES], rb[B_PASSES];
    auto load_tile = [&](int k0) {
        const int kk = k0 + ld_chunk;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            const int64_t gm = m0 + ld_row + 32 * p;
            ra[p] = make_uint4(0, 0, 0, 0);
            if (gm < rows && kk < dim)
                ra[p] = *reinterpret_cast<const uint4 *>(emb + (size_t)gm * dim + kk);
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int r = ld_row + 32 * p;
            const int gb = b0 + r;
            rb[p] = make_uint4(0, 0, 0, 0);
            if (r < BN && gb < batch && kk < dim)
                rb[p] = *reinterpret_cast<const uint4 *>(q + (size_t)gb * dim + kk);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            *reinterpret_cast<uint4 *>(&As[ld_row + 32 * p][ld_chunk]) = ra[p];
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int r = ld_row + 32 * p;
            if (r < BN) *reinterpret_cast<uint4 *>(&Bs[r][ld_chunk]) = rb[p];
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    for (int k0 = 0; k0 < dim; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < dim) load_tile(k0 + BK);  // in flight while the MFMAs below run
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            bf16x8 a[MI], b[NJ];
            const int kcol = s * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const bf16x8 *>(&As[(wm * MI + i) * 16 + (lane & 15)][kcol]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const bf16x8 *>(&Bs[(wn * NJ + j) * 16 + (lane & 15)][kcol]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane owns rows m..m+3 of query gb for every (i, j) fragment
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + (wm * MI + i) * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gb = b0 + (wn * NJ + j) * 16 + (lane & 15);
            if (gb >= batch || m >= rows) continue;
            float *dst = out + (size_t)gb * ld + m;
            f32x4 v = acc[i][j];
            if (accumulate) {   // multi-pass products (bf16x3 KNN): out += this pass
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) v[r] += dst[r];
            }
            if (m + 3 < rows && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) dst[r] = v[r];
            }
        }
    }
}

}  // namespace

hrag_status launch_sim_gemm(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q,
                            int32_t batch, float *out, int64_t ld, hipStream_t s, int32_t accumulate) {
    HRAG_REQUIRE(dim > 0 && dim % 8 == 0, "embedding dim %d must be a positive multiple of 8", dim);
    if (rows == 0 || batch == 0) return HRAG_OK;
    // latency path: GEMV.  Measured at F = 875k, D = 768: B = 1 0.27 ms (MFMA kernel 0.70), B = 2 0.31;
    // from B = 4 the per-lane dot products make it VALU-bound (0.96 ms) and the MFMA kernel wins.
    if (batch <= 2 && !accumulate && launch_sim_gemv(emb, rows, dim, q, batch, out, ld, s)) {
        HRAG_LAUNCH_CHECK();
        return HRAG_OK;
    }
    const int64_t tiles_m = ceil_div(rows, BM);
    if (batch > 64) {
        const int tn = (int)ceil_div(batch, 128);
        hipLaunchKernelGGL((sim_gemm_kernel<128, 2, 2>), dim3((unsigned)(tiles_m * tn)), dim3(256), 0, s,
                           emb, rows, dim, q, batch, out, ld, tn, accumulate);
    } else if (batch > 16) {
        const int tn = (int)ceil_div(batch, 64);
        hipLaunchKernelGGL((sim_gemm_kernel<64, 4, 1>), dim3((unsigned)(tiles_m * tn)), dim3(256), 0, s,
                           emb, rows, dim, q, batch, out, ld, tn, accumulate);
    } else {
        hipLaunchKernelGGL((sim_gemm_kernel<16, 4, 1>), dim3((unsigned)tiles_m), dim3(256), 0, s, emb,
                           rows, dim, q, batch, out, ld, 1, accumulate);
    }
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
