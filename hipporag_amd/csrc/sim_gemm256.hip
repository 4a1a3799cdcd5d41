// K1b -- the similarity GEMM for wide batches: 256 embedding rows x (128 | 256) queries per workgroup.
//
// Same contract as sim_gemm.hip (S[b][m] = sum_k Q[b][k] E[m][k], reference HippoRAG.py:1459 / :1496) and the
// same arithmetic: every score is the same chain of v_mfma_f32_16x16x32 in ascending k with the same
// lane <-> k mapping, so the values are BIT-IDENTICAL to sim_gemm_kernel's (the fused fact top-k's rescore
// kernel relies on that, tests/test_gpu_parity.py).  What differs is how the operands reach the matrix cores
// (every choice below was measured on the cfg-3 shape with tools/gemm_bench.hip):
//
//   * 8 wavefronts per workgroup (4 x 2), wave tile 64 x (64 | 128): half the LDS fragment bytes per MAC of
//     the 64 x 64 tiles of sim_gemm_kernel, which ran into the CU's LDS bandwidth (PMC: 42 M bank-conflict
//     cycles of 130 M LDS-active ones), and two waves per SIMD so that one wave's LDS reads / barrier waits
//     hide behind the other's MFMAs (4 waves of 128 x 128: 0.48 ms; 8 waves: 0.35 ms);
//   * global -> LDS by LDS-direct loads (global_load_lds_dwordx4, non-temporal for the embedding stream,
//     which every byte of is read once): no VGPR staging, no ds_write pass; stages of BK = 64 in an ASYMMETRIC
//     pipeline (round 4): three for the embedding rows (HBM, prefetch distance 2), two for the query tile (L2) --
//     160 KB, the whole LDS of a CU at 256 queries (symmetric deeper / finer pipelines measured slower);
//   * LDS image: [rows][8 chunks of 16 B], chunk c of row r stored at chunk c ^ (r & 7) -- no bank conflicts
//     (PMC: SQ_LDS_BANK_CONFLICT = 0).  An LDS-direct load writes wave-uniform base + lane * 16, so the
//     swizzle is applied on the GLOBAL side: lane l fetches logical chunk (l & 7) ^ ((l >> 3) & 7) of row
//     l >> 3 -- still one full 128-byte line per 8 lanes.
//
// What bounds it at B = 256: the loads alone (embedding rows from HBM + the query tile from L2, no MFMA) take
// 0.30 ms, the embedding stream alone 0.26 ms (5.2 TB/s), MFMAs + LDS reads alone 0.22 ms; the kernel runs in
// 0.35 ms = 39 % MFMA utilisation against a ceiling of 54 % set by the HBM stream (HBM-bound for
// B <= 256).
//
// Serves batch > 64 when dim % 64 == 0 (the passage GEMM and pass 1 of the fused fact top-k); everything
// else stays on sim_gemm_kernel.
#include <cstdlib>

#include "common.h"

namespace hrag {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

template <bool F16>
__device__ __forceinline__ f32x4 mfma32(const uint4 &ua, const uint4 &ub, f32x4 acc) {
    if constexpr (F16) {
        f16x8 a, b;
        __builtin_memcpy(&a, &ua, 16);
        __builtin_memcpy(&b, &ub, 16);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    } else {
        bf16x8 a, b;
        __builtin_memcpy(&a, &ua, 16);
        __builtin_memcpy(&b, &ub, 16);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
}

// LDS-direct load of 16 bytes per lane: LDS address = M0 (wave-uniform) + lane * 16.  Written as inline asm so
// that the compiler's waitcnt pass does not see an LDS write in flight: with the builtin it waits vmcnt(0)
// before the first ds_read of every k-step (it cannot prove that the stage being filled and the stage being read
// are disjoint) and the loads never overlap the MFMAs.  The kernel counts them itself: one
// `s_waitcnt vmcnt(0)` before the barrier that ends a k-step.
template <bool NT>
__device__ __forceinline__ void glds16(const void *g, uint32_t lds_addr) {
    const uint32_t uni = __builtin_amdgcn_readfirstlane(lds_addr);   // wave-uniform by construction
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(uni), "v"(g) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(uni), "v"(g) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

constexpr int BM2 = 256;   // embedding rows per workgroup (two 128-row tiles of the fused top-k)
constexpr int BK2 = 64;    // one 128-byte line of every row per step

// NJ: 16-query fragments per wave (4 -> BN = 128, 8 -> BN = 256); 8 waves: 4 (rows, 64 each) x 2 (queries)
template <int NJ, bool TILEMAX, bool F16>
__global__ __launch_bounds__(512, 1) void sim_gemm256_kernel(const uint16_t *__restrict__ emb, int64_t rows,
                                                             int32_t dim, const uint16_t *__restrict__ q,
                                                             int32_t batch, float *__restrict__ out, int64_t ld,
                                                             int32_t n_tiles_n, float *__restrict__ tmax,
                                                             float *__restrict__ tmin, int64_t emb_ld, int64_t q_ld) {
    // emb_ld / q_ld: elements between consecutive rows (>= dim).  dim < ld = a PREFIX of every row: the thresholded KNN's
    // first pass reads the hi halves of the split layout only (csrc/sim_gemm.hip launch_sim_topk_fused)
    constexpr int MI = 4, WGM = 4, WGN = 2, NW = WGM * WGN;
    constexpr int BN = WGN * NJ * 16;
    constexpr int A_BYTES = BM2 * 128, B_BYTES = BN * 128;
    constexpr int A_LOADS = BM2 / 8 / NW;   // LDS-direct loads per wave and embedding stage
    // Asymmetric pipeline (round 4; measured in tools/gemm_bench.hip: 0.341 -> 0.326 ms on the cfg 3 shape): THREE stages for
    // the embedding rows -- the HBM stream, prefetch distance 2 -- and TWO for the query tile (L2): 3 * 32 KB + 2 * 32 KB =
    // 160 KB at a 256 x 256 tile, the whole LDS of a CU.  Issue order per step: B(k + 1) then A(k + 2), so that
    // `s_waitcnt vmcnt(A_LOADS)` leaves exactly A(k + 2) in flight.  The MFMA chain of every score is untouched.
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [3][A_BYTES] then [2][B_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // XCD-aware tile order (see sim_gemm.hip): the query tiles of one row tile run back to back on one XCD
    const int64_t tile = blockIdx.x;
    const int64_t jj = tile >> 3;
    const int nt = (int)(jj % n_tiles_n);
    const int64_t mt = (jj / n_tiles_n) * 8 + (tile & 7);
    if (mt * BM2 >= rows) return;
    const int64_t m0 = mt * BM2;
    const int b0 = nt * BN;

    // LDS-direct load of one stage: 1 KB blocks of 8 rows dealt over the 8 waves (A: 4 blocks each, B: BN / 64)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (lrow & 7);          // logical 16-byte chunk this lane fetches
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto issue_a = [&](int stage, int k0) {
        const uint32_t sa = smem_base + (uint32_t)(stage * A_BYTES);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int blk = wave * A_LOADS + i;
            int64_t r = m0 + blk * 8 + lrow;
            r = r < rows ? r : rows - 1;                 // rows beyond the matrix: any valid row (never stored)
            glds16<true>(emb + (size_t)r * emb_ld + k0 + lchunk * 8, sa + (uint32_t)(blk * 1024));
        }
    };
    auto issue_b = [&](int stage, int k0) {
        const uint32_t sb = smem_base + (uint32_t)(3 * A_BYTES + stage * B_BYTES);
#pragma unroll
        for (int i = 0; i < BN / 8 / NW; ++i) {
            const int blk = wave * (BN / 8 / NW) + i;
            int r = b0 + blk * 8 + lrow;
            r = r < batch ? r : batch - 1;
            glds16<false>(q + (size_t)r * q_ld + k0 + lchunk * 8, sb + (uint32_t)(blk * 1024));
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;           // fragment row and 8-element k group of this lane
    const int nk = dim / BK2;
    issue_b(0, 0);
    issue_a(0, 0);
    if (nk > 1) issue_a(1, BK2);
    for (int kt = 0; kt < nk; ++kt) {
        // A(kt) and B(kt) have landed (the LDS-direct loads are not tracked by the compiler; they retire in order, so
        // allowing the A_LOADS youngest = A(kt + 1) to stay in flight is exact)
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // ... for every wave; and everybody is done reading the
                                                           // stages about to be refilled
        if (kt + 1 < nk) issue_b((kt + 1) & 1, (kt + 1) * BK2);   // in flight during the MFMAs below
        if (kt + 2 < nk) issue_a((kt + 2) % 3, (kt + 2) * BK2);
        const unsigned char *sa = smem + (kt % 3) * A_BYTES;
        const unsigned char *sb = smem + 3 * A_BYTES + (kt & 1) * B_BYTES;
#pragma unroll
        for (int s = 0; s < BK2 / 32; ++s) {
            const int c = s * 4 + fk;
            uint4 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * (MI * 16) + i * 16 + frow;
                a[i] = *reinterpret_cast<const uint4 *>(sa + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = wn * (NJ * 16) + j * 16 + frow;
                b[j] = *reinterpret_cast<const uint4 *>(sb + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<F16>(a[i], b[j], acc[i][j]);
        }
    }

    if constexpr (TILEMAX) {
        // waves (2 h, wn) and (2 h + 1, wn) hold the two 64-row halves of 128-row tile 2 * mt + h for their
        // NJ * 16 queries: per-wave max / min (lanes l, l + 16, l + 32, l + 48 share a query), combined through LDS
        __syncthreads();                                   // the stages are dead: reuse them
        float *red = reinterpret_cast<float *>(smem);      // [2 (max, min)][WGM][BN]
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float mx = -INFINITY, mn = INFINITY;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int64_t m = m0 + wm * (MI * 16) + i * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) {
                        mx = fmaxf(mx, acc[i][j][r]);
                        mn = fminf(mn, acc[i][j][r]);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            mn = fminf(mn, __shfl_xor(mn, 16, 64));
            mn = fminf(mn, __shfl_xor(mn, 32, 64));
            if (lane < 16) {
                const int col = wn * (NJ * 16) + j * 16 + lane;
                red[wm * BN + col] = mx;
                red[(WGM + wm) * BN + col] = mn;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += 512) {
            const int h = t / BN, col = t % BN;
            const int64_t t128 = mt * 2 + h;
            const int gb = b0 + col;
            if (t128 * 128 < rows && gb < batch) {
                tmax[(size_t)t128 * batch + gb] = fmaxf(red[(2 * h) * BN + col], red[(2 * h + 1) * BN + col]);
                tmin[(size_t)t128 * batch + gb] = fminf(red[(WGM + 2 * h) * BN + col], red[(WGM + 2 * h + 1) * BN + col]);
            }
        }
        return;
    }
    // epilogue: lane owns rows m..m+3 of query gb for every (i, j) fragment
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wm * (MI * 16) + i * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gb = b0 + wn * (NJ * 16) + j * 16 + (lane & 15);
            if (gb >= batch || m >= rows) continue;
            float *dst = out + (size_t)gb * ld + m;
            const f32x4 v = acc[i][j];
            if (m + 3 < rows && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) dst[r] = v[r];
            }
        }
    }
}

template <int NJ, bool TILEMAX, bool F16>
hrag_status launch256(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q, int32_t batch, float *out,
                      int64_t ld, float *tmax, float *tmin, hipStream_t s, int64_t row_ld) {
    constexpr int BN = NJ * 32;
    constexpr int lds_bytes = 3 * BM2 * 128 + 2 * BN * 128;   // three embedding stages, two query stages
    static_assert(lds_bytes >= 2 * 4 * BN * (int)sizeof(float), "the tile-max reduction reuses the stage memory");
    static bool configured = false;
    auto kernel = sim_gemm256_kernel<NJ, TILEMAX, F16>;
    if (!configured) {   // > 64 KB of dynamic LDS needs the opt-in once per kernel
        HRAG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        configured = true;
    }
    const int64_t tiles_m = ceil_div(rows, BM2);
    const int tn = (int)ceil_div(batch, BN);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(round_up(tiles_m, 8) * tn)), dim3(512), lds_bytes, s, emb, rows, dim, q,
                       batch, out, ld, tn, tmax, tmin, row_ld, row_ld);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace

bool sim_gemm256_serves(int64_t rows, int32_t dim, int32_t batch) {
    return batch > 64 && dim % BK2 == 0 && dim >= BK2 && rows >= BM2;
}

// tmax / tmin != nullptr: pass 1 of the fused fact top-k (per 128-row tile max / min, [ceil(rows / 128)][batch])
hrag_status launch_sim_gemm256(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q, int32_t batch,
                               float *out, int64_t ld, float *tmax, float *tmin, hipStream_t s, int32_t dtype,
                               int64_t row_ld) {
    if (row_ld <= 0) row_ld = dim;      // rows of both operands are row_ld elements apart; the product runs over the first dim
    const bool f16 = dtype == HRAG_FP16, tm = tmax != nullptr;
    const bool wide = batch > 128;
#define GO(NJ_, TM_, F_) return launch256<NJ_, TM_, F_>(emb, rows, dim, q, batch, out, ld, tmax, tmin, s, row_ld)
    if (wide) {
        if (tm) { if (f16) GO(8, true, true); else GO(8, true, false); }
        else { if (f16) GO(8, false, true); else GO(8, false, false); }
    } else {
        if (tm) { if (f16) GO(4, true, true); else GO(4, true, false); }
        else { if (f16) GO(4, false, true); else GO(4, false, false); }
    }
#undef GO
    return HRAG_OK;
}

}  // namespace hrag
