// Experiment harness (not product code), round 6: a PERSISTENT similarity GEMM with a deep embedding-row ring and
// specialised loader waves, next to the round-4 structure of csrc/sim_gemm256.hip (tools/gemm_bench.hip).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench2.hip -o tools/_bin/gemm_bench2 && tools/_bin/gemm_bench2 [a|b|c|d|e|f|all]
//
// Design under test (S[b][m] = sum_k Q[b][k] E[m][k], the same v_mfma_f32_16x16x32 chain per score as the product):
//   * one workgroup per CU (512 threads), each owning a CONTIGUOUS range of embedding rows and one 256-query tile; the
//     load pipeline runs over the flattened (tile, k-step) sequence, so a tile's fill / epilogue hides behind its
//     neighbours' loads;
//   * LDS: a ring of SA stages of 256 rows x 64 k (32 KB, XOR-swizzled 128-byte rows) for the embedding rows and TWO
//     16 KB half stages (256 queries x 32 k, 64-byte rows) for the query tile: SA = 4 fills the CU's 160 KB;
//   * waves 0-3 issue ONLY embedding loads (HBM), waves 4-7 ONLY query loads (L2): vmcnt retires in order, so a wave
//     that mixes them cannot wait for a fresh query stage without draining the deep embedding ring behind it;
//   * wave tile 128 rows x 64 queries (2 x 4 waves): a wave owns a whole 128-row top-k tile, so the fused min / max
//     epilogue needs no LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ f32x4 mfma32(const uint4 &ua, const uint4 &ub, f32x4 acc) {
    bf16x8 a, b;
    __builtin_memcpy(&a, &ua, 16);
    __builtin_memcpy(&b, &ub, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
template <bool NT>
__device__ __forceinline__ void glds16(const void *g, uint32_t lds_addr) {
    const uint32_t uni = __builtin_amdgcn_readfirstlane(lds_addr);
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(uni), "v"(g) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(uni), "v"(g) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

// MODE 0: the kernel; 1: loads only (no fragment reads, no MFMA); 2: embedding loads only; 3: no loads after the prologue
template <int SA, int MODE, bool BLOCKED, bool PRIO, bool TILEMAX>
__global__ __launch_bounds__(512) void gemm_p_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                     const uint16_t *__restrict__ q, int32_t batch,
                                                     float *__restrict__ out, int64_t ld, int32_t tn,
                                                     int64_t range_rows, float *__restrict__ tmax,
                                                     float *__restrict__ tmin) {
    constexpr int A_STAGE = 32768, B_HALF = 16384, AL = 8;   // AL: embedding loads per loader wave and stage
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [SA][A_STAGE] then [2][B_HALF]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool a_loader = wave < 4;
    // workgroup -> (row range, query tile): the query tiles of one row range run on the same XCD (block b -> XCD b % 8)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % tn;
    const int64_t range = (int64_t)(slot / tn) * 8 + xcd;
    const int64_t m_begin = range * range_rows;
    if (m_begin >= rows) return;
    const int64_t m_end = (m_begin + range_rows < rows) ? m_begin + range_rows : rows;
    const int ntile = (int)((m_end - m_begin + 255) / 256);
    const int nk = dim / 64;
    const int total = ntile * nk;
    const int b0 = nt * 256;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    // ---- loaders
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
    int ld_tile = 0, ld_kt = 0, ld_slot = 0, ld_g = 0;          // next embedding stage to issue
    auto issue_a = [&]() {
        const uint32_t sa = smem_base + (uint32_t)(ld_slot * A_STAGE);
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int blk = wave * AL + i;                      // a_loader: wave 0..3 -> blocks of 8 rows
            const uint16_t *p;
            if constexpr (BLOCKED) {
                const int64_t tg = m_begin / 256 + ld_tile;
                p = emb + ((size_t)(tg * nk + ld_kt) * 256 + blk * 8 + lrow) * 64 + lchunk * 8;
            } else {
                int64_t r = m_begin + (int64_t)ld_tile * 256 + blk * 8 + lrow;
                r = r < m_end ? r : m_end - 1;                  // beyond the range: any valid row (never stored)
                p = emb + (size_t)r * dim + ld_kt * 64 + lchunk * 8;
            }
            glds16<true>(p, sa + (uint32_t)(blk * 1024));
        }
        ++ld_g;
        if (++ld_slot == SA) ld_slot = 0;
        if (++ld_kt == nk) { ld_kt = 0; ++ld_tile; }
    };
    // query half stage h of k-step kt: rows of 64 bytes; chunk c of row r stored at c ^ f((r >> 2) & 3), f = (0, 2, 3, 1)
    const int qphys = lane & 3;
    const int qlog = qphys ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
    auto issue_b = [&](int h, int kt) {
        const uint32_t sb = smem_base + (uint32_t)(SA * A_STAGE + h * B_HALF);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int blk = (wave - 4) * 4 + i;                 // 16 queries per block
            int r = b0 + blk * 16 + (lane >> 2);
            r = r < batch ? r : batch - 1;
            glds16<false>(q + (size_t)r * dim + kt * 64 + h * 32 + qlog * 8, sb + (uint32_t)(blk * 1024));
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int bsw = (fk ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3)) << 4;
    const uint32_t a_lane = (uint32_t)((wm * 128 + frow) * 128);
    const uint32_t b_lane = (uint32_t)((wn * 64 + frow) * 64 + bsw);

    auto compute_half = [&](int a_slot, int s) {
        if constexpr (MODE == 1 || MODE == 2) return;
        const unsigned char *sa = smem + a_slot * A_STAGE + a_lane;
        const unsigned char *sb = smem + SA * A_STAGE + s * B_HALF + b_lane;
        uint4 a[8], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4 *>(sb + j * 1024);
        const int c = ((s * 4 + fk) ^ (frow & 7)) << 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const uint4 *>(sa + i * 2048 + c);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: SA - 1 embedding stages, both query halves of step 0
    if (a_loader) {
#pragma unroll
        for (int p = 0; p < SA - 1; ++p)
            if (ld_g < total) issue_a();
        if (total >= SA - 1) wait_vm<(SA - 2) * AL>(); else wait_vm<0>();
    } else if (MODE != 2) {
        issue_b(0, 0);
        issue_b(1, 0);
        wait_vm<4>();
    }
    wg_barrier();

    int tile_i = 0, kt = 0, a_slot = 0;
    for (int g = 0; g < total; ++g) {
        // A(g), B0(g) are visible; in flight: A(g + 1 .. g + SA - 2), B1(g)
        if (a_loader) {
            if (MODE != 3 && ld_g < total) issue_a();           // A(g + SA - 1) into the slot of A(g - 1)
        }
        compute_half(a_slot, 0);
        if (!a_loader) wait_vm<0>();                            // B1(g) has landed
        wg_barrier();
        const int kt_next = (kt + 1 == nk) ? 0 : kt + 1;
        if (!a_loader && MODE != 2 && MODE != 3 && g + 1 < total) issue_b(0, kt_next);   // B0(g + 1)
        compute_half(a_slot, 1);

        if (kt + 1 == nk) {
            // ---- tile epilogue: the wave owns rows m0 + wm * 128 .. + 127 of 64 queries
            const int64_t m0 = m_begin + (int64_t)tile_i * 256 + wm * 128;
            if constexpr (TILEMAX) {
                if (m0 < m_end) {
                    const int64_t t128 = m0 >> 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float mx = -INFINITY, mn = INFINITY;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int64_t m = m0 + i * 16 + 4 * fk;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (m + r < m_end) {
                                    mx = fmaxf(mx, acc[i][j][r]);
                                    mn = fminf(mn, acc[i][j][r]);
                                }
                        }
                        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                        mn = fminf(mn, __shfl_xor(mn, 16, 64));
                        mn = fminf(mn, __shfl_xor(mn, 32, 64));
                        const int gb = b0 + wn * 64 + j * 16 + lane;
                        if (lane < 16 && gb < batch) {
                            tmax[(size_t)t128 * batch + gb] = mx;
                            tmin[(size_t)t128 * batch + gb] = mn;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int64_t m = m0 + i * 16 + 4 * fk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int gb = b0 + wn * 64 + j * 16 + frow;
                        if (gb >= batch || m >= m_end) continue;
                        float *dst = out + (size_t)gb * ld + m;
                        const f32x4 v = acc[i][j];
                        if (m + 3 < m_end && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            for (int r = 0; r < 4; ++r)
                                if (m + r < m_end) dst[r] = v[r];
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        if (g + 1 < total) {
            if (a_loader) {
                // A(g + 1) must have landed; younger stages may fly: min(SA - 2, total - 2 - g) of them
                const int young = total - 2 - g;
                if (MODE == 3) wait_vm<0>();
                else if (young >= SA - 2) wait_vm<(SA - 2) * AL>();
                else if (SA > 3 && young == 1) wait_vm<AL>();
                else wait_vm<0>();
            } else {
                wait_vm<0>();                                   // B0(g + 1)
            }
        }
        wg_barrier();
        if (!a_loader && MODE != 2 && MODE != 3 && g + 1 < total) issue_b(1, kt_next);   // B1(g + 1)
        if (++a_slot == SA) a_slot = 0;
        if (++kt == nk) { kt = 0; ++tile_i; }
    }
    if constexpr (MODE != 0) {
        if (tid == 0 && total < 0) tmax[0] = acc[0][0][0];
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// X2: the round-4 product schedule (ONE barrier per k-step, embedding ring of 3 stages, TWO full query stages) +
// specialised loader waves (0-3 embeddings, 4-7 queries) + optionally persistent workgroups over contiguous row ranges
// + the 128 x 64 wave tile (register-only min / max epilogue).
template <int MODE, bool TILEMAX>
__global__ __launch_bounds__(512) void gemm_x2_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                      const uint16_t *__restrict__ q, int32_t batch,
                                                      float *__restrict__ out, int64_t ld, int32_t tn,
                                                      int64_t range_rows, float *__restrict__ tmax,
                                                      float *__restrict__ tmin) {
    constexpr int SA = 3, A_STAGE = 32768, B_STAGE = 32768, AL = 8;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [3][A_STAGE] then [2][B_STAGE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool a_loader = wave < 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % tn;
    const int64_t range = (int64_t)(slot / tn) * 8 + xcd;
    const int64_t m_begin = range * range_rows;
    if (m_begin >= rows) return;
    const int64_t m_end = (m_begin + range_rows < rows) ? m_begin + range_rows : rows;
    const int ntile = (int)((m_end - m_begin + 255) / 256);
    const int nk = dim / 64;
    const int total = ntile * nk;
    const int b0 = nt * 256;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
    int ld_tile = 0, ld_kt = 0, ld_slot = 0, ld_g = 0;
    auto issue_a = [&]() {
        const uint32_t sa = smem_base + (uint32_t)(ld_slot * A_STAGE);
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int blk = wave * AL + i;
            int64_t r = m_begin + (int64_t)ld_tile * 256 + blk * 8 + lrow;
            r = r < m_end ? r : m_end - 1;
            glds16<true>(emb + (size_t)r * dim + ld_kt * 64 + lchunk * 8, sa + (uint32_t)(blk * 1024));
        }
        ++ld_g;
        if (++ld_slot == SA) ld_slot = 0;
        if (++ld_kt == nk) { ld_kt = 0; ++ld_tile; }
    };
    auto issue_b = [&](int bslot, int kt) {
        const uint32_t sb = smem_base + (uint32_t)(SA * A_STAGE + bslot * B_STAGE);
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int blk = (wave - 4) * AL + i;
            int r = b0 + blk * 8 + lrow;
            r = r < batch ? r : batch - 1;
            glds16<false>(q + (size_t)r * dim + kt * 64 + lchunk * 8, sb + (uint32_t)(blk * 1024));
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const uint32_t a_lane = (uint32_t)((wm * 128 + frow) * 128);
    const uint32_t b_lane = (uint32_t)((wn * 64 + frow) * 128);

    if (a_loader) {
        issue_a();
        if (ld_g < total) issue_a();
        if (total >= 2) wait_vm<AL>(); else wait_vm<0>();
    } else if (MODE != 2) {
        issue_b(0, 0);
        wait_vm<0>();
    }
    wg_barrier();
    int tile_i = 0, kt = 0, a_slot = 0;
    for (int g = 0; g < total; ++g) {
        const int kt_next = (kt + 1 == nk) ? 0 : kt + 1;
        if (a_loader) { if (ld_g < total) issue_a(); }                       // A(g + 2)
        else if (MODE != 2 && g + 1 < total) issue_b((g + 1) & 1, kt_next);  // B(g + 1)
        if constexpr (MODE == 0) {
            const unsigned char *sa = smem + a_slot * A_STAGE + a_lane;
            const unsigned char *sb = smem + SA * A_STAGE + (g & 1) * B_STAGE + b_lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 a[8], b[4];
                const int c = ((s * 4 + fk) ^ (frow & 7)) << 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4 *>(sb + j * 2048 + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const uint4 *>(sa + i * 2048 + c);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
            }
        }
        if (kt + 1 == nk) {
            const int64_t m0 = m_begin + (int64_t)tile_i * 256 + wm * 128;
            if constexpr (TILEMAX) {
                if (m0 < m_end) {
                    const int64_t t128 = m0 >> 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float mx = -INFINITY, mn = INFINITY;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int64_t m = m0 + i * 16 + 4 * fk;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (m + r < m_end) {
                                    mx = fmaxf(mx, acc[i][j][r]);
                                    mn = fminf(mn, acc[i][j][r]);
                                }
                        }
                        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                        mn = fminf(mn, __shfl_xor(mn, 16, 64));
                        mn = fminf(mn, __shfl_xor(mn, 32, 64));
                        const int gb = b0 + wn * 64 + j * 16 + lane;
                        if (lane < 16 && gb < batch) {
                            tmax[(size_t)t128 * batch + gb] = mx;
                            tmin[(size_t)t128 * batch + gb] = mn;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (g + 1 < total) {
            if (a_loader) {
                if (g + 2 < total) wait_vm<AL>(); else wait_vm<0>();   // A(g + 1) landed, A(g + 2) may fly
            } else {
                wait_vm<0>();                                          // B(g + 1)
            }
        }
        wg_barrier();
        if (++a_slot == SA) a_slot = 0;
        if (++kt == nk) { kt = 0; ++tile_i; }
    }
    if constexpr (MODE != 0) {
        if (tid == 0 && total < 0) tmax[0] = acc[0][0][0];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong variant: the two waves of a SIMD run ONE PHASE apart (group 1 = waves 4-7 passes one extra barrier first):
// while one group issues the 32 MFMAs of a half k-step, the other reads its 12 fragments for the next one.  Four
// barriers per k-step.  LDS: embedding ring SA x 32 KB + a ring of four 16 KB query half stages.
//   phase 2h     group 0 READ(h)                      group 1 MFMA(h - 1)
//   phase 2h + 1 group 0 MFMA(h), issues A(g + SA-1)  group 1 READ(h), issues B(h + 3)
// Group 0 loads only embedding rows, group 1 only query rows (vmcnt is in order per wave).
template <int SA, int MODE, bool PRIO, bool TILEMAX, int GROUPING = 0, bool BBLOCKED = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                      const uint16_t *__restrict__ q, int32_t batch,
                                                      float *__restrict__ out, int64_t ld, int32_t tn,
                                                      int64_t range_rows, float *__restrict__ tmax,
                                                      float *__restrict__ tmin, unsigned long long *dbg) {
    constexpr int A_STAGE = 32768, B_HALF = 16384, AL = 8, BL = 4, SBH = 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [SA][A_STAGE] then [SBH][B_HALF]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = GROUPING == 0 ? wave >> 2 : wave & 1, wn = GROUPING == 0 ? wave & 3 : wave >> 1, wm = grp;
    const int lw = wn;                 // index of this wave among its group's four loader waves
    const bool trace = dbg != nullptr && blockIdx.x == 8 && lane == 0;
    if (trace) dbg[wave] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    auto stamp = [&](int h, int what) {
        if (trace && h >= 40 && h < 48) dbg[16 + ((h - 40) * 8 + wave) * 4 + what] = __builtin_readcyclecounter();
    };
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % tn;
    const int64_t range = (int64_t)(slot / tn) * 8 + xcd;
    const int64_t m_begin = range * range_rows;
    if (m_begin >= rows) return;
    const int64_t m_end = (m_begin + range_rows < rows) ? m_begin + range_rows : rows;
    const int ntile = (int)((m_end - m_begin + 255) / 256);
    const int nk = dim / 64;
    const int total = ntile * nk;      // k-steps
    const int H = 2 * total;           // half steps
    const int b0 = nt * 256;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int bsw = (fk ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3)) << 4;
    const uint32_t a_lane = (uint32_t)((wm * 128 + frow) * 128);
    const uint32_t b_lane = (uint32_t)((wn * 64 + frow) * 64 + bsw);
    uint4 a[8], b[4];

    auto read_frags = [&](int a_slot, int s, int b_slot) {
        if constexpr (MODE == 1 || MODE == 4) return;
        const unsigned char *sa = smem + a_slot * A_STAGE + a_lane;
        const unsigned char *sb = smem + SA * A_STAGE + b_slot * B_HALF + b_lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4 *>(sb + j * 1024);
        const int c = ((s * 4 + fk) ^ (frow & 7)) << 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const uint4 *>(sa + i * 2048 + c);
    };
    auto epilogue = [&](int tile_i) {
        const int64_t m0 = m_begin + (int64_t)tile_i * 256 + wm * 128;
        if constexpr (TILEMAX) {
            if (m0 < m_end) {
                const int64_t t128 = m0 >> 7;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int64_t m = m0 + i * 16 + 4 * fk;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (m + r < m_end) {
                                mx = fmaxf(mx, acc[i][j][r]);
                                mn = fminf(mn, acc[i][j][r]);
                            }
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    mn = fminf(mn, __shfl_xor(mn, 16, 64));
                    mn = fminf(mn, __shfl_xor(mn, 32, 64));
                    const int gb = b0 + wn * 64 + j * 16 + lane;
                    if (lane < 16 && gb < batch) {
                        tmax[(size_t)t128 * batch + gb] = mx;
                        tmin[(size_t)t128 * batch + gb] = mn;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t m = m0 + i * 16 + 4 * fk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gb = b0 + wn * 64 + j * 16 + frow;
                    if (gb >= batch || m >= m_end) continue;
                    float *dst = out + (size_t)gb * ld + m;
                    const f32x4 v = acc[i][j];
                    if (m + 3 < m_end && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (m + r < m_end) dst[r] = v[r];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    if (grp == 0) {
        // ================= group 0: embedding loader
        const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
        int ld_tile = 0, ld_kt = 0, ld_slot = 0, ld_g = 0;
        const uint16_t *ap[AL];
        auto prep_a = [&]() {          // addresses of the next stage (VALU work, done in a READ phase)
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int blk = lw * AL + i;
                int64_t r = m_begin + (int64_t)ld_tile * 256 + blk * 8 + lrow;
                r = r < m_end ? r : m_end - 1;
                ap[i] = emb + (size_t)r * dim + ld_kt * 64 + lchunk * 8;
            }
        };
        auto issue_a_one = [&](int i) {
            glds16<true>(ap[i], smem_base + (uint32_t)(ld_slot * A_STAGE + (lw * AL + i) * 1024));
        };
        auto advance_a = [&]() {
            ++ld_g;
            if (++ld_slot == SA) ld_slot = 0;
            if (++ld_kt == nk) { ld_kt = 0; ++ld_tile; }
        };
#pragma unroll
        for (int p = 0; p < SA - 1; ++p) {
            if (ld_g < total && MODE != 4) {
                prep_a();
#pragma unroll
                for (int i = 0; i < AL; ++i) issue_a_one(i);
                advance_a();
            }
        }
        if (total >= SA - 1) wait_vm<(SA - 2) * AL>(); else wait_vm<0>();
        wg_barrier();                                            // E_pre
        int tile_i = 0, kt = 0, a_slot = 0, b_slot = 0;
        for (int h = 0; h < H; ++h) {
            const int s = h & 1;
            // ---- READ(h)
            stamp(h, 0);
            read_frags(a_slot, s, b_slot);
            const bool issue = (s == 0) && (MODE != 3) && (MODE != 4) && (ld_g < total);    // A(g + SA - 1) goes out in MFMA(g, s = 0)
            if (issue) prep_a();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(h, 1);
            wg_barrier();                                        // E_2h
            stamp(h, 2);
            // ---- MFMA(h)
            if constexpr (MODE != 1 && MODE != 4) {
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (issue) issue_a_one(i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
                }
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            } else {
                if (issue) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) issue_a_one(i);
                }
            }
            if (issue) advance_a();
            stamp(h, 3);
            if (s == 1) {
                if (kt + 1 == nk) epilogue(tile_i);
                // A(g + 1) must be visible after this barrier; younger stages may fly
                const int g = h >> 1;
                if (g + 1 < total) {
                    const int young = total - 2 - g;
                    if (MODE == 3) wait_vm<0>();
                    else if (young >= SA - 2) wait_vm<(SA - 2) * AL>();
                    else if (SA > 3 && young == 1) wait_vm<AL>();
                    else wait_vm<0>();
                }
                if (++a_slot == SA) a_slot = 0;
                if (++kt == nk) { kt = 0; ++tile_i; }
            }
            if (++b_slot == SBH) b_slot = 0;
            wg_barrier();                                        // E_2h+1
        }
        wg_barrier();                                            // pairs with group 1's last barrier
    } else {
        // ================= group 1: query loader, one phase behind
        const int qlog = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
        int lb_h = 0, lb_kt = 0, lb_slot = 0;                    // next query half stage to issue
        auto issue_b = [&]() {
            const uint32_t sb = smem_base + (uint32_t)(SA * A_STAGE + lb_slot * B_HALF);
#pragma unroll
            for (int i = 0; i < BL; ++i) {
                const int blk = lw * BL + i;
                int r = b0 + blk * 16 + (lane >> 2);
                r = r < batch ? r : batch - 1;
                const uint16_t *src = BBLOCKED
                    ? q + ((size_t)(nt * nk + lb_kt) * 256 + (r - b0)) * 64 + (lb_h & 1) * 32 + qlog * 8   // [tile][k-step][256][64]
                    : q + (size_t)r * dim + lb_kt * 64 + (lb_h & 1) * 32 + qlog * 8;
                glds16<false>(src, sb + (uint32_t)(blk * 1024));
            }
            if (lb_h & 1) { if (++lb_kt == nk) lb_kt = 0; }
            ++lb_h;
            if (++lb_slot == SBH) lb_slot = 0;
        };
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (lb_h < H) issue_b();
        if (H >= 3) wait_vm<2 * BL>(); else wait_vm<0>();
        wg_barrier();                                            // E_pre
        wg_barrier();                                            // E_0: the stagger
        int tile_i = 0, kt = 0, a_slot = 0, b_slot = 0;
        for (int h = 0; h < H; ++h) {
            const int s = h & 1;
            // ---- READ(h)
            stamp(h, 0);
            if (MODE != 3 && lb_h < H) issue_b();                // B(h + 3)
            read_frags(a_slot, s, b_slot);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(h, 1);
            if (h + 1 < H) {                                     // B(h + 1) visible after this barrier
                const int young = H - 2 - h;
                if (MODE == 3) wait_vm<0>();
                else if (young >= 2) wait_vm<2 * BL>();
                else if (young == 1) wait_vm<BL>();
                else wait_vm<0>();
            }
            wg_barrier();                                        // E_2h+1
            stamp(h, 2);
            // ---- MFMA(h)
            if constexpr (MODE != 1 && MODE != 4) {
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            }
            stamp(h, 3);
            if (s == 1) {
                if (kt + 1 == nk) epilogue(tile_i);
                if (++a_slot == SA) a_slot = 0;
                if (++kt == nk) { kt = 0; ++tile_i; }
            }
            if (++b_slot == SBH) b_slot = 0;
            wg_barrier();                                        // E_2h+2
        }
    }
    if constexpr (MODE != 0) {
        if (tid == 0 && total < 0) tmax[0] = acc[0][0][0] + __uint_as_float(a[0].x);
    }
}

static int n_cus() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    return p.multiProcessorCount;
}

struct Times { float best, med; };
template <typename F>
Times time_it(F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        const int n = 10;
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / n);
    }
    std::sort(t.begin(), t.end());
    return {t[0], t[2]};
}

// Asymmetric pipeline: THREE stages for the embedding rows (HBM: the long-latency stream, prefetch distance 2) and TWO
// for the query tile (L2), BK = 64: 3 * 32 KB + 2 * 32 KB = 160 KB at a 256 x 256 tile -- the whole LDS of a CU.
// Issue order per step: B(k + 1) then A(k + 2), so that `vmcnt(loads of one A stage)` leaves exactly A(k + 2) in flight.
template <int MI, int NJ, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void gemm_asym_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                                      const uint16_t *__restrict__ q, int32_t batch,
                                                                      float *__restrict__ tmax) {
    constexpr int NW = WGM * WGN, BK = 64;
    constexpr int BM = WGM * MI * 16, BN = WGN * NJ * 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int A_LOADS = BM / 8 / NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int64_t mt = blockIdx.x;
    if (mt * BM >= rows) return;
    const int64_t m0 = mt * BM;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto issue_a = [&](int stage, int k0) {
        const uint32_t sa = smem_base + (uint32_t)(stage * A_BYTES);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int blk = wave * A_LOADS + i;
            int64_t r = m0 + blk * 8 + lrow;
            r = r < rows ? r : rows - 1;
            glds16<true>(emb + (size_t)r * dim + k0 + lchunk * 8, sa + (uint32_t)(blk * 1024));
        }
    };
    auto issue_b = [&](int stage, int k0) {
        const uint32_t sb = smem_base + (uint32_t)(3 * A_BYTES + stage * B_BYTES);
#pragma unroll
        for (int i = 0; i < BN / 8 / NW; ++i) {
            const int blk = wave * (BN / 8 / NW) + i;
            int r = blk * 8 + lrow;
            r = r < batch ? r : batch - 1;
            glds16<false>(q + (size_t)r * dim + k0 + lchunk * 8, sb + (uint32_t)(blk * 1024));
        }
    };
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int nk = dim / BK;
    issue_b(0, 0);
    issue_a(0, 0);
    if (nk > 1) issue_a(1, BK);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LOADS) : "memory");   // A(kt + 1) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue_b((kt + 1) & 1, (kt + 1) * BK);
        if (kt + 2 < nk) issue_a((kt + 2) % 3, (kt + 2) * BK);
        const unsigned char *sa = smem + (kt % 3) * A_BYTES;
        const unsigned char *sb = smem + 3 * A_BYTES + (kt & 1) * B_BYTES;
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
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
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int gb = wn * (NJ * 16) + j * 16 + lane;
        if (lane < 16 && gb < batch) tmax[((size_t)mt * WGM + wm) * batch + gb] = mx;
    }
}

template <int MI, int NJ, int WGM, int WGN>
void run_asym(const char *name, const uint16_t *emb, int64_t rows, int dim, const uint16_t *q, int batch, float *tmax) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NJ * 16;
    const int lds = 3 * BM * 128 + 2 * BN * 128;
    auto k = gemm_asym_kernel<MI, NJ, WGM, WGN>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) { printf("%-44s %d KB of LDS refused: %s\n", name, lds / 1024, hipGetErrorString(err)); (void)hipGetLastError(); return; }
    const unsigned grid = (unsigned)((rows + BM - 1) / BM);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WGM * WGN * 64), lds, 0, emb, rows, dim, q, batch, tmax);
    err = hipGetLastError();
    if (err != hipSuccess) { printf("%-44s launch failed: %s\n", name, hipGetErrorString(err)); return; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WGM * WGN * 64), lds, 0, emb, rows, dim, q, batch, tmax);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
    const double flop = 2.0 * rows * batch * dim;
    printf("%-44s tile %3dx%3d lds %3d KB  %.3f ms  %.0f TFLOP/s (%.1f %% of 2.5 PF)  %.2f TB/s of A\n", name, BM, BN, lds / 1024,
           ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0, (double)rows * dim * 2 / ms / 1e9);
}

template <int MI, int NJ, int WGM, int WGN, int ISSUE>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void gemm_asym2_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                                      const uint16_t *__restrict__ q, int32_t batch,
                                                                      float *__restrict__ tmax) {
    constexpr int NW = WGM * WGN, BK = 64;
    constexpr int BM = WGM * MI * 16, BN = WGN * NJ * 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int A_LOADS = BM / 8 / NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int64_t mt = blockIdx.x;
    if (mt * BM >= rows) return;
    const int64_t m0 = mt * BM;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto issue_a = [&](int stage, int k0) {
        const uint32_t sa = smem_base + (uint32_t)(stage * A_BYTES);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int blk = wave * A_LOADS + i;
            int64_t r = m0 + blk * 8 + lrow;
            r = r < rows ? r : rows - 1;
            glds16<true>(emb + (size_t)r * dim + k0 + lchunk * 8, sa + (uint32_t)(blk * 1024));
        }
    };
    auto issue_b = [&](int stage, int k0) {
        const uint32_t sb = smem_base + (uint32_t)(3 * A_BYTES + stage * B_BYTES);
#pragma unroll
        for (int i = 0; i < BN / 8 / NW; ++i) {
            const int blk = wave * (BN / 8 / NW) + i;
            int r = blk * 8 + lrow;
            r = r < batch ? r : batch - 1;
            glds16<false>(q + (size_t)r * dim + k0 + lchunk * 8, sb + (uint32_t)(blk * 1024));
        }
    };
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int nk = dim / BK;
    issue_b(0, 0);
    issue_a(0, 0);
    if (nk > 1) issue_a(1, BK);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LOADS) : "memory");   // A(kt + 1) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto issue_next = [&]() {
            if (kt + 1 < nk) issue_b((kt + 1) & 1, (kt + 1) * BK);
            if (kt + 2 < nk) issue_a((kt + 2) % 3, (kt + 2) * BK);
        };
        if constexpr (ISSUE == 0) issue_next();
        const unsigned char *sa = smem + (kt % 3) * A_BYTES;
        const unsigned char *sb = smem + 3 * A_BYTES + (kt & 1) * B_BYTES;
        uint4 a[MI], b[NJ];
        auto frags = [&](int s) {
            const int c = s * 4 + fk;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = wn * (NJ * 16) + j * 16 + frow;
                b[j] = *reinterpret_cast<const uint4 *>(sb + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * (MI * 16) + i * 16 + frow;
                a[i] = *reinterpret_cast<const uint4 *>(sa + r * 128 + ((c ^ (r & 7)) << 4));
            }
        };
        auto mfmas = [&]() {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        };
        frags(0);
        if constexpr (ISSUE == 1) issue_next();       // behind the first fragment reads: issue overlaps the LDS latency
        mfmas();
        if constexpr (ISSUE == 2) issue_next();       // between the halves
        frags(1);
        if constexpr (ISSUE == 3) issue_next();       // behind the second half's reads
        mfmas();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int gb = wn * (NJ * 16) + j * 16 + lane;
        if (lane < 16 && gb < batch) tmax[((size_t)mt * WGM + wm) * batch + gb] = mx;
    }
}


template <int MI, int NJ, int WGM, int WGN, int ISSUE>
void run_asym2(const char *name, const uint16_t *emb, int64_t rows, int dim, const uint16_t *q, int batch, float *tmax) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NJ * 16;
    const int lds = 3 * BM * 128 + 2 * BN * 128;
    auto k = gemm_asym2_kernel<MI, NJ, WGM, WGN, ISSUE>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) { printf("%-44s %d KB of LDS refused: %s\n", name, lds / 1024, hipGetErrorString(err)); (void)hipGetLastError(); return; }
    const unsigned grid = (unsigned)((rows + BM - 1) / BM);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WGM * WGN * 64), lds, 0, emb, rows, dim, q, batch, tmax);
    err = hipGetLastError();
    if (err != hipSuccess) { printf("%-44s launch failed: %s\n", name, hipGetErrorString(err)); return; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WGM * WGN * 64), lds, 0, emb, rows, dim, q, batch, tmax);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
    const double flop = 2.0 * rows * batch * dim;
    printf("%-44s tile %3dx%3d lds %3d KB  %.3f ms  %.0f TFLOP/s (%.1f %% of 2.5 PF)  %.2f TB/s of A\n", name, BM, BN, lds / 1024,
           ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0, (double)rows * dim * 2 / ms / 1e9);
}


template <int MODE, bool TILEMAX>
void run_x2(const char *name, bool persistent, const uint16_t *emb, int64_t rows, int dim, const uint16_t *q, int batch,
            float *out, float *tmax, float *tmin) {
    const int tn = (batch + 255) / 256;
    const int lds = 3 * 32768 + 2 * 32768;
    auto k = gemm_x2_kernel<MODE, TILEMAX>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int64_t nranges, range_rows;
    if (persistent) {
        int g = n_cus() / (8 * tn) * (8 * tn);
        nranges = g / tn;
        range_rows = ((rows + nranges - 1) / nranges + 127) / 128 * 128;
    } else {
        range_rows = 256;
        nranges = ((rows + 255) / 256 + 7) / 8 * 8;
    }
    const unsigned grid = (unsigned)(nranges * tn);
    auto launch = [&]() {
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, emb, rows, dim, q, batch, out, (int64_t)rows, tn, range_rows, tmax, tmin);
    };
    const Times t = time_it(launch);
    const double flop = 2.0 * rows * batch * dim;
    printf("%-58s grid %5u range %6lld  best %.3f med %.3f ms  %5.0f TFLOP/s (%4.1f %%)  %.2f TB/s of A\n", name, grid,
           (long long)range_rows, t.best, t.med, flop / t.med / 1e9, flop / t.med / 1e9 / 25.0, (double)rows * dim * 2 / t.med / 1e9);
    fflush(stdout);
}

// persistent = true: one workgroup per CU and contiguous row ranges; false: one 256-row tile per workgroup
template <int SA, int MODE, bool BLOCKED, bool PRIO, bool TILEMAX>
void run_p(const char *name, bool persistent, const uint16_t *emb, int64_t rows, int dim, const uint16_t *q, int batch,
           float *out, float *tmax, float *tmin, int wgs_per_cu_x100 = 100) {
    const int tn = (batch + 255) / 256;
    const int lds = SA * 32768 + 2 * 16384;
    auto k = gemm_p_kernel<SA, MODE, BLOCKED, PRIO, TILEMAX>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int64_t nranges, range_rows;
    if (persistent) {
        int g = n_cus() * wgs_per_cu_x100 / 100;
        g = g / (8 * tn) * (8 * tn);
        nranges = g / tn;
        const int64_t unit = BLOCKED ? 256 : 128;
        range_rows = ((rows + nranges - 1) / nranges + unit - 1) / unit * unit;
    } else {
        range_rows = 256;
        nranges = ((rows + 255) / 256 + 7) / 8 * 8;
    }
    const unsigned grid = (unsigned)(nranges * tn);
    auto launch = [&]() {
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, emb, rows, dim, q, batch, out, (int64_t)rows, tn, range_rows,
                           tmax, tmin);
    };
    const Times t = time_it(launch);
    const double flop = 2.0 * rows * batch * dim;
    printf("%-58s grid %5u range %6lld  best %.3f med %.3f ms  %5.0f TFLOP/s (%4.1f %%)  %.2f TB/s of A\n", name, grid,
           (long long)range_rows, t.best, t.med, flop / t.med / 1e9, flop / t.med / 1e9 / 25.0,
           (double)rows * dim * 2 / t.med / 1e9);
    fflush(stdout);
}

// persistent = true: one workgroup per CU and contiguous row ranges; false: one 256-row tile per workgroup
template <int SA, int MODE, bool PRIO, bool TILEMAX, int GROUPING = 0, bool BBLOCKED = false>
void run_pp(const char *name, bool persistent, const uint16_t *emb, int64_t rows, int dim, const uint16_t *q, int batch,
           float *out, float *tmax, float *tmin, unsigned long long *dbg = nullptr) {
    const int wgs_per_cu_x100 = 100;
    const int tn = (batch + 255) / 256;
    const int lds = SA * 32768 + 4 * 16384;
    auto k = gemm_pp_kernel<SA, MODE, PRIO, TILEMAX, GROUPING, BBLOCKED>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int64_t nranges, range_rows;
    if (persistent) {
        int g = n_cus() * wgs_per_cu_x100 / 100;
        g = g / (8 * tn) * (8 * tn);
        nranges = g / tn;
        const int64_t unit = 128;
        range_rows = ((rows + nranges - 1) / nranges + unit - 1) / unit * unit;
    } else {
        range_rows = 256;
        nranges = ((rows + 255) / 256 + 7) / 8 * 8;
    }
    const unsigned grid = (unsigned)(nranges * tn);
    auto launch = [&]() {
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, emb, rows, dim, q, batch, out, (int64_t)rows, tn, range_rows,
                           tmax, tmin, dbg);
    };
    const Times t = time_it(launch);
    const double flop = 2.0 * rows * batch * dim;
    printf("%-58s grid %5u range %6lld  best %.3f med %.3f ms  %5.0f TFLOP/s (%4.1f %%)  %.2f TB/s of A\n", name, grid,
           (long long)range_rows, t.best, t.med, flop / t.med / 1e9, flop / t.med / 1e9 / 25.0,
           (double)rows * dim * 2 / t.med / 1e9);
    fflush(stdout);
}

static float bf16_to_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv) {
    // usage: gemm_bench2 [a|b|c|d|e|f|all]   (the experiments of docs/experiments/README.md, round 6; logs: profiles/r06_gemm_exp_*)
    const std::string exp = argc > 1 ? argv[1] : "all";
    auto want = [&](const char *e) { return exp == "all" || exp == e; };
    const int64_t rows = 875000;
    const int dim = 768;
    const int max_batch = 1024;
    uint16_t *emb, *q; float *tmax, *tmin, *out;
    const int64_t rows_alloc = rows + 256 * 600;   // the blocked layout addresses whole tiles per range
    CK(hipMalloc(&emb, (size_t)rows_alloc * dim * 2)); CK(hipMalloc(&q, (size_t)max_batch * dim * 2));
    CK(hipMalloc(&tmax, (size_t)(rows / 128 + 8) * max_batch * 4));
    CK(hipMalloc(&tmin, (size_t)(rows / 128 + 8) * max_batch * 4));
    CK(hipMalloc(&out, (size_t)256 * 131072 * 4));
    std::vector<uint16_t> h((size_t)rows_alloc * dim);
    uint32_t s = 1;
    for (auto &v : h) {   // uniform [-1, 1) truncated to bf16: full-range operands (sign and exponent vary)
        s = s * 1664525u + 1013904223u;
        const float f = (float)(int32_t)s * (1.0f / 2147483648.0f);
        uint32_t u; memcpy(&u, &f, 4);
        v = (uint16_t)(u >> 16);
    }
    CK(hipMemcpy(emb, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<uint16_t> hq((size_t)max_batch * dim);
    for (auto &v : hq) {
        s = s * 1664525u + 1013904223u;
        const float f = (float)(int32_t)s * (1.0f / 2147483648.0f);
        uint32_t u; memcpy(&u, &f, 4);
        v = (uint16_t)(u >> 16);
    }
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    printf("CUs: %d, rows %lld, dim %d\n", n_cus(), (long long)rows, dim);

    // ---- correctness of the candidate (fp64 host reference on sampled 128-row tiles)
    auto check = [&](int batch, const char *what) {
        std::vector<float> hx((size_t)(rows / 128 + 8) * batch), hn(hx.size());
        CK(hipMemcpy(hx.data(), tmax, hx.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hn.data(), tmin, hn.size() * 4, hipMemcpyDeviceToHost));
        const int64_t n128 = (rows + 127) / 128;
        double worst = 0;
        uint32_t t = 12345;
        for (int it = 0; it < 48; ++it) {
            t = t * 1664525u + 1013904223u;
            int64_t tile = it < 4 ? (it < 2 ? it : n128 - 1 - (it - 2)) : (int64_t)(t >> 8) % n128;
            t = t * 1664525u + 1013904223u;
            const int b = (int)((t >> 8) % batch);
            double mx = -1e30, mn = 1e30;
            for (int64_t m = tile * 128; m < tile * 128 + 128 && m < rows; ++m) {
                double acc = 0;
                for (int k = 0; k < dim; ++k) acc += (double)bf16_to_f(h[(size_t)m * dim + k]) * bf16_to_f(hq[(size_t)b * dim + k]);
                mx = acc > mx ? acc : mx; mn = acc < mn ? acc : mn;
            }
            worst = fmax(worst, fabs(hx[(size_t)tile * batch + b] - mx));
            worst = fmax(worst, fabs(hn[(size_t)tile * batch + b] - mn));
        }
        printf("check %-40s worst |tile max/min - fp64| = %.3e %s\n", what, worst, worst < 2e-3 ? "OK" : "MISMATCH");
    };
    auto clear = [&](int batch) {
        CK(hipMemset(tmax, 0, (size_t)(rows / 128 + 8) * batch * 4)); CK(hipMemset(tmin, 0, (size_t)(rows / 128 + 8) * batch * 4));
    };
    unsigned long long *dbg;
    CK(hipMalloc(&dbg, 4096 * 8));
    auto dump = [&](const char *what) {
        std::vector<unsigned long long> d(4096);
        CK(hipMemcpy(d.data(), dbg, 4096 * 8, hipMemcpyDeviceToHost));
        printf("trace %s\n  HW_ID per wave (simd = bits 5:4):", what);
        for (int w = 0; w < 8; ++w) printf(" w%d:simd%llu", w, (d[w] >> 4) & 3);
        printf("\n");
        for (int w = 0; w < 8; w += 4) {
            printf("  wave %d: per half step [read-start, reads-done, barrier-passed, mfma-issued] deltas:\n   ", w);
            unsigned long long prev = d[16 + (0 * 8 + w) * 4 + 0];
            for (int h = 0; h < 8; ++h) {
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long v = d[16 + (h * 8 + w) * 4 + k];
                    printf(" %5lld", (long long)(v - prev));
                    prev = v;
                }
                printf(" |");
            }
            printf("\n");
        }
    };

    uint16_t *qb;      // the query tile re-laid [tile][k-step][256 queries][64 k]: every 32 KB stage contiguous
    CK(hipMalloc(&qb, (size_t)max_batch * dim * 2));
    {
        std::vector<uint16_t> hb((size_t)max_batch * dim);
        const int nk = dim / 64;
        for (int b = 0; b < max_batch; ++b)
            for (int k = 0; k < dim; ++k)
                hb[((size_t)((b / 256) * nk + k / 64) * 256 + (b % 256)) * 64 + (k % 64)] = hq[(size_t)b * dim + k];
        CK(hipMemcpy(qb, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    }
    float *tmax_old;
    CK(hipMalloc(&tmax_old, (size_t)(rows / 16 + 64) * 256 * 4));

    if (want("a")) {   // lockstep, deep embedding ring + half query stages, loader-specialised waves
        printf("==== (a) lockstep variant: ring of SA stages + two half query stages, waves 0-3 / 4-7 load A / B\n");
        clear(256);
        run_p<4, 0, false, false, true>("[check] SA4 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
        check(256, "SA4 persistent B=256");
        clear(1024);
        run_p<3, 0, false, false, true>("[check] SA3 persistent B=1000", true, emb, rows, dim, q, 1000, out, tmax, tmin);
        check(1000, "SA3 persistent B=1000");
        for (int rep = 0; rep < 2; ++rep) {
            printf("---- round %d, B = 256\n", rep);
            run_p<4, 0, false, false, true>("SA4 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 0, false, true, true>("SA4 persistent setprio", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 0, false, false, true>("SA4 one tile per workgroup", false, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<3, 0, false, false, true>("SA3 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 0, true, false, true>("SA4 persistent, blocked A layout", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 1, false, false, true>("SA4 persistent, loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 2, false, false, true>("SA4 persistent, A loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<3, 2, false, false, true>("SA3 persistent, A loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 2, true, false, true>("SA4 persistent, A loads only, blocked", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 2, false, false, true>("SA4 one tile per WG, A loads only", false, emb, rows, dim, q, 256, out, tmax, tmin);
            run_p<4, 3, false, false, true>("SA4 persistent, compute only", true, emb, rows, dim, q, 256, out, tmax, tmin);
        }
        printf("---- B = 1024 (MFMA-bound shape)\n");
        run_p<4, 0, false, false, true>("SA4 persistent B=1024", true, emb, rows, dim, q, 1024, out, tmax, tmin);
        run_p<4, 0, false, true, true>("SA4 persistent setprio B=1024", true, emb, rows, dim, q, 1024, out, tmax, tmin);
        run_p<4, 0, false, false, true>("SA4 one tile per workgroup B=1024", false, emb, rows, dim, q, 1024, out, tmax, tmin);
        printf("---- passage shape (rows 125000, stores S)\n");
        run_p<4, 0, false, false, false>("SA4 persistent rows=125000 store", true, emb, 125000, dim, q, 256, out, tmax, tmin);
        run_p<4, 0, false, false, false>("SA4 tile per WG rows=125000 store", false, emb, 125000, dim, q, 256, out, tmax, tmin);
    }
    if (want("b")) {   // staggered ping-pong wave groups
        printf("==== (b) ping-pong variant: waves 4-7 one phase behind waves 0-3\n");
        clear(256);
        run_pp<3, 0, false, true>("[check] PP SA3 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
        check(256, "PP SA3 persistent B=256");
        for (int rep = 0; rep < 2; ++rep) {
            printf("---- round %d, B = 256\n", rep);
            run_pp<3, 0, false, true>("PP SA3 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 0, true, true>("PP SA3 persistent setprio", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 0, false, true>("PP SA3 one tile per workgroup", false, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 1, false, true>("PP SA3 persistent, loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 3, false, true>("PP SA3 persistent, compute only", true, emb, rows, dim, q, 256, out, tmax, tmin);
        }
        printf("---- B = 1024\n");
        run_pp<3, 0, false, true>("PP SA3 persistent B=1024", true, emb, rows, dim, q, 1024, out, tmax, tmin);
        run_pp<3, 3, false, true>("PP SA3 persistent compute only B=1024", true, emb, rows, dim, q, 1024, out, tmax, tmin);
    }
    if (want("c")) {   // in-kernel timing trace of the ping-pong variant, wave -> SIMD map, alternate grouping
        printf("==== (c) ping-pong variant: s_memtime trace and wave grouping\n");
        CK(hipMemset(dbg, 0, 4096 * 8));
        run_pp<3, 3, false, true, 0>("[trace] PP compute only, groups = wave >> 2", true, emb, rows, dim, q, 256, out, tmax, tmin, dbg);
        dump("compute only, wave >> 2");
        CK(hipMemset(dbg, 0, 4096 * 8));
        run_pp<3, 0, false, true, 0>("[trace] PP full, groups = wave >> 2", true, emb, rows, dim, q, 256, out, tmax, tmin, dbg);
        dump("full, wave >> 2");
        run_pp<3, 0, false, true, 1>("PP SA3 persistent, groups = wave & 1", true, emb, rows, dim, q, 256, out, tmax, tmin);
        run_pp<3, 3, false, true, 1>("PP SA3 persistent, compute only, groups = wave & 1", true, emb, rows, dim, q, 256, out, tmax, tmin);
    }
    if (want("d")) {   // blocked query layout, query loads alone
        printf("==== (d) query tile stored [k-step][query][64]\n");
        clear(256);
        run_pp<3, 0, false, true, 0, true>("[check] PP SA3 persistent, blocked queries", true, emb, rows, dim, qb, 256, out, tmax, tmin);
        check(256, "PP SA3 blocked queries B=256");
        for (int rep = 0; rep < 2; ++rep) {
            run_pp<3, 0, false, true, 0, false>("PP SA3 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 0, false, true, 0, true>("PP SA3 persistent, blocked queries", true, emb, rows, dim, qb, 256, out, tmax, tmin);
            run_pp<3, 1, false, true, 0, false>("PP SA3 loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 1, false, true, 0, true>("PP SA3 loads only, blocked queries", true, emb, rows, dim, qb, 256, out, tmax, tmin);
            run_pp<3, 4, false, true, 0, false>("PP SA3 query loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_pp<3, 4, false, true, 0, true>("PP SA3 query loads only, blocked queries", true, emb, rows, dim, qb, 256, out, tmax, tmin);
        }
    }
    if (want("e")) {   // X2 = the round-4 schedule + loader split + persistence, against the round-4 structure on the same data
        printf("==== (e) X2 against the round-4 structure\n");
        clear(256);
        run_x2<0, true>("[check] X2 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
        check(256, "X2 persistent B=256");
        for (int rep = 0; rep < 3; ++rep) {
            printf("---- round %d, B = 256\n", rep);
            run_asym<4, 8, 4, 2>("round-4 product structure (8 waves 64x128, A3 + B2)", emb, rows, dim, q, 256, tmax_old);
            run_x2<0, true>("X2 persistent", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_x2<0, true>("X2 one tile per workgroup", false, emb, rows, dim, q, 256, out, tmax, tmin);
            run_x2<1, true>("X2 persistent, loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
            run_x2<2, true>("X2 persistent, A loads only", true, emb, rows, dim, q, 256, out, tmax, tmin);
        }
        printf("---- B = 1024\n");
        run_x2<0, true>("X2 persistent B=1024", true, emb, rows, dim, q, 1024, out, tmax, tmin);
        run_x2<0, true>("X2 one tile per workgroup B=1024", false, emb, rows, dim, q, 1024, out, tmax, tmin);
    }
    if (want("f")) {   // where the next stages' loads are issued inside a k-step of the round-4 structure
        printf("==== (f) load-issue placement inside a k-step (round-4 structure)\n");
        for (int rep = 0; rep < 4; ++rep) {
            printf("---- round %d, B = 256\n", rep);
            run_asym<4, 8, 4, 2>("round-4 product structure", emb, rows, dim, q, 256, tmax_old);
            run_asym2<4, 8, 4, 2, 0>("same, restructured source, issue first (as product)", emb, rows, dim, q, 256, tmax_old);
            run_asym2<4, 8, 4, 2, 1>("issue behind the first fragment reads", emb, rows, dim, q, 256, tmax_old);
            run_asym2<4, 8, 4, 2, 2>("issue between the halves", emb, rows, dim, q, 256, tmax_old);
            run_asym2<4, 8, 4, 2, 3>("issue behind the second half's reads", emb, rows, dim, q, 256, tmax_old);
        }
    }
    return 0;
}
