// Experiment harness (not product code) for the wide-batch similarity GEMM of csrc/sim_gemm256.hip: the same
// kernel structure with compile-time knobs to find out what bounds it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o tools/_bin/gemm_bench && tools/_bin/gemm_bench
// Prints the launch time of every variant at M = 875 000, N = 256, K = 768 (bf16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
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
__device__ __forceinline__ void glds16(const void *g, uint32_t lds_addr) {
    const uint32_t uni = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(uni), "v"(g) : "memory");
}
__device__ __forceinline__ void glds16_nt(const void *g, uint32_t lds_addr) {
    const uint32_t uni = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(uni), "v"(g) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// MI x NJ fragments per wave (wave tile 16 MI x 16 NJ), WGM x WGN waves; STAGES LDS stages of BK = 64
// knobs: LOAD (0: only the first stage is ever loaded), MFMA (0: skip), LDSR (0: fragments are not re-read)
template <int MI, int NJ, int WGM, int WGN, int STAGES, int LOAD, int MFMA, int LDSR, int BK = 64>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void gemm_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                                 const uint16_t *__restrict__ q, int32_t batch,
                                                                 float *__restrict__ tmax) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * MI * 16, BN = WGN * NJ * 16;
    constexpr int RB = BK * 2;            // bytes of a row per stage
    constexpr int CPR = RB / 16;          // 16-byte chunks per row
    constexpr int RPB = 1024 / RB;        // rows per 1 KB LDS-direct block
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int64_t mt = blockIdx.x;
    if (mt * BM >= rows) return;
    const int64_t m0 = mt * BM;
    const int lrow = lane / CPR, lchunk = (lane % CPR) ^ (lrow % CPR);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto issue = [&](int stage, int k0) {
        const uint32_t sa = smem_base + (uint32_t)(stage * STAGE);
#pragma unroll
        for (int i = 0; i < BM / RPB / NW; ++i) {
            const int blk = wave * (BM / RPB / NW) + i;
            int64_t r = m0 + blk * RPB + lrow;
            r = r < rows ? r : rows - 1;
            if (LOAD == 3) glds16_nt(emb + (size_t)r * dim + k0 + lchunk * 8, sa + (uint32_t)(blk * 1024));
            else glds16(emb + (size_t)r * dim + k0 + lchunk * 8, sa + (uint32_t)(blk * 1024));
        }
        if (LOAD == 2) return;
#pragma unroll
        for (int i = 0; i < BN / RPB / NW; ++i) {
            const int blk = wave * (BN / RPB / NW) + i;
            int r = blk * RPB + lrow;
            r = r < batch ? r : batch - 1;
            glds16(q + (size_t)r * dim + k0 + lchunk * 8, sa + (uint32_t)(A_BYTES + blk * 1024));
        }
    };
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int nk = dim / BK;
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
        if (p < nk) issue(p, p * BK);
    uint4 a[MI], b[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = make_uint4(lane, i, 3, 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = make_uint4(lane, j, 5, 6);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt must have landed: at most STAGES - 2 younger stages may still be in flight
        if constexpr (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
            if (kt + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * ((BM + (LOAD == 2 ? 0 : BN)) / RPB / NW)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (LOAD && kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * BK);
        const unsigned char *sa = smem + (kt % STAGES) * STAGE;
        const unsigned char *sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            const int c = s * 4 + fk;
            if (LDSR || kt == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int r = wm * (MI * 16) + i * 16 + frow;
                    a[i] = *reinterpret_cast<const uint4 *>(sa + r * RB + ((c ^ (r % CPR)) << 4));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int r = wn * (NJ * 16) + j * 16 + frow;
                    b[j] = *reinterpret_cast<const uint4 *>(sb + r * RB + ((c ^ (r % CPR)) << 4));
                }
            }
            if (MFMA) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][0][0] += __uint_as_float(a[i].x);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[0][j][1] += __uint_as_float(b[j].y);
            }
        }
    }
    // tile max over the wave's rows per query (the TILEMAX epilogue, simplified)
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
            glds16_nt(emb + (size_t)r * dim + k0 + lchunk * 8, sa + (uint32_t)(blk * 1024));
        }
    };
    auto issue_b = [&](int stage, int k0) {
        const uint32_t sb = smem_base + (uint32_t)(3 * A_BYTES + stage * B_BYTES);
#pragma unroll
        for (int i = 0; i < BN / 8 / NW; ++i) {
            const int blk = wave * (BN / 8 / NW) + i;
            int r = blk * 8 + lrow;
            r = r < batch ? r : batch - 1;
            glds16(q + (size_t)r * dim + k0 + lchunk * 8, sb + (uint32_t)(blk * 1024));
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

template <int MI, int NJ, int WGM, int WGN, int STAGES, int LOAD, int MFMA, int LDSR, int BK = 64>
void run(const char *name, const uint16_t *emb, int64_t rows, int dim, const uint16_t *q, int batch, float *tmax) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NJ * 16;
    if (BN < batch) { printf("%-44s skipped (BN %d < batch)\n", name, BN); return; }
    const int lds = STAGES * (BM + BN) * BK * 2;
    auto k = gemm_kernel<MI, NJ, WGM, WGN, STAGES, LOAD, MFMA, LDSR, BK>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const unsigned grid = (unsigned)((rows + BM - 1) / BM);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WGM * WGN * 64), lds, 0, emb, rows, dim, q, batch, tmax);
    CK(hipGetLastError());
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

int main() {
    const int64_t rows = 875000; const int dim = 768, batch = 256;
    uint16_t *emb, *q; float *tmax;
    CK(hipMalloc(&emb, (size_t)rows * dim * 2)); CK(hipMalloc(&q, (size_t)batch * dim * 2));
    CK(hipMalloc(&tmax, (size_t)(rows / 16 + 64) * batch * 4));   // one row per wave-row-tile of the smallest variant
    std::vector<uint16_t> h((size_t)rows * dim);
    uint32_t s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff)); }   // random bf16 around 0.01
    CK(hipMemcpy(emb, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, h.data(), (size_t)batch * dim * 2, hipMemcpyHostToDevice));
    //      MI NJ WGM WGN ST LOAD MFMA LDSR BK
    run<4, 4, 4, 4, 2, 3, 1, 1>("256x256 16 waves 64x64, 2 st, A nt", emb, rows, dim, q, batch, tmax);
    run<4, 4, 4, 4, 2, 3, 1, 1>("256x256 16 waves 64x64, 2 st, A nt (again)", emb, rows, dim, q, batch, tmax);
    run<4, 4, 4, 4, 4, 3, 1, 1, 32>("256x256 16 waves 64x64, 4 st BK 32, A nt", emb, rows, dim, q, batch, tmax);
    run<4, 4, 4, 4, 3, 3, 1, 1, 32>("256x256 16 waves 64x64, 3 st BK 32, A nt", emb, rows, dim, q, batch, tmax);
    run<8, 2, 2, 8, 2, 3, 1, 1>("256x256 16 waves 128x32, 2 st, A nt", emb, rows, dim, q, batch, tmax);
    run<2, 8, 8, 2, 2, 3, 1, 1>("256x256 16 waves 32x128, 2 st, A nt", emb, rows, dim, q, batch, tmax);
    run<4, 8, 4, 2, 2, 3, 1, 1>("256x256 8 waves 64x128, 2 st, A nt", emb, rows, dim, q, batch, tmax);
    run_asym<4, 8, 4, 2>("256x256 8 waves 64x128, A 3 st + B 2 st", emb, rows, dim, q, batch, tmax);
    run_asym<4, 4, 4, 4>("256x256 16 waves 64x64, A 3 st + B 2 st", emb, rows, dim, q, batch, tmax);
    run<4, 8, 4, 2, 2, 3, 1, 1>("256x256 8 waves 64x128, 2 st, A nt (again)", emb, rows, dim, q, batch, tmax);
    run<8, 4, 2, 4, 2, 3, 1, 1>("256x256 8 waves 128x64, 2 st, A nt", emb, rows, dim, q, batch, tmax);
    return 0;
}
