// Experiment harness (not product code): the B = 1 sweep with the HOT PREFIX of a hub-first numbered state held in LDS.
//
// With graph.degree_order the columns most entries point at are the lowest ids.  A persistent workgroup copies
// x[0 .. HOT) (fp16, up to 64 K vertices = 128 KB) into LDS once per sweep and serves every gather with col < HOT from
// there; the rest go to the L2 as before.  The B = 1 sweep is bound by the L2 request rate, so
// every gather that stays inside the CU is a request saved.  Column model of the benchmark generator (synth.make_kg):
// half of the entries draw their column Zipf(0.6) by popularity rank (= id under the hub-first numbering), half
// uniformly.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/spmv_hot_bench tools/spmv_hot_bench.hip && tools/_bin/spmv_hot_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// reference: CSR, 8 lanes per row, every gather from global memory (the structure of csrc/ppr_sv.hip at its simplest)
__global__ __launch_bounds__(256) void spmv_ref_kernel(const int *__restrict__ rp, const int *__restrict__ ci, const float *__restrict__ va,
                                                       const _Float16 *__restrict__ x, float *__restrict__ y, int n) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 3, gl = threadIdx.x & 7;
    if (row >= n) return;
    float a = 0.f;
    for (int k = rp[row] + gl; k < rp[row + 1]; k += 8) a = fmaf(va[k], (float)x[ci[k]], a);
    for (int o = 1; o < 8; o <<= 1) a += __shfl_xor(a, o, 64);
    if (gl == 0) y[row] = a;
}

// persistent workgroups: copy the hot prefix, then walk a contiguous range of rows, 8 lanes per row
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spmv_hot_kernel(const int *__restrict__ rp, const int *__restrict__ ci,
                                                           const float *__restrict__ va, const _Float16 *__restrict__ x,
                                                           float *__restrict__ y, int n, int hot, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *hx = reinterpret_cast<_Float16 *>(smem);
    const int tid = threadIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(x);
        uint4 *dst = reinterpret_cast<uint4 *>(hx);
        for (int i = tid; i < hot / 8; i += THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
    const int gl = tid & 7;
    for (int row = r0 + (tid >> 3); row < r1; row += THREADS / 8) {
        float a = 0.f;
        const int e = rp[row + 1];
        for (int k = rp[row] + gl; k < e; k += 8) {
            const int c = ci[k];
            const float xv = c < hot ? (float)hx[c] : (float)x[c];
            a = fmaf(va[k], xv, a);
        }
        for (int o = 1; o < 8; o <<= 1) a += __shfl_xor(a, o, 64);
        if (gl == 0) y[row] = a;
    }
}

int main(int argc, char **argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 1 << 20;
    const int deg = argc > 2 ? atoi(argv[2]) : 20;
    std::mt19937_64 rng(777);
    std::vector<int> rp(V + 1, 0);
    {
        std::poisson_distribution<int> pd(deg);
        for (int r = 0; r < V; ++r) rp[r + 1] = rp[r] + std::max(1, pd(rng));
    }
    const int64_t nnz = rp[V];
    std::vector<int> ci(nnz);
    std::vector<float> va(nnz, 0.05f);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    for (int64_t k = 0; k < nnz; ++k) {
        if (rng() & 1) ci[k] = std::min<int>(V - 1, (int)(std::pow(u01(rng), 2.5) * V));   // Zipf(0.6) rank = id (hub-first)
        else ci[k] = (int)(rng() % V);
    }
    for (int r = 0; r < V; ++r) std::sort(ci.begin() + rp[r], ci.begin() + rp[r + 1]);
    for (int hot : {16384, 32768, 65536}) {
        int64_t in = 0;
        for (int64_t k = 0; k < nnz; ++k) in += ci[k] < hot;
        printf("V=%d nnz=%lld  share of entries with col < %d: %.3f\n", V, (long long)nnz, hot, (double)in / nnz);
    }
    std::vector<_Float16> hx(V);
    for (int i = 0; i < V; ++i) hx[i] = (_Float16)(0.25f + (float)(i % 97) / 97.f);
    int *d_rp, *d_ci; float *d_va, *d_y, *d_y2; _Float16 *d_x;
    CK(hipMalloc(&d_rp, (V + 1) * 4)); CK(hipMalloc(&d_ci, nnz * 4)); CK(hipMalloc(&d_va, nnz * 4));
    CK(hipMalloc(&d_y, V * 4)); CK(hipMalloc(&d_y2, V * 4)); CK(hipMalloc(&d_x, V * 2));
    CK(hipMemcpy(d_rp, rp.data(), (V + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_va, va.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, hx.data(), V * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto &&launch, const char *name) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int n = 20;
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %7.1f us / sweep\n", name, ms * 1e3 / n);
    };
    time_it([&] { hipLaunchKernelGGL(spmv_ref_kernel, dim3((V * 8 + 255) / 256), dim3(256), 0, 0, d_rp, d_ci, d_va, d_x, d_y, V); },
            "global gathers (CSR, 8 lanes / row)");
    std::vector<float> y_ref(V), y_hot(V);
    CK(hipMemcpy(y_ref.data(), d_y, V * 4, hipMemcpyDeviceToHost));
    auto run_hot = [&](auto kernel, int threads, int hot, int wgs, const char *tag) {
        const int lds = hot * 2;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const int rows_per_wg = (V + wgs - 1) / wgs;
        char name[128];
        snprintf(name, sizeof name, "hot prefix %d in LDS, %d x %d threads%s", hot, wgs, threads, tag);
        time_it([&] { hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), lds, 0, d_rp, d_ci, d_va, d_x, d_y2, V, hot, rows_per_wg); }, name);
        CK(hipMemcpy(y_hot.data(), d_y2, V * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int i = 0; i < V; ++i) worst = std::max(worst, (double)std::fabs(y_hot[i] - y_ref[i]));
        if (worst != 0) printf("    MISMATCH %g\n", worst);
    };
    for (int hot : {0, 16384, 32768, 65536}) {
        run_hot(spmv_hot_kernel<1024>, 1024, hot, 256, "");
        run_hot(spmv_hot_kernel<1024>, 1024, hot, 512, "");     // 2 workgroups per CU only fit when hot <= 32768 (64 KB each)
        run_hot(spmv_hot_kernel<512>, 512, hot, 1024, "");
    }
    return 0;
}
