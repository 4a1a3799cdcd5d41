// Memory-system microbenchmarks for the PPR SpMM design (not product code):
//   stream  : float4 streaming read of a buffer of S bytes
//   gather  : pseudo-random chunk gathers (chunk = g bytes, g/16 lanes per chunk) from a buffer of S bytes
//   replay  : the gathers of a REAL column-index stream (membench --cols <int32 file> [--rows V]): every index fetches
//             one piece of g bytes from a state of V pieces -- the gather traffic of one PPR sweep and nothing else
// Prints GB/s for each (S, g).  Usage: membench   |   membench --cols cols.bin --rows 1000000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4 *p, size_t n4, float *out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = p[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// LPC = lanes per chunk (chunk bytes = 16 * LPC); each lane group performs `iters` x U gathers
template <int LPC, int U>
__global__ __launch_bounds__(256) void gather_kernel(const float4 *p, uint32_t n_chunks, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t grp = tid / LPC, gl = tid % LPC;
    float4 acc = make_float4(0, 0, 0, 0);
    uint32_t s = grp * 0x9E3779B9u + 12345u;
    for (int it = 0; it < iters; ++it) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s = mix(s + u + 1);
            const uint32_t c = (uint32_t)(((uint64_t)s * n_chunks) >> 32);
            v[u] = p[(size_t)c * LPC + gl];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// replay of a column stream: lane group `grp` takes entries grp, grp + n_groups, ... (8 in flight per lane)
template <int LPC>
__global__ __launch_bounds__(256) void replay_kernel(const float4 *p, const int32_t *__restrict__ cols, int64_t nnz,
                                                     float *out) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t grp = tid / LPC, n_groups = (int64_t)gridDim.x * 256 / LPC;
    const int gl = (int)(tid % LPC);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t e = grp; e < nnz; e += n_groups * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = e + (int64_t)u * n_groups;
            v[u] = i < nnz ? p[(size_t)cols[i] * LPC + gl] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// the small-batch shape: every lane owns its entries and fetches ES bytes (2 = one fp16 value, B = 1) per entry
template <typename T>
__global__ __launch_bounds__(256) void replay_lane_kernel(const T *p, const int32_t *__restrict__ cols, int64_t nnz, float *out) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, n = (int64_t)gridDim.x * 256;
    float acc = 0.f;
    for (int64_t e = tid; e < nnz; e += n * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = e + (int64_t)u * n;
            v[u] = i < nnz ? p[cols[i]] : T{};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (float)v[u];
    }
    if (acc == 123.456f) out[0] = 1.f;
}
template <typename T>
void run_replay_lane(const void *buf, const int32_t *cols, int64_t nnz, float *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 16;
    hipLaunchKernelGGL(replay_lane_kernel<T>, dim3(blocks), dim3(256), 0, 0, (const T *)buf, cols, nnz, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(replay_lane_kernel<T>, dim3(blocks), dim3(256), 0, 0, (const T *)buf, cols, nnz, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("  per-lane %zu-byte gathers: %8.3f ms per pass, %6.1f G gathers/s\n", sizeof(T), ms, (double)nnz / (ms * 1e-3) / 1e9);
}

template <int LPC>
void run_replay(const float4 *buf, const int32_t *cols, int64_t nnz, float *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 24;
    hipLaunchKernelGGL((replay_kernel<LPC>), dim3(blocks), dim3(256), 0, 0, buf, cols, nnz, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((replay_kernel<LPC>), dim3(blocks), dim3(256), 0, 0, buf, cols, nnz, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("  piece %4d B: %8.3f ms per pass, %8.0f GB/s useful (+ %0.0f MB of indices)\n", 16 * LPC, ms,
           (double)nnz * 16 * LPC / (ms * 1e-3) / 1e9, (double)nnz * 4 / 1e6);
}

template <int LPC>
double run_gather(const float4 *buf, size_t bytes, float *out, int blocks, int iters) {
    const uint32_t n_chunks = (uint32_t)(bytes / (16 * LPC));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<LPC, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_chunks, iters, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r)
        hipLaunchKernelGGL((gather_kernel<LPC, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_chunks, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double total = 3.0 * (double)blocks * 256 * iters * 8 * 16;
    return total / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv) {
    const char *cols_path = nullptr;
    int64_t rows = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--cols")) cols_path = argv[i + 1];
        if (!strcmp(argv[i], "--rows")) rows = atoll(argv[i + 1]);
    }
    if (cols_path) {
        FILE *f = fopen(cols_path, "rb");
        if (!f) { printf("cannot open %s\n", cols_path); return 1; }
        fseek(f, 0, SEEK_END);
        const int64_t nnz = ftell(f) / 4;
        fseek(f, 0, SEEK_SET);
        std::vector<int32_t> h((size_t)nnz);
        if (fread(h.data(), 4, (size_t)nnz, f) != (size_t)nnz) { printf("short read\n"); return 1; }
        fclose(f);
        if (rows <= 0) for (int32_t c : h) rows = std::max<int64_t>(rows, (int64_t)c + 1);
        int32_t *dcols; float4 *state; float *o;
        CK(hipMalloc(&dcols, (size_t)nnz * 4)); CK(hipMemcpy(dcols, h.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&state, (size_t)rows * 1024)); CK(hipMemset(state, 0, (size_t)rows * 1024)); CK(hipMalloc(&o, 64));
        printf("replay of %lld column indices over %lld state rows (gathers only: no streams, no arithmetic)\n",
               (long long)nnz, (long long)rows);
        run_replay<8>(state, dcols, nnz, o);
        run_replay<16>(state, dcols, nnz, o);
        run_replay<32>(state, dcols, nnz, o);
        run_replay_lane<_Float16>(state, dcols, nnz, o);
        run_replay_lane<float>(state, dcols, nnz, o);
        return 0;
    }
    const size_t maxb = (size_t)4 << 30;
    float4 *buf; float *out;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, maxb));
    const size_t sizes[] = {2u << 20, 16u << 20, 64u << 20, 128u << 20, 192u << 20, 256u << 20, 512u << 20, (size_t)1 << 30, (size_t)4 << 30};
    printf("stream read GB/s\n");
    for (size_t S : sizes) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = (int)std::max<size_t>(4, ((size_t)8 << 30) / S);
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, S / 16, out);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, S / 16, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  S=%6zu MiB  %8.0f\n", S >> 20, (double)S * reps / (ms * 1e-3) / 1e9);
    }
    printf("random gather GB/s (useful bytes), columns: chunk 64 128 256 512 1024 B\n");
    const int blocks = 256 * 32, iters = 32;
    for (size_t S : sizes) {
        printf("  S=%6zu MiB ", S >> 20);
        printf(" %8.0f", run_gather<4>(buf, S, out, blocks, iters));
        printf(" %8.0f", run_gather<8>(buf, S, out, blocks, iters));
        printf(" %8.0f", run_gather<16>(buf, S, out, blocks, iters));
        printf(" %8.0f", run_gather<32>(buf, S, out, blocks, iters));
        printf(" %8.0f\n", run_gather<64>(buf, S, out, blocks, iters));
        fflush(stdout);
    }
    return 0;
}
