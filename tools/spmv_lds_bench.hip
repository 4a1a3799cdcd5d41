// Prototype harness (not product code): B = 1 PPR sweep with the state staged in LDS.
//
// The B = 1 sweep of csrc/ppr_sv.hip is bound by the L2 request rate: 20 M two-byte gathers at ~180 G requests/s
// = 110-136 us at cfg 3.  Here every CU owns 1/256 of the rows and walks the matrix COLUMN BLOCK
// by column block: a block of the fp16 state (65 536 columns = 128 KB) is copied into the CU's LDS once per phase and
// all gathers of the phase are LDS reads (the north star's "LDS-staged" form).  Per sweep: 256 x 2 MB of L2 -> LDS
// copies (the state is L2-resident) + one pass over the matrix from HBM; y accumulates in LDS, every (row, block)
// group of entries belongs to ONE lane, so there are no atomics and the summation order is fixed.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/spmv_lds_bench tools/spmv_lds_bench.hip
//   tools/_bin/spmv_lds_bench [V=1048576] [deg=20] [hubs=64] [cap=8] [rotate=1]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s -> %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int kWG = 256;        // workgroups = CUs
constexpr int kThreads = 1024;  // 16 wavefronts
constexpr int kWaves = kThreads / 64;
constexpr int kCB = 65536;      // columns per block (fp16: 128 KB of LDS)
constexpr int kYMax = 7168;     // accumulator slots per workgroup (real + virtual rows): 28 KB

typedef int v2i_t __attribute__((ext_vector_type(2)));

struct Tiles {
    const int2 *stream;      // entries: (lrow << 16 | col16, fp32 bits), step-major per (wg, phase, wave), 64 per step
    uint32_t stream_bytes;
    const int2 *meta;        // [kWG][n_phase][kWaves] (first step, n steps)
    int n_phase;
    const int32_t *row_of;   // [kWG][kYMax] global row of a real accumulator slot (-1: none / virtual)
    const int32_t *n_real;   // [kWG]
    const int32_t *vt;       // [kWG][kVtMax][3] (target slot, first virtual slot, count)
    const int32_t *n_vt;     // [kWG]
};
constexpr int kVtMax = 256;

__device__ __forceinline__ void lds_dma16(uint32_t lds_uniform, const void *g) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_uniform), "v"(g) : "memory");
}

template <bool DMA, int UNROLL>
__global__ __launch_bounds__(kThreads) void spmv_lds_kernel(Tiles t, const _Float16 *__restrict__ x, _Float16 *__restrict__ y,
                                                            const float *__restrict__ tele, float alpha, float beta, int rotate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *xs = reinterpret_cast<_Float16 *>(smem);
    float *yacc = reinterpret_cast<float *>(smem + (size_t)kCB * 2);
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kYMax; i += kThreads) yacc[i] = 0.f;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int2 *>(t.stream), 0, (int)t.stream_bytes, 0x00020000);
    const uint32_t xs_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)xs;   // LDS byte address
    for (int pp = 0; pp < t.n_phase; ++pp) {
        const int ph = rotate ? (pp + wg) % t.n_phase : pp;
        __syncthreads();   // everybody is done with the previous block image
        const char *src = reinterpret_cast<const char *>(x) + (size_t)ph * kCB * 2;
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < (kCB * 2) / (kThreads * 16); ++i) {
                const int chunk = i * kWaves + wave;   // 1 KB per wavefront instruction
                lds_dma16(__builtin_amdgcn_readfirstlane(xs_addr + (uint32_t)chunk * 1024u), src + (size_t)chunk * 1024 + lane * 16);
            }
        } else {
            uint4 tmp[(kCB * 2) / (kThreads * 16)];
#pragma unroll
            for (int i = 0; i < (kCB * 2) / (kThreads * 16); ++i)
                tmp[i] = reinterpret_cast<const uint4 *>(src)[i * kThreads + tid];
#pragma unroll
            for (int i = 0; i < (kCB * 2) / (kThreads * 16); ++i)
                reinterpret_cast<uint4 *>(xs)[i * kThreads + tid] = tmp[i];
        }
        const int2 m = t.meta[((size_t)wg * t.n_phase + ph) * kWaves + wave];
        const unsigned base = (unsigned)m.x * 512u, voff = (unsigned)lane * 8u;
        // first batch of this phase's entries: in flight together with the block copy
        v2i_t e[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) e[u] = __builtin_amdgcn_raw_buffer_load_b64(srs, voff + (unsigned)u * 512u, base, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the block image is complete
        for (int s0 = 0; s0 < m.y; s0 += UNROLL) {
            v2i_t nx[UNROLL];
            const bool more = s0 + UNROLL < m.y;   // wave-uniform
            if (more) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    nx[u] = __builtin_amdgcn_raw_buffer_load_b64(srs, voff + (unsigned)(s0 + UNROLL + u) * 512u, base, 2);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (s0 + u < m.y && e[u].y != 0) {   // padding entries carry val = 0 bits
                    const unsigned key = (unsigned)e[u].x;
                    const float xv = (float)xs[key & 0xffffu];
                    const unsigned r = key >> 16;
                    yacc[r] = fmaf(__int_as_float(e[u].y), xv, yacc[r]);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) e[u] = nx[u];
            }
        }
    }
    __syncthreads();
    // long (row, block) groups were cut into virtual rows: add them to their row in a fixed order
    const int nvt = t.n_vt[wg];
    for (int i = tid; i < nvt; i += kThreads) {
        const int32_t *v = t.vt + ((size_t)wg * kVtMax + i) * 3;
        float a = yacc[v[0]];
        for (int k = 0; k < v[2]; ++k) a += yacc[v[1] + k];
        yacc[v[0]] = a;
    }
    __syncthreads();
    const int nr = t.n_real[wg];
    for (int i = tid; i < nr; i += kThreads) {
        const int row = t.row_of[(size_t)wg * kYMax + i];
        y[row] = (_Float16)fmaf(alpha, yacc[i], beta * tele[row]);
    }
}

// v2: the block image is double-buffered (2 x kCB2 columns) and the copy of block p + 1 AND the first entries of phase
// p + 1 are in flight while phase p is processed: one barrier per phase, no exposed HBM latency in the steady state.
constexpr int kCB2 = 32768;
template <int UNROLL, bool DO_COPY = true, bool DO_ENTRIES = true>
__global__ __launch_bounds__(kThreads) void spmv_lds2_kernel(Tiles t, const _Float16 *__restrict__ x, _Float16 *__restrict__ y,
                                                             const float *__restrict__ tele, float alpha, float beta, int rotate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *xs = reinterpret_cast<_Float16 *>(smem);                       // [2][kCB2]
    float *yacc = reinterpret_cast<float *>(smem + (size_t)kCB2 * 2 * 2);
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kYMax; i += kThreads) yacc[i] = 0.f;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int2 *>(t.stream), 0, (int)t.stream_bytes, 0x00020000);
    const uint32_t xs_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)xs;
    const int np = t.n_phase;
    auto phase_of = [&](int pp) { return rotate ? (pp + wg) % np : pp; };
    auto copy_block = [&](int ph, int buf) {
        const char *src = reinterpret_cast<const char *>(x) + (size_t)ph * kCB2 * 2;
#pragma unroll
        for (int i = 0; i < (kCB2 * 2) / (kThreads * 16); ++i) {
            const int chunk = i * kWaves + wave;   // 1 KB per wavefront instruction
            lds_dma16(__builtin_amdgcn_readfirstlane(xs_addr + (uint32_t)buf * (kCB2 * 2) + (uint32_t)chunk * 1024u),
                      src + (size_t)chunk * 1024 + lane * 16);
        }
    };
    const unsigned voff = (unsigned)lane * 8u;
    int2 m = t.meta[((size_t)wg * np + phase_of(0)) * kWaves + wave];
    v2i_t e[UNROLL];
    if constexpr (DO_COPY) copy_block(phase_of(0), 0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) e[u] = DO_ENTRIES ? __builtin_amdgcn_raw_buffer_load_b64(srs, voff + (unsigned)u * 512u, (unsigned)m.x * 512u, 2) : v2i_t{0, 0};
    for (int pp = 0; pp < np; ++pp) {
        const int buf = pp & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this phase's block image and first entries have landed
        __syncthreads();                                    // ... for everybody; the other buffer is free again
        int2 mn = make_int2(0, 0);
        v2i_t en[UNROLL];
        if (pp + 1 < np) {                                   // next phase: block copy + first entries, in flight from here
            if constexpr (DO_COPY) copy_block(phase_of(pp + 1), buf ^ 1);
            mn = t.meta[((size_t)wg * np + phase_of(pp + 1)) * kWaves + wave];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) en[u] = DO_ENTRIES ? __builtin_amdgcn_raw_buffer_load_b64(srs, voff + (unsigned)u * 512u, (unsigned)mn.x * 512u, 2) : v2i_t{0, 0};
        }
        const _Float16 *xb = xs + (size_t)buf * kCB2;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (u < m.y && e[u].y != 0) {
                const unsigned key = (unsigned)e[u].x;
                const unsigned r = key >> 16;
                yacc[r] = fmaf(__int_as_float(e[u].y), (float)xb[key & 0xffffu], yacc[r]);
            }
        }
        for (int s0 = UNROLL; DO_ENTRIES && s0 < m.y; ++s0) { // rare: a wavefront with more than UNROLL steps in a phase
            const v2i_t ex = __builtin_amdgcn_raw_buffer_load_b64(srs, voff + (unsigned)s0 * 512u, (unsigned)m.x * 512u, 2);
            if (ex.y != 0) {
                const unsigned key = (unsigned)ex.x;
                const unsigned r = key >> 16;
                yacc[r] = fmaf(__int_as_float(ex.y), (float)xb[key & 0xffffu], yacc[r]);
            }
        }
        m = mn;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) e[u] = en[u];
    }
    __syncthreads();
    const int nvt = t.n_vt[wg];
    for (int i = tid; i < nvt; i += kThreads) {
        const int32_t *v = t.vt + ((size_t)wg * kVtMax + i) * 3;
        float a = yacc[v[0]];
        for (int k = 0; k < v[2]; ++k) a += yacc[v[1] + k];
        yacc[v[0]] = a;
    }
    __syncthreads();
    const int nr = t.n_real[wg];
    for (int i = tid; i < nr; i += kThreads) {
        const int row = t.row_of[(size_t)wg * kYMax + i];
        y[row] = (_Float16)fmaf(alpha, yacc[i], beta * tele[row]);
    }
}

// reference gather kernel from global memory (the structure of csrc/ppr_sv.hip at its simplest: CSR, 8 lanes per row)
__global__ __launch_bounds__(256) void spmv_ref_kernel(const int *__restrict__ rp, const int *__restrict__ ci, const float *__restrict__ va,
                                                       const _Float16 *__restrict__ x, float *__restrict__ y, int n) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 3, gl = threadIdx.x & 7;
    if (row >= n) return;
    float a = 0.f;
    for (int k = rp[row] + gl; k < rp[row + 1]; k += 8) a = fmaf(va[k], (float)x[ci[k]], a);
    for (int o = 1; o < 8; o <<= 1) a += __shfl_xor(a, o, 64);
    if (gl == 0) y[row] = a;
}

int main(int argc, char **argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 1048576;
    const int deg = argc > 2 ? atoi(argv[2]) : 20;
    const int hubs = argc > 3 ? atoi(argv[3]) : 64;
    const int cap = argc > 4 ? atoi(argv[4]) : 16;
    const int rotate = argc > 5 ? atoi(argv[5]) : 1;
    const int cb = argc > 6 ? atoi(argv[6]) : kCB;   // columns per block: 65536 (v1 kernels) or 32768 (v2, double-buffered)
    if (V % cb || (cb != kCB && cb != kCB2)) { fprintf(stderr, "V must be a multiple of the block size %d\n", cb); return 1; }
    const int n_phase = V / cb;
    std::mt19937_64 rng(12345);
    // ---- graph: Poisson(deg) rows + `hubs` rows of 2000..20000 entries, uniform columns
    std::vector<int> rp(V + 1, 0);
    {
        std::poisson_distribution<int> pd(deg);
        for (int r = 0; r < V; ++r) rp[r + 1] = std::max(1, pd(rng));
        for (int h = 0; h < hubs; ++h) rp[1 + (rng() % V)] = 2000 + (int)(rng() % 18000);
        for (int r = 0; r < V; ++r) rp[r + 1] += rp[r];
    }
    const int64_t nnz = rp[V];
    std::vector<int> ci(nnz);
    std::vector<float> va(nnz);
    for (int r = 0; r < V; ++r) {
        const float w = 1.0f / (float)(rp[r + 1] - rp[r]);
        for (int k = rp[r]; k < rp[r + 1]; ++k) { ci[k] = (int)(rng() % V); va[k] = w * (0.5f + (float)(rng() % 1000) / 1000.f); }
    }
    printf("V=%d nnz=%lld block=%d phases=%d cap=%d rotate=%d\n", V, (long long)nnz, cb, n_phase, cap, rotate);

    // ---- tiles.  Rows sorted by length and dealt round-robin to the workgroups
    std::vector<int> order(V);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rp[a + 1] - rp[a] > rp[b + 1] - rp[b]; });
    std::vector<int32_t> row_of((size_t)kWG * kYMax, -1), n_real(kWG, 0), n_vt(kWG, 0), vt((size_t)kWG * kVtMax * 3, 0);
    std::vector<int2> meta((size_t)kWG * n_phase * kWaves);
    std::vector<int2> stream;
    stream.reserve((size_t)(nnz * 1.4));
    struct Group { int slot, begin, cnt; };   // entries [begin, begin + cnt) of `ent`
    int64_t pad_steps = 0, real_steps = 0;
    int max_slots = 0;
    for (int wg = 0; wg < kWG; ++wg) {
        std::vector<int> rows;
        for (int i = wg; i < V; i += kWG) rows.push_back(order[i]);
        const int nr = (int)rows.size();
        n_real[wg] = nr;
        for (int i = 0; i < nr; ++i) row_of[(size_t)wg * kYMax + i] = rows[i];
        // bucket the entries of this workgroup's rows by column block
        std::vector<std::vector<std::pair<int, int>>> by_phase(n_phase);   // (lrow, k)
        for (int i = 0; i < nr; ++i)
            for (int k = rp[rows[i]]; k < rp[rows[i] + 1]; ++k) by_phase[ci[k] / cb].push_back({i, k});
        // a (row, block) group of more than `cap` entries is cut: the extra parts accumulate in virtual slots, ONE
        // contiguous range per row (added to the row in slot order by one thread of the epilogue)
        std::vector<int> vbase(nr, 0), vnext(nr, 0);
        int next_slot = nr, nv = 0;
        {
            std::vector<int> extra(nr, 0);
            for (int ph = 0; ph < n_phase; ++ph) {
                auto &ent = by_phase[ph];
                for (size_t s = 0; s < ent.size();) {
                    size_t e = s;
                    while (e < ent.size() && ent[e].first == ent[s].first) ++e;
                    extra[ent[s].first] += ((int)(e - s) + cap - 1) / cap - 1;
                    s = e;
                }
            }
            for (int i = 0; i < nr; ++i) {
                if (!extra[i]) continue;
                vbase[i] = next_slot;
                next_slot += extra[i];
                if (nv >= kVtMax) { fprintf(stderr, "too many cut rows in workgroup %d\n", wg); return 1; }
                int32_t *v = &vt[((size_t)wg * kVtMax + nv) * 3];
                v[0] = i; v[1] = vbase[i]; v[2] = extra[i];
                ++nv;
            }
        }
        for (int ph = 0; ph < n_phase; ++ph) {
            auto &b = by_phase[ph];   // already grouped by lrow (rows were appended in order)
            std::vector<Group> groups;
            std::vector<std::pair<int, int>> &ent = b;
            for (size_t s = 0; s < ent.size();) {
                size_t e = s;
                while (e < ent.size() && ent[e].first == ent[s].first) ++e;
                int left = (int)(e - s), at = (int)s;
                bool first = true;
                while (left > 0) {
                    const int c = std::min(left, cap);
                    int slot = ent[s].first;
                    if (!first) slot = vbase[ent[s].first] + vnext[ent[s].first]++;
                    groups.push_back({slot, at, c});
                    at += c; left -= c; first = false;
                }
                s = e;
            }
            // longest-processing-time deal of the groups to the 1024 lanes
            std::stable_sort(groups.begin(), groups.end(), [](const Group &a, const Group &b) { return a.cnt > b.cnt; });
            std::vector<std::vector<int>> lane_items(kThreads);
            std::vector<int> load(kThreads, 0);
            // groups are sorted: round-robin in serpentine order is within one entry of LPT and O(n)
            for (size_t g = 0; g < groups.size(); ++g) {
                const size_t rnd = g / kThreads, pos = g % kThreads;
                const int ln = (rnd & 1) ? (int)(kThreads - 1 - pos) : (int)pos;
                lane_items[ln].push_back((int)g);
                load[ln] += groups[g].cnt;
            }
            for (int w = 0; w < kWaves; ++w) {
                int steps = 0;
                for (int l = 0; l < 64; ++l) steps = std::max(steps, load[w * 64 + l]);
                meta[((size_t)wg * n_phase + ph) * kWaves + w] = make_int2((int)(stream.size() / 64), steps);
                const size_t at0 = stream.size();
                stream.resize(at0 + (size_t)steps * 64, make_int2(0, 0));
                for (int l = 0; l < 64; ++l) {
                    int s = 0;
                    for (int gi : lane_items[w * 64 + l]) {
                        const Group &g = groups[gi];
                        for (int k = 0; k < g.cnt; ++k, ++s) {
                            const int kk = ent[g.begin + k].second;
                            float v = va[kk];
                            int bits;
                            memcpy(&bits, &v, 4);
                            if (bits == 0) bits = 1;   // a true zero weight would look like padding: make it a denormal
                            stream[at0 + (size_t)s * 64 + l] = make_int2((g.slot << 16) | (ci[kk] % cb), bits);
                        }
                    }
                    real_steps += s;
                    pad_steps += steps - s;
                }
            }
        }
        n_vt[wg] = nv;
        max_slots = std::max(max_slots, next_slot);
        if (next_slot > kYMax || next_slot > 65535) { fprintf(stderr, "workgroup %d needs %d accumulator slots\n", wg, next_slot); return 1; }
    }
    stream.resize(stream.size() + 64 * 16, make_int2(0, 0));   // read-ahead padding
    printf("stream %.1f MB (entries %.1f MB, padding %.1f %%), max accumulator slots %d\n", stream.size() * 8 / 1e6, nnz * 8 / 1e6,
           100.0 * pad_steps / std::max<int64_t>(1, real_steps), max_slots);

    // ---- device
    std::vector<_Float16> hx(V);
    std::vector<float> htele(V);
    for (int i = 0; i < V; ++i) { hx[i] = (_Float16)((float)(rng() % 1000) / 1000.f); htele[i] = (float)(rng() % 100) / 1000.f; }
    int2 *d_stream, *d_meta;
    int32_t *d_rowof, *d_nreal, *d_vt, *d_nvt;
    _Float16 *d_x, *d_y;
    float *d_tele, *d_yref, *d_va;
    int *d_rp, *d_ci;
    CK(hipMalloc(&d_stream, stream.size() * 8)); CK(hipMemcpy(d_stream, stream.data(), stream.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_meta, meta.size() * 8)); CK(hipMemcpy(d_meta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_rowof, row_of.size() * 4)); CK(hipMemcpy(d_rowof, row_of.data(), row_of.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nreal, kWG * 4)); CK(hipMemcpy(d_nreal, n_real.data(), kWG * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_vt, vt.size() * 4)); CK(hipMemcpy(d_vt, vt.data(), vt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nvt, kWG * 4)); CK(hipMemcpy(d_nvt, n_vt.data(), kWG * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, (size_t)V * 2)); CK(hipMemcpy(d_x, hx.data(), (size_t)V * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, (size_t)V * 2)); CK(hipMemset(d_y, 0, (size_t)V * 2));
    CK(hipMalloc(&d_tele, (size_t)V * 4)); CK(hipMemcpy(d_tele, htele.data(), (size_t)V * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_yref, (size_t)V * 4));
    CK(hipMalloc(&d_rp, (size_t)(V + 1) * 4)); CK(hipMemcpy(d_rp, rp.data(), (size_t)(V + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ci, nnz * 4)); CK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_va, nnz * 4)); CK(hipMemcpy(d_va, va.data(), nnz * 4, hipMemcpyHostToDevice));
    Tiles t{d_stream, (uint32_t)(stream.size() * 8), d_meta, n_phase, d_rowof, d_nreal, d_vt, d_nvt};
    const size_t smem = (size_t)kCB * 2 + (size_t)kYMax * 4;
    const float alpha = 0.5f, beta = 0.5f;

    int grid_wgs = kWG;
    auto run = [&](auto kern, const char *name) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3(grid_wgs), dim3(kThreads), smem, 0, t, d_x, d_y, d_tele, alpha, beta, rotate);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        // check against the CPU (double)
        std::vector<_Float16> hy(V);
        CK(hipMemcpy(hy.data(), d_y, (size_t)V * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int r = 0; r < V; r += 97) {
            double a = 0;
            for (int k = rp[r]; k < rp[r + 1]; ++k) a += (double)va[k] * (double)(float)hx[ci[k]];
            const double want = alpha * a + beta * htele[r];
            worst = std::max(worst, std::fabs((double)(float)hy[r] - want) / std::max(1e-6, std::fabs(want)));
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int n = 40;
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(grid_wgs), dim3(kThreads), smem, 0, t, (i & 1) ? d_y : d_x, (i & 1) ? d_x : d_y, d_tele, alpha, beta, rotate);
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid_wgs), dim3(kThreads), smem, 0, t, (i & 1) ? d_y : d_x, (i & 1) ? d_x : d_y, d_tele, alpha, beta, rotate);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.1f us / sweep   (max rel err vs fp64 on the fp16 result %.2e; stream %.2f TB/s)\n", name, ms * 1e3 / n, worst,
               stream.size() * 8 / (ms * 1e-3 / n) / 1e12);
        CK(hipMemcpy(d_x, hx.data(), (size_t)V * 2, hipMemcpyHostToDevice));
    };
    if (cb == kCB) {
        run(spmv_lds_kernel<true, 4>, "lds-blocked, DMA copy, unroll 4");
        run(spmv_lds_kernel<true, 8>, "lds-blocked, DMA copy, unroll 8");
        run(spmv_lds_kernel<false, 4>, "lds-blocked, register copy, unroll 4");
    } else {
        run(spmv_lds2_kernel<4, true, false>, "v2 COPY ONLY (no entries)");
        grid_wgs = 128; run(spmv_lds2_kernel<4, true, false>, "v2 COPY ONLY, 128 workgroups");
        grid_wgs = 64; run(spmv_lds2_kernel<4, true, false>, "v2 COPY ONLY, 64 workgroups");
        grid_wgs = kWG;
        run(spmv_lds2_kernel<4, false, true>, "v2 ENTRIES ONLY (no block copies)");
        run(spmv_lds2_kernel<4>, "v2 double-buffered, prefetch 4");
        run(spmv_lds2_kernel<6>, "v2 double-buffered, prefetch 6");
        run(spmv_lds2_kernel<8>, "v2 double-buffered, prefetch 8");
    }
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int n = 20;
        hipLaunchKernelGGL(spmv_ref_kernel, dim3((V * 8 + 255) / 256), dim3(256), 0, 0, d_rp, d_ci, d_va, d_x, d_yref, V);
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spmv_ref_kernel, dim3((V * 8 + 255) / 256), dim3(256), 0, 0, d_rp, d_ci, d_va, d_x, d_yref, V);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.1f us / sweep\n", "global gathers (CSR, 8 lanes / row)", ms * 1e3 / n);
    }
    return 0;
}
