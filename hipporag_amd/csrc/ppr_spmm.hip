// K3 -- Personalized PageRank power iteration as a CSR SpMM over a slab of right-hand sides.
//
// Replaces igraph/PRPACK behind HippoRAG.run_ppr (reference src/hipporag/HippoRAG.py:1736-1743)
// with the fixed-count leaky iteration  y = alpha * P x + (1 - alpha) * v  (SURVEY.md section 7).
//
// Layout (DESIGN.md "PPR state"): x, y are [n_slabs][V][BC] fp32, BC = 4*G queries per slab.
// One gather of a source vertex therefore reads BC*4 contiguous bytes (128 B = one cache line at
// BC = 32) that serve BC queries at once; a slab (V*BC*4 bytes) is sized to stay resident in the
// 256 MiB Infinity Cache while its rows are gathered nnz times, and blockIdx.y (slowest) walks the
// slabs so that only ~one slab is live at a time.
//
// Main kernel: G lanes own one output row (64/G rows per 64-wide wavefront).  The G lanes fetch
// G consecutive (col, val) pairs with one coalesced load each, broadcast them inside the group
// with ds_swizzle (no LDS storage, no address VGPRs) and issue G independent float4 gathers per
// step, so a wavefront keeps 64 gathers x 16 B in flight per step.  Rows are visited in
// degree-descending order (row_order), which makes the trip count uniform inside a wavefront and
// starts the heaviest rows first.  A row's entries are walked serially by its G lanes (one
// dependent gather round per G entries, ~1 us each), so rows above the short-row threshold
// (8 rounds) are cut into segments handled one wavefront each (ppr_spmm_seg_kernel) -- otherwise a
// single hub row (degree 39 512 at cfg 3) would set the duration of the whole sweep.
// No atomics anywhere: results are bit-reproducible.
#include "common.h"

namespace hrag {
namespace {

// ds_swizzle needs an immediate pattern, so the broadcast source lane k is a template parameter.
template <int G, int K>
__device__ __forceinline__ int group_bcast(int v) {
    if constexpr (G == 1) {
        return v;
    } else {
        constexpr int and_mask = 0x1f & ~(G - 1);
        constexpr int pattern = (K << 5) | and_mask;
        return __builtin_amdgcn_ds_swizzle(v, pattern);
    }
}

__device__ __forceinline__ void fma4(float4 &acc, float w, const float4 &x) {
    acc.x = fmaf(w, x.x, acc.x);
    acc.y = fmaf(w, x.y, acc.y);
    acc.z = fmaf(w, x.z, acc.z);
    acc.w = fmaf(w, x.w, acc.w);
}

template <int G, int K>
struct GatherStep {
    __device__ __forceinline__ static void run(float4 &acc, int c, float w, const float4 *xs,
                                               int gl) {
        const int ck = group_bcast<G, K>(c);
        const float wk = __int_as_float(group_bcast<G, K>(__float_as_int(w)));
        const float4 xv = xs[(size_t)ck * G + gl];
        fma4(acc, wk, xv);
        if constexpr (K + 1 < G) GatherStep<G, K + 1>::run(acc, c, w, xs, gl);
    }
};

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int G, bool NT_ST = false>
__device__ __forceinline__ void write_row(const SpmmArgs &a, int slab, int row, int gl,
                                          const float4 &acc) {
    constexpr int BC = 4 * G;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t slot = a.row_to_tele ? (int64_t)a.row_to_tele[row] : a.row_offset + row;
    if (slot >= 0) {
        const float4 *tp = reinterpret_cast<const float4 *>(
            a.tele + ((size_t)slab * a.tele_rows + (size_t)slot) * BC);
        t = tp[gl];
    }
    float4 out;
    out.x = fmaf(a.alpha, acc.x, a.beta * t.x);
    out.y = fmaf(a.alpha, acc.y, a.beta * t.y);
    out.z = fmaf(a.alpha, acc.z, a.beta * t.z);
    out.w = fmaf(a.alpha, acc.w, a.beta * t.w);
    float4 *yp = reinterpret_cast<float4 *>(a.y + (size_t)slab * a.num_vertices * BC);
    if constexpr (NT_ST) {
        f32x4_t o = {out.x, out.y, out.z, out.w};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t *>(yp + (size_t)(a.row_offset + row) * G + gl));
    } else {
        yp[(size_t)(a.row_offset + row) * G + gl] = out;
    }
}

template <bool NT>
__device__ __forceinline__ int ld_i(const int32_t *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ float ld_f(const float *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <int G, bool NT_CSR, bool NT_ST>
__global__ __launch_bounds__(256) void ppr_spmm_kernel(const SpmmArgs a) {
    constexpr int BC = 4 * G;
    constexpr int RPW = 64 / G;   // rows per wavefront
    constexpr int RPB = 4 * RPW;  // rows per 256-thread workgroup
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int gl = lane & (G - 1);
    const int grp = lane / G;
    const int slab = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * RPB + wave * RPW + grp;
    const bool active = r < a.n_short;
    int row = 0, e = 0, end = 0;
    if (active) {
        row = a.row_order[r];
        e = a.row_ptr[row];
        end = a.row_ptr[row + 1];
    }
    const float4 *xs = reinterpret_cast<const float4 *>(a.x + (size_t)slab * a.num_vertices * BC);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0;
    float w = 0.f;
    if (e + gl < end) {
        c = ld_i<NT_CSR>(a.col_idx + e + gl);
        w = ld_f<NT_CSR>(a.val + e + gl);
    }
    while (e < end) {  // trip count is uniform inside a G-lane group
        // prefetch the next G (col, val) pairs before the dependent gathers of this step
        int cn = 0;
        float wn = 0.f;
        const int en = e + G + gl;
        if (en < end) {
            cn = ld_i<NT_CSR>(a.col_idx + en);
            wn = ld_f<NT_CSR>(a.val + en);
        }
        // tail lanes carry (c = 0, w = 0): they gather row 0 (cache resident) and add 0
        GatherStep<G, 0>::run(acc, c, w, xs, gl);
        c = cn;
        w = wn;
        e += G;
    }
    if (active) write_row<G, NT_ST>(a, slab, row, gl, acc);
}

// Rows longer than the short-row threshold are cut into segments of <= seg_len entries; one
// wavefront per segment.  The 64 lanes fetch 64 consecutive (col, val) pairs with one coalesced
// load each; group g (G lanes) then walks pairs g*G .. g*G+G-1 with the same swizzle broadcast as
// the short-row kernel, so a step is again 64/G row gathers of BC*4 bytes.  The 64/G partial sums
// are combined with xor-shuffles in a fixed order.  Single-segment rows are finished here;
// multi-segment rows leave per-segment partials that ppr_spmm_reduce_kernel adds in segment order
// (deterministic, no atomics).
template <int G>
__global__ __launch_bounds__(256) void ppr_spmm_seg_kernel(const SpmmArgs a) {
    constexpr int BC = 4 * G;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int gl = lane & (G - 1);
    const int grp = lane / G;
    const int slab = blockIdx.y;
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= a.n_seg) return;  // wave-uniform
    const int row = a.seg_row[seg];
    const int e_end = a.seg_end[seg];
    const int slot = a.seg_slot[seg];
    const float4 *xs = reinterpret_cast<const float4 *>(a.x + (size_t)slab * a.num_vertices * BC);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = a.seg_begin[seg]; base < e_end; base += 64) {
        const int idx = base + lane;
        int c = 0;
        float w = 0.f;
        if (idx < e_end) {
            c = a.col_idx[idx];
            w = a.val[idx];
        }
        GatherStep<G, 0>::run(acc, c, w, xs, gl);
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64);
        acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64);
        acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (grp == 0) {
        if (slot < 0) {
            write_row<G>(a, slab, row, gl, acc);
        } else {
            reinterpret_cast<float4 *>(a.partial)[((size_t)slab * a.n_partial + slot) * G + gl] = acc;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void ppr_spmm_reduce_kernel(const SpmmArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t / G, gl = t % G;
    const int slab = blockIdx.y;
    if (m >= a.n_mrow) return;
    const int first = a.mrow_first[m], cnt = a.mrow_cnt[m];
    const float4 *pp = reinterpret_cast<const float4 *>(a.partial) + ((size_t)slab * a.n_partial + first) * G + gl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < cnt; ++s) {
        const float4 v = pp[(size_t)s * G];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    write_row<G>(a, slab, a.mrow_row[m], gl, acc);
}

// x0 = v: owned rows of y <- teleport slot (or 0).
template <int G>
__global__ __launch_bounds__(256) void ppr_init_kernel(const SpmmArgs a) {
    constexpr int BC = 4 * G;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / G;
    const int gl = (int)(t % G);
    const int slab = blockIdx.y;
    if (row >= a.n_rows) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t slot = a.row_to_tele ? (int64_t)a.row_to_tele[row] : a.row_offset + row;
    if (slot >= 0)
        v = reinterpret_cast<const float4 *>(a.tele + ((size_t)slab * a.tele_rows + (size_t)slot) * BC)[gl];
    reinterpret_cast<float4 *>(a.y + (size_t)slab * a.num_vertices * BC)[(size_t)(a.row_offset + row) * G + gl] = v;
}

__global__ void seed_scatter_kernel(float *y, int64_t num_vertices, int64_t row_offset,
                                    int64_t n_rows, const int32_t *seed_vtx, const float *seed_w,
                                    const int32_t *seed_cnt, int32_t max_seeds, int32_t batch,
                                    float scale, int32_t bc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = t / max_seeds;
    const int j = t % max_seeds;
    if (q >= batch || j >= seed_cnt[q]) return;
    const int64_t v = seed_vtx[q * max_seeds + j];
    const int64_t lv = v - row_offset;
    if (lv < 0 || lv >= n_rows) return;
    const int slab = q / bc, col = q % bc;
    // vertices are unique per query and queries own distinct columns: no write conflict
    y[((size_t)slab * num_vertices + (size_t)v) * bc + col] += scale * seed_w[q * max_seeds + j];
}

template <int G>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *x, int64_t num_vertices,
                                                             int64_t row_offset, int64_t n_rows,
                                                             double *partial) {
    constexpr int BC = 4 * G;
    constexpr int NG = 256 / G;
    __shared__ double red[256 * 4];
    const int tid = threadIdx.x;
    const int gl = tid & (G - 1);
    const int g = tid / G;
    const int slab = blockIdx.y;
    const float4 *xs = reinterpret_cast<const float4 *>(x + (size_t)slab * num_vertices * BC);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t r = (int64_t)blockIdx.x * NG + g; r < n_rows; r += (int64_t)gridDim.x * NG) {
        const float4 v = xs[(size_t)(row_offset + r) * G + gl];
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    red[tid * 4 + 0] = s0; red[tid * 4 + 1] = s1; red[tid * 4 + 2] = s2; red[tid * 4 + 3] = s3;
    __syncthreads();
    for (int s = 128; s >= G; s >>= 1) {
        if (tid < s)
            for (int j = 0; j < 4; ++j) red[tid * 4 + j] += red[(tid + s) * 4 + j];
        __syncthreads();
    }
    if (tid < G) {
        double *p = partial + ((size_t)slab * gridDim.x + blockIdx.x) * BC + gl * 4;
        for (int j = 0; j < 4; ++j) p[j] = red[tid * 4 + j];
    }
}

// one wavefront per query: lane l adds blocks l, l + 64, ... in that order, then a fixed xor-butterfly
// (a single thread walking all kColsumBlocks partials took 62 us -- 4 % of a cfg-2 batch)
__global__ __launch_bounds__(64) void colsum_final_kernel(const double *partial, int32_t batch, int32_t bc,
                                                          int32_t nblk, double *sums) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= batch) return;
    const int slab = q / bc, col = q % bc;
    double s = 0;
    for (int b = lane; b < nblk; b += 64) s += partial[((size_t)slab * nblk + b) * bc + col];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) sums[q] = s;
}

template <int G>
hrag_status spmm_dispatch(const SpmmArgs &a, SlabLayout lay, bool main_only, hipStream_t s) {
    constexpr int RPB = 4 * (64 / G);
    if (a.n_short > 0) {
        dim3 grid((unsigned)ceil_div(a.n_short, RPB), (unsigned)lay.n_slabs);
        const bool nt_csr = (a.flags & HRAG_OPT_NT_CSR) != 0, nt_st = (a.flags & HRAG_OPT_NT_STORE) != 0;
        if (nt_csr && nt_st) hipLaunchKernelGGL((ppr_spmm_kernel<G, true, true>), grid, dim3(256), 0, s, a);
        else if (nt_csr) hipLaunchKernelGGL((ppr_spmm_kernel<G, true, false>), grid, dim3(256), 0, s, a);
        else if (nt_st) hipLaunchKernelGGL((ppr_spmm_kernel<G, false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ppr_spmm_kernel<G, false, false>), grid, dim3(256), 0, s, a);
        HRAG_LAUNCH_CHECK();
    }
    if (!main_only && a.n_seg > 0) {
        dim3 grid((unsigned)ceil_div(a.n_seg, 4), (unsigned)lay.n_slabs);
        hipLaunchKernelGGL(ppr_spmm_seg_kernel<G>, grid, dim3(256), 0, s, a);
        HRAG_LAUNCH_CHECK();
        if (a.n_mrow > 0) {
            dim3 rgrid((unsigned)ceil_div((int64_t)a.n_mrow * G, 256), (unsigned)lay.n_slabs);
            hipLaunchKernelGGL(ppr_spmm_reduce_kernel<G>, rgrid, dim3(256), 0, s, a);
            HRAG_LAUNCH_CHECK();
        }
    }
    return HRAG_OK;
}

template <int G>
hrag_status init_dispatch(const SpmmArgs &a, SlabLayout lay, hipStream_t s) {
    if (a.n_rows == 0) return HRAG_OK;
    dim3 grid((unsigned)ceil_div(a.n_rows * G, 256), (unsigned)lay.n_slabs);
    hipLaunchKernelGGL(ppr_init_kernel<G>, grid, dim3(256), 0, s, a);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

template <int G>
hrag_status colsum_dispatch(const float *x, int64_t V, int64_t row_offset, int64_t n_rows,
                            SlabLayout lay, double *partial, hipStream_t s) {
    dim3 grid(kColsumBlocks, (unsigned)lay.n_slabs);
    hipLaunchKernelGGL(colsum_partial_kernel<G>, grid, dim3(256), 0, s, x, V, row_offset, n_rows,
                       partial);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace

#define HRAG_DISPATCH_G(lay, CALL)                                         \
    switch ((lay).bc) {                                                    \
        case 4: return CALL(1);                                            \
        case 8: return CALL(2);                                            \
        case 16: return CALL(4);                                           \
        case 32: return CALL(8);                                           \
        case 64: return CALL(16);                                          \
        default:                                                           \
            set_error("unsupported slab width %d", (int)(lay).bc);         \
            return HRAG_EINVAL;                                            \
    }

hrag_status launch_ppr_spmm(const SpmmArgs &a, SlabLayout lay, bool main_only, hipStream_t s) {
#define CALL(G) spmm_dispatch<G>(a, lay, main_only, s)
    HRAG_DISPATCH_G(lay, CALL)
#undef CALL
}

hrag_status launch_ppr_init(const SpmmArgs &a, SlabLayout lay, hipStream_t s) {
#define CALL(G) init_dispatch<G>(a, lay, s)
    HRAG_DISPATCH_G(lay, CALL)
#undef CALL
}

hrag_status launch_seed_scatter(float *y, int64_t num_vertices, int64_t row_offset, int64_t n_rows,
                                const int32_t *seed_vtx, const float *seed_w,
                                const int32_t *seed_cnt, int32_t max_seeds, int32_t batch,
                                float scale, SlabLayout lay, hipStream_t s) {
    const int total = batch * max_seeds;
    if (total == 0) return HRAG_OK;
    hipLaunchKernelGGL(seed_scatter_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, y,
                       num_vertices, row_offset, n_rows, seed_vtx, seed_w, seed_cnt, max_seeds,
                       batch, scale, lay.bc);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_colsum(const float *x, int64_t num_vertices, int64_t row_offset, int64_t n_rows,
                          int32_t batch, SlabLayout lay, double *partial, double *sums,
                          hipStream_t s) {
    hrag_status st;
#define CALL(G) colsum_dispatch<G>(x, num_vertices, row_offset, n_rows, lay, partial, s)
    st = [&]() -> hrag_status { HRAG_DISPATCH_G(lay, CALL) }();
#undef CALL
    if (st != HRAG_OK) return st;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)batch), dim3(64), 0, s, partial, batch, lay.bc,
                       kColsumBlocks, sums);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
