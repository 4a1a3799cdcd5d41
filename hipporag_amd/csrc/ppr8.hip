// K3b -- PPR power iteration with an fp8 (OCP e4m3) state and an fp32 true residual.
//
// Replaces igraph/PRPACK behind HippoRAG.run_ppr (reference src/hipporag/HippoRAG.py:1736-1743)
// for batches wider than 64 queries: one gather = one 128-byte line = 128 queries.
//
// Why: the sweep is bound by the random row gathers of the state (nnz * B * sizeof(state) bytes,
// ~7 TB/s of 128-byte lines, tools/membench.hip), so bytes per gathered element are the lever.
// An fp8 state alone is useless (2^-4 per rounding); what makes it work is iterative refinement:
// the TRUE residual of the running estimate is kept in fp32 and only the correction that is
// being iterated lives in fp8.  In the degree-scaled variable z = D^-1 x (D = weighted degree):
//
//     z = a At z + b D^-1 v,      At = D^-1 A  row-stochastic   (a = damping, b = 1 - a)
//
// so the max-norm of any residual contracts by `a` per sweep and all fp8 scales are static powers
// of two (v is pre-scaled per query so that max(v/d) is in (1/2, 1]).  z also has the small
// dynamic range an 8-bit float needs: the diffuse part of a PPR vector on an undirected graph is
// proportional to the degree, i.e. flat in z.
//
//   init       c_0 = Q(v/d * 2^7),  R_0 = b v/d (fp32, on the fly)  X_0 = c_0 / 2^7 ~ v/d: the start of
//                                                            the reference iteration (x_0 = v), mass
//                                                            matched per connected component
//   boundary   R <- R + (a At c - c) / cs    = true residual of X + c / cs;   X += c / cs   (mode B)
//              rt = Q(R * cs')                 the next stage's right-hand side and first iterate
//   stage      c <- Q(a At c + rt)             m - 1 sweeps                              (mode C)
//   final      z = sum_s c_s / cs_s + R        the last boundary, fused with the combine   (mode F)
//              x = d * z                       (X + R = one more exact sweep, free)
//
// Every gather sweep is one of the `ppr_iters` iterations: X_0 = v/d costs none, stage s of m_s
// sweeps advances the exact iteration by m_s applications of (a At . + b v/d) up to the fp8
// rounding of that stage, which the NEXT boundary measures exactly and hands to the next stage.
// Stages: 1, 2, 2, 3, 3, 3, ... (short first: that is where the residual is large).  Measured
// (tools/exp_fp8_final.py): max relative error over all passages 2.6e-7 at 20 sweeps on the
// benchmark graph -- the level of the fp32 iteration -- and within 3x of the plain 20-sweep
// iteration on slowly mixing graphs (ring, stars) whose own truncation error is the larger term.
//
// Matrix: the SELL-8 form of ppr16.hip with the row-normalised values At (engine.hip builds it
// from P and the weighted degrees: at_ij = p_ij d_j / d_i).  Long rows: segments + fixed-order
// reduce, no atomics, bit-reproducible.  All arithmetic is fp32; fp8 -> fp32 is exact.
#include "common.h"

namespace hrag {
namespace {

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));

constexpr float kE4m3Max = 448.f;

template <int K>
__device__ __forceinline__ int bcast8(int v) {
    // ds_swizzle bit mode inside each 8-lane group: src = (lane & 0x18) | K
    return __builtin_amdgcn_ds_swizzle(v, (K << 5) | 0x18);
}

// 16 fp8 (one dwordx4) -> 8 float pairs; element j of the line is pair j / 2, component j % 2
__device__ __forceinline__ void decode16(const v4i_t &x, f32x2_t (&f)[8]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        f[2 * d] = __builtin_amdgcn_cvt_pk_f32_fp8(x[d], false);
        f[2 * d + 1] = __builtin_amdgcn_cvt_pk_f32_fp8(x[d], true);
    }
}

// round-to-nearest-even to e4m3, saturating (the clamp keeps the conversion away from its NaN)
__device__ __forceinline__ v4i_t encode16(const f32x2_t (&f)[8]) {
    v4i_t o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float a0 = __builtin_amdgcn_fmed3f(f[2 * d].x, -kE4m3Max, kE4m3Max);
        const float a1 = __builtin_amdgcn_fmed3f(f[2 * d].y, -kE4m3Max, kE4m3Max);
        const float a2 = __builtin_amdgcn_fmed3f(f[2 * d + 1].x, -kE4m3Max, kE4m3Max);
        const float a3 = __builtin_amdgcn_fmed3f(f[2 * d + 1].y, -kE4m3Max, kE4m3Max);
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, p, true);
        o[d] = p;
    }
    return o;
}

// acc += float(x) * w : 8 conversions + 8 packed FMAs per 16 gathered elements
__device__ __forceinline__ void fma16(f32x2_t (&acc)[8], float w, const v4i_t &x) {
    const f32x2_t w2 = {w, w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(x[d], false);
        const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(x[d], true);
        acc[2 * d] = __builtin_elementwise_fma(lo, w2, acc[2 * d]);
        acc[2 * d + 1] = __builtin_elementwise_fma(hi, w2, acc[2 * d + 1]);
    }
}

// One step = 8 gathers per lane.  All 8 loads are issued before the first conversion (the
// sched_barrier keeps hipcc from interleaving them with the FMAs, which would leave only 2-3
// lines in flight per wavefront).
template <int K>
struct Gather8 {
    __device__ __forceinline__ static void load(v4i_t (&xv)[8], float (&wk)[8], int c, int wbits,
                                                const char *xs, unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        wk[K] = __int_as_float(bcast8<K>(wbits));
        // 128 bytes per vertex, 16 per lane; V * 128 < 2^32 is checked at engine creation
        xv[K] = *reinterpret_cast<const v4i_t *>(xs + (size_t)(ck * 128u + lane_off));
        if constexpr (K + 1 < 8) Gather8<K + 1>::load(xv, wk, c, wbits, xs, lane_off);
    }
};
__device__ __forceinline__ void gather_step(f32x2_t (&acc)[8], int c, int wbits, const char *xs,
                                            unsigned lane_off) {
    v4i_t xv[8];
    float wk[8];
    Gather8<0>::load(xv, wk, c, wbits, xs, lane_off);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) fma16(acc, wk[k], xv[k]);
}

// (col, val) pairs through a buffer descriptor (hipcc keeps raw buffer loads where they are written:
// see ld_pair in ppr16.hip)
// plain (cacheable) loads: the workgroup of the next slab re-reads the same blocks from L2, see below
__device__ __forceinline__ int2 ld_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
    return make_int2(v.x, v.y);
}

// fp32 rows of the internal arrays (R, partial sums) are stored LANE-INTERLEAVED: the 128 floats of a
// row are 32 float4s, and float4 number 8 * i + gl holds queries 16 * gl + 4 * i .. + 3 (lane gl's
// i-th quad).  One store instruction of an 8-lane group then writes one whole 128-byte line; with
// the natural order (lane gl owning 64 contiguous bytes) every instruction left 16-byte pieces at a
// 64-byte stride and the streams ran at 1 TB/s.  Only ppr8.hip reads or writes these arrays.
__device__ __forceinline__ void ld16i(const float *row, int gl, f32x2_t (&f)[8]) {
    const f32x4_t *p4 = reinterpret_cast<const f32x4_t *>(row) + gl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4_t v = __builtin_nontemporal_load(p4 + 8 * i);
        f[2 * i] = f32x2_t{v.x, v.y};
        f[2 * i + 1] = f32x2_t{v.z, v.w};
    }
}
__device__ __forceinline__ void st16i(float *row, int gl, const f32x2_t (&f)[8]) {
    f32x4_t *p4 = reinterpret_cast<f32x4_t *>(row) + gl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4_t v = {f[2 * i].x, f[2 * i].y, f[2 * i + 1].x, f[2 * i + 1].y};
        __builtin_nontemporal_store(v, p4 + 8 * i);
    }
}
// natural order (caller-facing arrays): lane owns 16 consecutive floats
__device__ __forceinline__ void st16f(float *p, const f32x2_t (&f)[8]) {
    f32x4_t *p4 = reinterpret_cast<f32x4_t *>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4_t v = {f[2 * i].x, f[2 * i].y, f[2 * i + 1].x, f[2 * i + 1].y};
        p4[i] = v;
    }
}

// z_v = v / d for one vertex row: v comes from the teleport rows of the fp16 path's layout
// (fp32 [n_slabs64][tele_rows][64] + row_slot), already scaled per query.
__device__ __forceinline__ void load_zv(const float *__restrict__ tele, int64_t tele_rows,
                                        const int32_t *__restrict__ row_slot, const float *__restrict__ deg,
                                        int32_t n_slabs64, int slab, int64_t row, int gl, f32x2_t (&z)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = f32x2_t{0.f, 0.f};
    const int slot = row_slot[row];
    const int slab64 = 2 * slab + (gl >> 2);
    if (slot >= 0 && slab64 < n_slabs64) {
        const f32x4_t *tp = reinterpret_cast<const f32x4_t *>(
            tele + ((size_t)slab64 * tele_rows + (size_t)slot) * 64 + (size_t)(gl & 3) * 16);
        const float invd = __fdiv_rn(1.0f, deg[row]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4_t v = tp[i];
            z[2 * i] = f32x2_t{v.x, v.y} * invd;
            z[2 * i + 1] = f32x2_t{v.z, v.w} * invd;
        }
    }
}

// Finish one output row: lane gl of its group owns queries 16*gl .. 16*gl+15 of the 128-wide slab.
// Mode F returns the row's x (16 queries of this lane) in xs for the caller's column sums.
template <int MODE>
__device__ __forceinline__ void finish_row(const Ppr8Args &a, int slab, int row, int gl,
                                           const f32x2_t (&acc)[8], f32x2_t (&xs)[8]) {
    const size_t off = ((size_t)slab * a.num_vertices + (size_t)row) * 128 + (size_t)gl * 16;
    f32x2_t out[8];
    if constexpr (MODE == kP8ModeC) {
        f32x2_t r[8];
        decode16(__builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(a.rt + off)), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = __builtin_elementwise_fma(acc[j], f32x2_t{a.alpha, a.alpha}, r[j]);
        __builtin_nontemporal_store(encode16(out), reinterpret_cast<v4i_t *>(a.y + off));
    } else {
        f32x2_t c[8], rin[8];
        decode16(__builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(a.x + off)), c);
        float *rrow = a.R + ((size_t)slab * a.num_vertices + (size_t)row) * 128;
        if constexpr (MODE == kP8ModeB0) {
            load_zv(a.tele, a.tele_rows, a.row_slot, a.deg, a.n_slabs64, slab, row, gl, rin);   // R_in = b v/d
#pragma unroll
            for (int j = 0; j < 8; ++j) rin[j] *= a.beta;
        } else {
            ld16i(rrow, gl, rin);
        }
        const f32x2_t al = {a.alpha, a.alpha}, inv = {a.inv_cs, a.inv_cs};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2_t t = __builtin_elementwise_fma(acc[j], al, -c[j]);   // a (At c) - c, one rounding
            out[j] = __builtin_elementwise_fma(t, inv, rin[j]);              // inv is a power of two: exact
        }
        if constexpr (MODE == kP8ModeB || MODE == kP8ModeB0) {
            st16i(rrow, gl, out);
            f32x2_t q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = out[j] * a.cs_next;
            __builtin_nontemporal_store(encode16(q), reinterpret_cast<v4i_t *>(a.y + off));
        } else {   // kP8ModeF: z = R' + sum_s c_s / cs_s (earliest stage first), x = d z
            f32x2_t z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = f32x2_t{0.f, 0.f};
            for (int s = 0; s < a.n_stage; ++s) {
                f32x2_t cs[8];
                decode16(__builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(a.stage[s] + off)), cs);
                const f32x2_t si = {a.stage_inv[s], a.stage_inv[s]};
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = __builtin_elementwise_fma(cs[j], si, z[j]);
            }
            const float dg = a.deg[row];
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[j] = (z[j] + out[j]) * dg;
            // only the passage rows are needed downstream (HippoRAG.py:1745), in passage order:
            // xp is [n_slabs64][Np][64] fp32, the layout slab_to_rows reads without a gather
            const int slot = a.row_slot[row];
            const int slab64 = 2 * slab + (gl >> 2);
            if (slot >= 0 && slot < a.n_passages && slab64 < a.n_slabs64)
                st16f(a.out + ((size_t)slab64 * a.n_passages + (size_t)slot) * 64 + (size_t)(gl & 3) * 16, xs);
        }
    }
}

// 1-D grid, XCD-aware: workgroups are dealt to the 8 XCDs round-robin, so ids 8 apart run back to back
// on the same XCD.  They are given the SAME chunk group for consecutive slabs: the second reader of
// a (col, val) block then finds it in that XCD's L2 (measured: C sweep 0.822 -> 0.802 ms at cfg 3;
// with non-temporal pair loads the remap alone changes nothing).
template <int MODE>
__global__ __launch_bounds__(256, 4) void ppr8_kernel(const Ppr8Args a) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7, grp = lane >> 3;
    const int id = blockIdx.x, ns = a.n_slabs;
    const int slab = (id >> 3) % ns;
    const int cg = (id / (8 * ns)) * 8 + (id & 7);
    const int chunk = __builtin_amdgcn_readfirstlane(cg * 4 + (threadIdx.x >> 6));
    if (chunk >= a.n_chunks) return;
    const int2 meta = a.chunk_meta[chunk];  // (first step, number of steps)
    const int n_steps = meta.y;
    const char *xs = reinterpret_cast<const char *>(a.x + (size_t)slab * a.num_vertices * 128);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int2 *>(a.pairs), 0, (int)a.pairs_bytes, 0x00020000);
    const unsigned pbase = (unsigned)meta.x * 512u;   // scalar: first byte of this chunk's pairs
    const unsigned poff = (unsigned)lane * 8u;
    const unsigned lane_off = (unsigned)gl * 16u;
    f32x2_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x2_t{0.f, 0.f};
    // pair stream read two steps ahead, unconditionally (the array carries the padding)
    int2 p0 = ld_pair(prs, poff, pbase);
    int2 p1 = ld_pair(prs, poff + 512u, pbase);
    for (int s = 0; s < n_steps; ++s) {
        const int2 p2 = ld_pair(prs, poff + (unsigned)(s + 2) * 512u, pbase);
        gather_step(acc, p0.x, p0.y, xs, lane_off);
        p0 = p1;
        p1 = p2;
    }
    const int tgt = a.vrow[chunk * 8 + grp];
    f32x2_t xr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[j] = f32x2_t{0.f, 0.f};
    if (tgt >= 0) {
        finish_row<MODE>(a, slab, tgt, gl, acc, xr);
    } else if (tgt != kVrowNone) {
        st16i(a.partial + ((size_t)slab * a.n_partial + (size_t)(-(tgt + 1))) * 128, gl, acc);
    }
    if constexpr (MODE == kP8ModeF) {
        // column sums of x over this wavefront's 8 rows (fixed order), one partial row per chunk;
        // ppr8_colsum_kernel adds the partial rows up in double
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xr[j].x += __shfl_xor(xr[j].x, o, 64);
                xr[j].y += __shfl_xor(xr[j].y, o, 64);
            }
        if (grp == 0) st16i(a.csum + ((size_t)slab * a.n_csum + (size_t)chunk) * 128, gl, xr);
    }
}

// One wavefront per long row: the 8 lane groups stride over the row's partial sums, then the 8
// group totals are added with xor-shuffles -- a fixed summation order.
template <int MODE>
__global__ __launch_bounds__(256) void ppr8_reduce_kernel(const Ppr8Args a) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7, grp = lane >> 3;
    const int slab = blockIdx.y;
    const int m = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (m >= a.n_lrow) return;
    const int first = a.lrow_first[m], cnt = a.lrow_cnt[m];
    const float *base = a.partial + ((size_t)slab * a.n_partial + (size_t)first) * 128;
    f32x2_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x2_t{0.f, 0.f};
    for (int s = grp; s < cnt; s += 8) {
        f32x2_t v[8];
        ld16i(base + (size_t)s * 128, gl, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j].x += __shfl_xor(acc[j].x, o, 64);
            acc[j].y += __shfl_xor(acc[j].y, o, 64);
        }
    if (grp == 0) {
        f32x2_t xs[8];
        finish_row<MODE>(a, slab, a.lrow_row[m], gl, acc, xs);
        if constexpr (MODE == kP8ModeF)   // a long row is its own partial row (after the chunk rows)
            st16i(a.csum + ((size_t)slab * a.n_csum + (size_t)a.n_chunks + (size_t)m) * 128, gl, xs);
    }
}

// Column sums of x = d z from the partial rows mode F wrote (lane-interleaved 128-float rows):
// pass 1 (grid kP8ColsumBlocks x n_slabs): block b sums rows b, b + kP8ColsumBlocks, ... in double;
// pass 2: sums[q] = sum over blocks, q -> (slab, interleaved position).  Fixed order, no atomics.
constexpr int kP8ColsumBlocks = 128;
__global__ __launch_bounds__(256) void ppr8_colsum_kernel(const float *__restrict__ csum, int32_t n_csum,
                                                          double *__restrict__ partial) {
    __shared__ double red[8][128];
    const int tid = threadIdx.x, f4 = tid & 31, rl = tid >> 5;   // 32 float4 per row, 8 rows per pass
    const int slab = blockIdx.y;
    const f32x4_t *base = reinterpret_cast<const f32x4_t *>(csum + (size_t)slab * n_csum * 128);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t r = (int64_t)blockIdx.x * 8 + rl; r < n_csum; r += (int64_t)kP8ColsumBlocks * 8) {
        const f32x4_t v = base[r * 32 + f4];
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    red[rl][f4 * 4 + 0] = s0; red[rl][f4 * 4 + 1] = s1; red[rl][f4 * 4 + 2] = s2; red[rl][f4 * 4 + 3] = s3;
    __syncthreads();
    if (tid < 128) {
        double t = 0;
        for (int i = 0; i < 8; ++i) t += red[i][tid];
        partial[((size_t)slab * kP8ColsumBlocks + blockIdx.x) * 128 + tid] = t;
    }
}
__global__ void ppr8_colsum_final_kernel(const double *__restrict__ partial, int32_t batch, double *sums) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    const int slab = q >> 7, qq = q & 127;
    const int gl = qq >> 4, i = (qq >> 2) & 3, k = qq & 3;   // query qq sits in float4 8 i + gl, component k
    const int phys = (8 * i + gl) * 4 + k;
    double s = 0;
    for (int b = 0; b < kP8ColsumBlocks; ++b) s += partial[((size_t)slab * kP8ColsumBlocks + b) * 128 + phys];
    sums[q] = s;
}

// c_0 = Q(v/d * c0_scale) for every vertex row of every slab (R_0 = b v/d is formed on the fly by the
// first boundary sweep, mode B0).
__global__ __launch_bounds__(256) void ppr8_init_kernel(const float *__restrict__ tele, int64_t tele_rows,
                                                        const int32_t *__restrict__ row_slot,
                                                        const float *__restrict__ deg, int64_t num_vertices,
                                                        int32_t n_slabs64, float c0_scale,
                                                        uint8_t *__restrict__ c0) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t >> 3;
    const int gl = (int)(t & 7);
    const int slab = blockIdx.y;
    if (row >= num_vertices) return;
    f32x2_t z[8], q[8];
    load_zv(tele, tele_rows, row_slot, deg, n_slabs64, slab, row, gl, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = z[j] * c0_scale;
    const size_t off = ((size_t)slab * num_vertices + (size_t)row) * 128 + (size_t)gl * 16;
    *reinterpret_cast<v4i_t *>(c0 + off) = encode16(q);
}

// Per-query power-of-two scale s_q with  max_i v_i / d_i * s_q  in (1/2, 1]:
//   bound_q = passage_weight * max_p minmax(score_qp) / d_p  +  max_j seed_w / d(seed_j)
// (the sum of the two maxima covers a seed that is also a passage vertex).
// Pass 1: grid (kScaleSplit, B) partial maxima, combined with an integer atomicMax on the bits of the
// non-negative floats (order-preserving, so the result does not depend on the arrival order).
constexpr int kScaleSplit = 16;
__global__ __launch_bounds__(256) void ppr8_zmax_kernel(const float *__restrict__ scores, int64_t ld,
                                                        int64_t n_passages, const float *__restrict__ mn,
                                                        const float *__restrict__ mx,
                                                        const float *__restrict__ pinvdeg,
                                                        const int32_t *__restrict__ flags, int32_t *zmax_bits) {
    __shared__ float red[256];
    const int q = blockIdx.y, tid = threadIdx.x;
    float best = 0.f;
    if (!(flags[q] & 1)) {
        const float lo = mn[q], range = mx[q] - mn[q];
        const float *row = scores + (size_t)q * ld;
        for (int64_t p = (int64_t)blockIdx.x * 256 + tid; p < n_passages; p += (int64_t)kScaleSplit * 256) {
            const float nrm = range == 0.f ? 1.f : __fdiv_rn(row[p] - lo, range);
            best = fmaxf(best, nrm * pinvdeg[p]);
        }
    }
    red[tid] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) atomicMax(&zmax_bits[q], __float_as_int(fmaxf(red[0], 0.f)));
}

__global__ void ppr8_scale_kernel(const int32_t *__restrict__ zmax_bits, float passage_weight,
                                  const int32_t *__restrict__ seed_vtx, const float *__restrict__ seed_w,
                                  const int32_t *__restrict__ seed_cnt, const float *__restrict__ deg,
                                  int64_t num_vertices, const int32_t *__restrict__ flags, int32_t batch,
                                  float *qscale) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    float bound = 0.f;
    if (!(flags[q] & 1)) {
        bound = fmaxf(passage_weight, 0.f) * __int_as_float(zmax_bits[q]);
        float sb = 0.f;
        for (int j = 0; j < seed_cnt[q]; ++j) {
            const int64_t v = seed_vtx[q * kMaxSeeds + j];
            if (v >= 0 && v < num_vertices) sb = fmaxf(sb, __fdiv_rn(fmaxf(seed_w[q * kMaxSeeds + j], 0.f), deg[v]));
        }
        bound += sb;
    }
    float s = 1.f;
    if (bound > 0.f && bound < 3e38f) {
        int ex;
        const float m = frexpf(bound, &ex);    // bound = m * 2^ex, m in [0.5, 1)
        if (m == 0.5f) ex -= 1;                // exact power of two: bound * 2^-(ex-1) = 1
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        s = ldexpf(1.f, -ex);
    }
    qscale[q] = s;
}

template <int MODE>
hrag_status sweep_mode(const Ppr8Args &a, int n_slabs, bool main_only, hipStream_t s) {
    if (a.n_chunks > 0) {
        const unsigned ncg = (unsigned)ceil_div(a.n_chunks, 4);
        Ppr8Args b = a;
        b.n_slabs = n_slabs;
        // (86 VGPRs = 5 wavefronts per SIMD; 6 measured the same: the sweep is bandwidth-bound)
        hipLaunchKernelGGL(ppr8_kernel<MODE>, dim3((unsigned)round_up(ncg, 8) * (unsigned)n_slabs), dim3(256), 0, s, b);
        HRAG_LAUNCH_CHECK();
    }
    if (!main_only && a.n_lrow > 0) {
        dim3 grid((unsigned)ceil_div(a.n_lrow, 4), (unsigned)n_slabs);
        hipLaunchKernelGGL(ppr8_reduce_kernel<MODE>, grid, dim3(256), 0, s, a);
        HRAG_LAUNCH_CHECK();
    }
    return HRAG_OK;
}

}  // namespace

hrag_status launch_ppr8_sweep(const Ppr8Args &a, int mode, int n_slabs, bool main_only, hipStream_t s) {
    switch (mode) {
        case kP8ModeC: return sweep_mode<kP8ModeC>(a, n_slabs, main_only, s);
        case kP8ModeB: return sweep_mode<kP8ModeB>(a, n_slabs, main_only, s);
        case kP8ModeB0: return sweep_mode<kP8ModeB0>(a, n_slabs, main_only, s);
        case kP8ModeF: return sweep_mode<kP8ModeF>(a, n_slabs, main_only, s);
        default: set_error("bad ppr8 mode %d", mode); return HRAG_EINVAL;
    }
}

// partial: n_slabs * 128 * 128 doubles
hrag_status launch_ppr8_colsum(const float *csum, int32_t n_csum, int n_slabs, int32_t batch, double *partial,
                               double *sums, hipStream_t s) {
    hipLaunchKernelGGL(ppr8_colsum_kernel, dim3(kP8ColsumBlocks, (unsigned)n_slabs), dim3(256), 0, s, csum,
                       n_csum, partial);
    HRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ppr8_colsum_final_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, s, partial,
                       batch, sums);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_init(const float *tele, int64_t tele_rows, const int32_t *row_slot, const float *deg,
                             int64_t num_vertices, int n_slabs, int n_slabs64, float c0_scale, uint8_t *c0,
                             hipStream_t s) {
    dim3 grid((unsigned)ceil_div(num_vertices * 8, 256), (unsigned)n_slabs);
    hipLaunchKernelGGL(ppr8_init_kernel, grid, dim3(256), 0, s, tele, tele_rows, row_slot, deg, num_vertices,
                       n_slabs64, c0_scale, c0);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_scale(const float *scores, int64_t ld, int64_t n_passages, const float *mn,
                              const float *mx, float passage_weight, const float *pinvdeg,
                              const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt,
                              const float *deg, int64_t num_vertices, const int32_t *flags, int32_t batch,
                              int32_t *zmax_bits, float *qscale, hipStream_t s) {
    HRAG_HIP_TRY(hipMemsetAsync(zmax_bits, 0, (size_t)batch * sizeof(int32_t), s));
    if (n_passages > 0) {
        hipLaunchKernelGGL(ppr8_zmax_kernel, dim3(kScaleSplit, (unsigned)batch), dim3(256), 0, s, scores, ld,
                           n_passages, mn, mx, pinvdeg, flags, zmax_bits);
        HRAG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ppr8_scale_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s, zmax_bits,
                       passage_weight, seed_vtx, seed_w, seed_cnt, deg, num_vertices, flags, batch, qscale);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
