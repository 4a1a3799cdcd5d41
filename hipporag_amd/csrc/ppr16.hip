// K3h -- PPR power iteration with a two-stage fp16 state ("hi + correction"), fp32 arithmetic.
//
// Replaces igraph/PRPACK behind HippoRAG.run_ppr (reference src/hipporag/HippoRAG.py:1736-1743)
// for batches of 9..64 queries (and wider ones when the fp8 path of ppr8.hip is unavailable).
//
// Why: the sweep  y = alpha P x + (1 - alpha) v  is bound by the random row gathers of x
// (nnz * B * sizeof(state) bytes per sweep, ~7 TB/s of 128-byte lines whether they come from HBM or
// the Infinity Cache -- tools/membench.hip), so the only lever is bytes per gathered element.
// A plain fp16 state cannot meet the 1e-5 parity bar (2^-11 per rounding).  This file keeps the
// state in fp16 AND follows the fp32 trajectory to ~4e-7 (measured):
//
//   sweeps 1..K1      h_{k+1} = f16(alpha P h_k + beta v)                          (mode H)
//   sweep  K1+1       r       = (alpha P h + beta v) - h   in fp32, stored f16(r * cs)   (mode R)
//                     -- this IS sweep K1+1 of the iteration started at x_K1 = h: x_{K1+1} = h + r
//   sweeps K1+2..K    c_{k+1} = f16(alpha P c_k + r),  c_{K1+1} = r               (mode C)
//   sweep  K          the last C sweep runs over the passage rows only (a second SELL-8 matrix) and writes
//                     x_K = h + c_K / cs in fp32 at the passages, in passage order           (mode F)
//                     -- nothing else is read afterwards (HippoRAG.py:1745); the normalising mass of x_K is the
//                     closed form of ppr8.hip / ppr_sv.hip (sum(v) and the mass on isolated vertices)
//   sweep  1          gathers only the columns where h_0 = f16(v) is non-zero (passages, seeds: column bitmap)
//
// x_k = h + c_k obeys exactly the recurrence of the fp32 iteration from x_K1 = h, so the truncation
// error after K sweeps is that of K ordinary sweeps; the rounding of h is removed by r (computed
// in fp32 from the exact fp16 values) and the rounding of c is 2^-11 relative to |c| ~ 2^-K1 |x|.
// All products are exact (fp16 x fp32 -> v_fma_mix_f32), sums are fp32 in a fixed order.
//
// Storage: x is [n_slabs][V][64] fp16 -- one gather = one 128-byte line = 64 queries.
// Matrix: SELL-8 ("sliced ELLPACK", slice = the 8 rows of one wavefront): rows sorted by length,
// rows longer than 64 .. 512 entries (engine.hip sell8_seg_len) cut into segments ("virtual rows" whose partial sums are
// combined in a fixed order by the wavefront of the segment that arrives last -- no atomics on the data, one
// arrival counter per row; bit-reproducible, no second kernel), 8 virtual
// rows per wavefront, entries stored step-major as (col, val) pairs so that a wavefront's CSR
// read is ONE coalesced 512-byte load per 8-gather step and every fetched byte is used once.
#include <hip/hip_fp16.h>

#include "common.h"

namespace hrag {
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr float kHalfMax = 65504.f;

template <int K>
__device__ __forceinline__ int bcast8(int v) {
    // ds_swizzle bit mode inside each 8-lane group: src = (lane & 0x18) | K
    return __builtin_amdgcn_ds_swizzle(v, (K << 5) | 0x18);
}

typedef int v4i_t __attribute__((ext_vector_type(4)));

// acc[j] += float(x[j]) * w with v_fma_mix_f32 (fp16 operand converted inside the FMA: exact product,
// one rounding).  Written as asm because hipcc otherwise emits 8 v_cvt_f32_f16 + 4 v_pk_fma_f32 per
// gather, whose temporaries cost a wavefront of occupancy (84 -> 64 VGPRs).
__device__ __forceinline__ void fma8(float (&acc)[8], float w, const v4i_t &x) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * d]) : "v"(x[d]), "v"(w));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * d + 1]) : "v"(x[d]), "v"(w));
    }
}

template <int K>
struct Gather8 {
    __device__ __forceinline__ static void run(float (&acc)[8], int c, int wbits, const char *xs,
                                               unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        const float wk = __int_as_float(bcast8<K>(wbits));
        // 128 bytes per vertex, 16 per lane; V * 128 < 2^32 is checked at engine creation
        const v4i_t xv = *reinterpret_cast<const v4i_t *>(xs + (size_t)(ck * 128u + lane_off));
        fma8(acc, wk, xv);
        if constexpr (K + 1 < 8) Gather8<K + 1>::run(acc, c, wbits, xs, lane_off);
    }
};

// First sweep: h_0 = f16(v) is zero outside the passage / seed vertices; the gather of a column whose bit is clear in
// the column bitmap is not issued (its lanes are masked off for the load), ~6 % of the entries remain.
template <int K>
struct Gather8M {
    __device__ __forceinline__ static void load(v4i_t (&xv)[8], float (&wk)[8], int c, int wbits, int on, const char *xs,
                                                unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        wk[K] = __int_as_float(bcast8<K>(wbits));
        v4i_t v = {0, 0, 0, 0};
        if (bcast8<K>(on)) v = *reinterpret_cast<const v4i_t *>(xs + (size_t)(ck * 128u + lane_off));
        xv[K] = v;
        if constexpr (K + 1 < 8) Gather8M<K + 1>::load(xv, wk, c, wbits, on, xs, lane_off);
    }
};
// all (predicated) loads of a step are issued before the first FMA
__device__ __forceinline__ void gather_step_masked(float (&acc)[8], int c, int wbits, int on, const char *xs,
                                                   unsigned lane_off) {
    v4i_t xv[8];
    float wk[8];
    Gather8M<0>::load(xv, wk, c, wbits, on, xs, lane_off);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) fma8(acc, wk[k], xv[k]);
}

typedef int v2i_t __attribute__((ext_vector_type(2)));

// (col, val) pairs are read through a buffer descriptor: hipcc keeps raw buffer loads where they are
// written (a plain or __builtin_nontemporal_load of this loop-invariant stream is sunk back to its
// first use, which turns the read-ahead into a stall per step) and the nt bit rides in `aux`.
template <bool NT>
__device__ __forceinline__ int2 ld_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, NT ? 2 : 0);
    return make_int2(v.x, v.y);
}

// 16-byte store / load with sc1 (aux bit 4): write-through to memory / served past the CU's L1 -- the forms that
// make data written by one workgroup readable by another INSIDE a launch (per-XCD L2s are not coherent)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, float a, float b, float c, float d) {
    const v4u_t v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 16);
}
__device__ __forceinline__ f32x4_t ld_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 16);
    return f32x4_t{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}

__device__ __forceinline__ float clamp_half(float v) { return fminf(fmaxf(v, -kHalfMax), kHalfMax); }

// What a row's finish reads besides the accumulated sums: its teleport row (modes H, R), its own h (R, F), the
// right-hand side r (C, F) and its slot.  Loaded by the wavefront BEFORE the gather loop (the row is known from the
// start), so that the vrow -> row_slot -> tele chain of dependent loads overlaps with the gathers instead of
// following them: on small graphs (cfg 2: 12.5k wavefronts, ~3 gather steps each) that chain was a third of the sweep.
struct RowIn {
    f32x4_t t0, t1;     // H, R: teleport row (zero without one)
    half8_t h;          // R: a.x own row; F: a.hfin own row
    half8_t r;          // C, F: a.aux own row
    half8_t p;          // H, C with a.prev: the iterate before x, own row (Chebyshev step)
    int slot;           // F: passage number
};
template <int MODE>
__device__ __forceinline__ void load_row_in(const Ppr16Args &a, int slab, int row, int gl, RowIn &in) {
    const size_t state_off = ((size_t)slab * a.num_vertices + (size_t)row) * 64 + (size_t)gl * 8;
    in.t0 = in.t1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
    in.slot = -1;
    if constexpr (MODE == kPprModeH || MODE == kPprModeC) {
        in.p = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (a.prev) in.p = *reinterpret_cast<const half8_t *>(a.prev + state_off);   // wave-uniform branch
    }
    if constexpr (MODE == kPprModeH || MODE == kPprModeR) {
        const int slot = a.row_slot[row];
        if (slot >= 0) {
            const f32x4_t *tp = reinterpret_cast<const f32x4_t *>(
                a.tele + ((size_t)slab * a.tele_rows + (size_t)slot) * 64 + (size_t)gl * 8);
            in.t0 = tp[0];
            in.t1 = tp[1];
        }
        if constexpr (MODE == kPprModeR) in.h = *reinterpret_cast<const half8_t *>(a.x + state_off);
    } else {
        in.r = *reinterpret_cast<const half8_t *>(a.aux + state_off);
        if constexpr (MODE == kPprModeF) {
            in.h = *reinterpret_cast<const half8_t *>(a.hfin + state_off);
            in.slot = a.row_slot[row];
        }
    }
}

// Finish one output row: lane gl of its group owns queries 8*gl .. 8*gl+7 of the slab.
// er (mode F with a.est): the relative size of this (last) sweep's update of the lane's 8 queries
template <int MODE, bool NT_ST = false>
__device__ __forceinline__ void finish_row(const Ppr16Args &a, int slab, int row, int gl,
                                           const float (&acc)[8], const RowIn &in, float (&er)[8]) {
    float out[8];
    const size_t state_off = ((size_t)slab * a.num_vertices + (size_t)row) * 64 + (size_t)gl * 8;
    if constexpr (MODE == kPprModeH || MODE == kPprModeR) {
        const float t[8] = {in.t0.x, in.t0.y, in.t0.z, in.t0.w, in.t1.x, in.t1.y, in.t1.z, in.t1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = fmaf(a.alpha, acc[j], a.beta * t[j]);
        if constexpr (MODE == kPprModeR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = (out[j] - (float)in.h[j]) * a.cscale;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = fmaf(a.alpha, acc[j], (float)in.r[j]);
        if constexpr (MODE == kPprModeF) {
            // last correction sweep, passage rows only: x = h + c / cscale in fp32, passage order
            // [n_slabs][p_rows][64] -- the layout slab_to_rows reads without a gather
            const float ics = 1.0f / a.cscale;
            f32x4_t *op = reinterpret_cast<f32x4_t *>(
                a.out + ((size_t)slab * a.p_rows + (size_t)in.slot) * 64 + (size_t)gl * 8);
            const f32x4_t v0 = {fmaf(out[0], ics, (float)in.h[0]), fmaf(out[1], ics, (float)in.h[1]),
                                fmaf(out[2], ics, (float)in.h[2]), fmaf(out[3], ics, (float)in.h[3])};
            const f32x4_t v1 = {fmaf(out[4], ics, (float)in.h[4]), fmaf(out[5], ics, (float)in.h[5]),
                                fmaf(out[6], ics, (float)in.h[6]), fmaf(out[7], ics, (float)in.h[7])};
            op[0] = v0;
            op[1] = v1;
            if (a.est) {   // the relative size of this (last) sweep's update of the passage score
                const half8_t cold = *reinterpret_cast<const half8_t *>(a.x + state_off);
                const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    er[j] = xs[j] > 0.f ? fabsf(out[j] - (float)cold[j]) * ics / xs[j] : 0.f;
            }
            return;
        }
    }
    if constexpr (MODE == kPprModeH || MODE == kPprModeC) {
        if (a.omega != 1.f) {   // wave-uniform: Chebyshev step (HRAG_OPT_ACCEL), omega (plain result - prev) + prev
            const float om = a.omega, om1 = 1.f - a.omega;
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = fmaf(om, out[j], om1 * (float)in.p[j]);
        }
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)clamp_half(out[j]);
    if constexpr (NT_ST) {
        __builtin_nontemporal_store(o, reinterpret_cast<half8_t *>(a.y + state_off));
    } else {
        *reinterpret_cast<half8_t *>(a.y + state_off) = o;
    }
}

template <int MODE, bool NT, bool NT_ST, bool MASK = false>
__global__ __launch_bounds__(256, 6) void ppr16_kernel(const Ppr16Args a) {
    if (a.gate && *a.gate != a.gate_want) return;   // a conditional step the device decided not to run
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7, grp = lane >> 3;
    const int slab = blockIdx.y;
    const int chunk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (chunk >= a.n_chunks) return;
    const int2 meta = a.chunk_meta[chunk];  // (first step, number of steps)
    const int n_steps = meta.y;
    const char *xs = reinterpret_cast<const char *>(a.x + (size_t)slab * a.num_vertices * 64);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int2 *>(a.pairs), 0, (int)a.pairs_bytes, 0x00020000);
    const unsigned pbase = (unsigned)meta.x * 512u;   // scalar: first byte of this chunk's pairs
    const unsigned poff = (unsigned)lane * 8u;
    const unsigned lane_off = (unsigned)gl * 16u;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tgt = a.vrow[chunk * 8 + grp];
    RowIn in;
    if (tgt >= 0) load_row_in<MODE>(a, slab, tgt, gl, in);
    // the pair stream is read two steps ahead, unconditionally (the array carries two steps of
    // padding), so that the loop body is branch-free and the compiler can wait with vmcnt(N > 0)
    int2 p0 = ld_pair<NT>(prs, poff, pbase);
    int2 p1 = ld_pair<NT>(prs, poff + 512u, pbase);
    if constexpr (MASK) {
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(a.colmask), 0, (int)a.colmask_bytes, 0x00020000);
        int m0 = __builtin_amdgcn_raw_buffer_load_b32(mrs, ((unsigned)p0.x >> 5) * 4u, 0, 0);
        for (int s = 0; s < n_steps; ++s) {
            const int2 p2 = ld_pair<NT>(prs, poff + (unsigned)(s + 2) * 512u, pbase);
            const int m1 = __builtin_amdgcn_raw_buffer_load_b32(mrs, ((unsigned)p1.x >> 5) * 4u, 0, 0);
            gather_step_masked(acc, p0.x, p0.y, (m0 >> (p0.x & 31)) & 1, xs, lane_off);
            p0 = p1;
            p1 = p2;
            m0 = m1;
        }
    } else {
        for (int s = 0; s < n_steps; ++s) {
            const int2 p2 = ld_pair<NT>(prs, poff + (unsigned)(s + 2) * 512u, pbase);
            Gather8<0>::run(acc, p0.x, p0.y, xs, lane_off);
            p0 = p1;
            p1 = p2;
        }
    }
    const bool seg = tgt < 0 && tgt != kVrowNone;
    // partial sums travel write-through / L1-bypassing (sc1): writer and reader may sit on different XCDs
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        a.partial + (size_t)slab * a.n_partial * 64, 0, a.n_partial * 256, 0x00020000);
    float er[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tgt >= 0) {
        finish_row<MODE, NT_ST>(a, slab, tgt, gl, acc, in, er);
    } else if (seg) {
        const unsigned at = (unsigned)(-(tgt + 1)) * 256u + (unsigned)gl * 32u;
        st_sc1(qrs, at, acc[0], acc[1], acc[2], acc[3]);
        st_sc1(qrs, at + 16u, acc[4], acc[5], acc[6], acc[7]);
    }
    // Long rows arrive as segments in different wavefronts; the segment that arrives LAST (agent-scope
    // counter) lends its whole wavefront to the row: the 8 lane groups stride over the row's partial sums, the 8
    // group totals are added with xor-shuffles -- a fixed summation order, whoever comes last -- and the row is
    // finished.  No second kernel per sweep.
    if constexpr (MODE == kPprModeF) {
        if (a.est) {   // wave-uniform.  The wavefront's maximum per query -> its slot of est_ws[slab][chunk][64] (plain
                       // stores; launch_est_reduce takes the column maxima: no atomics per row)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) er[j] = fmaxf(er[j], __shfl_xor(er[j], o, 64));
            if (grp == 0) {
                f32x4_t *wp = reinterpret_cast<f32x4_t *>(a.est_ws + ((size_t)slab * a.n_chunks + (size_t)chunk) * 64 + (size_t)gl * 8);
                wp[0] = f32x4_t{er[0], er[1], er[2], er[3]};
                wp[1] = f32x4_t{er[4], er[5], er[6], er[7]};
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(seg) == 0) return;   // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's partial sums have left the CU
    int m = -1;
    bool last = false;
    int32_t *cnts = a.lcount + (size_t)slab * a.n_lrow;
    if (seg && gl == 0) {
        m = a.seg_lrow[-(tgt + 1)];
        const int before = __hip_atomic_fetch_add(cnts + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = before == a.lrow_cnt[m] - 1;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(last);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int mm = __builtin_amdgcn_readlane(m, l);
        const int first = a.lrow_first[mm], cnt = a.lrow_cnt[mm];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int sg = grp; sg < cnt; sg += 8) {
            const unsigned at = (unsigned)(first + sg) * 256u + (unsigned)gl * 32u;
            const f32x4_t v0 = ld_sc1(qrs, at), v1 = ld_sc1(qrs, at + 16u);
            acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w;
            acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
        }
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
        if (grp == 0) {
            const int row = a.lrow_row[mm];
            load_row_in<MODE>(a, slab, row, gl, in);
#pragma unroll
            for (int j = 0; j < 8; ++j) er[j] = 0.f;
            finish_row<MODE>(a, slab, row, gl, acc, in, er);
            if constexpr (MODE == kPprModeF) {
                if (a.est) {   // a handful of rows
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = slab * 64 + gl * 8 + j;
                        if (q < a.batch && er[j] > 0.f) atomicMax(&a.est[q], __float_as_int(er[j]));
                    }
                }
            }
        }
        if (lane == 0) __hip_atomic_store(cnts + mm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// h_0 = f16(v): every vertex row of every slab
__global__ __launch_bounds__(256) void ppr16_init_kernel(const Ppr16Args a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t >> 3;
    const int gl = (int)(t & 7);
    const int slab = blockIdx.y;
    if (row >= a.num_vertices) return;
    half8_t o = {0, 0, 0, 0, 0, 0, 0, 0};
    const int slot = a.row_slot[row];
    if (slot >= 0) {
        const f32x4_t *tp = reinterpret_cast<const f32x4_t *>(
            a.tele + ((size_t)slab * a.tele_rows + (size_t)slot) * 64 + (size_t)gl * 8);
        const f32x4_t t0 = tp[0], t1 = tp[1];
        o[0] = (_Float16)clamp_half(t0.x); o[1] = (_Float16)clamp_half(t0.y);
        o[2] = (_Float16)clamp_half(t0.z); o[3] = (_Float16)clamp_half(t0.w);
        o[4] = (_Float16)clamp_half(t1.x); o[5] = (_Float16)clamp_half(t1.y);
        o[6] = (_Float16)clamp_half(t1.z); o[7] = (_Float16)clamp_half(t1.w);
    }
    *reinterpret_cast<half8_t *>(a.y + ((size_t)slab * a.num_vertices + (size_t)row) * 64 + (size_t)gl * 8) = o;
}

// Per-query scale s_q (a power of two) such that sum(v_q) * s_q is in (2^14, 2^15]: every entry of
// every iterate is then <= 2^15 < 65504 (the leaky iteration never gains mass).
//   total_q = passage_node_weight * sum_p minmax(score_qp) + sum_j seed_w[q][j]
__global__ void ppr16_scale_kernel(const float *mn, const float *mx, const float *ssum, int64_t n_passages,
                                   float passage_weight, const float *seed_w, const int32_t *seed_cnt,
                                   const int32_t *flags, int32_t batch, float *qscale) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    double total = 0.0;
    if (!(flags[q] & 1)) {
        const double range = (double)mx[q] - (double)mn[q];
        const double norm_sum = range == 0.0 ? (double)n_passages
                                             : ((double)ssum[q] - (double)n_passages * (double)mn[q]) / range;
        total = (double)passage_weight * fmax(norm_sum, 0.0);
        for (int j = 0; j < seed_cnt[q]; ++j) total += fmax((double)seed_w[q * kMaxSeeds + j], 0.0);
    }
    float s = 1.f;
    if (total > 0.0 && total < 1e300) {
        int ex;
        (void)frexp(32768.0 / total, &ex);   // 32768/total = m * 2^ex, m in [0.5, 1)
        ex -= 1;                              // floor(log2)
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        s = ldexpf(1.f, ex);
    }
    qscale[q] = s;
}

// Entity seeds become extra teleport rows: vertex v of query q gets slot n_passages + q*kMaxSeeds+j
// (first claimant wins; another query seeding the same vertex reuses the winner's row), or, when
// v is a passage vertex, its weight is added to that passage's teleport row.
__global__ void ppr16_seed_rows_kernel(const int32_t *seed_vtx, const float *seed_w,
                                       const int32_t *seed_cnt, const float *qscale, int32_t batch,
                                       int64_t n_passages, int64_t num_vertices, int32_t *row_slot,
                                       float *tele, int64_t tele_rows, int32_t bc, int64_t row_offset,
                                       int64_t n_rows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = t / kMaxSeeds, j = t % kMaxSeeds;
    if (q >= batch || j >= seed_cnt[q]) return;
    const int64_t v = seed_vtx[q * kMaxSeeds + j];
    if (v < 0 || v >= num_vertices) return;
    const int64_t lv = v - row_offset;          // row shard: only the owner of v carries its teleport row
    if (lv < 0 || lv >= n_rows) return;
    const int mine = (int)n_passages + q * kMaxSeeds + j;
    const int old = atomicCAS(&row_slot[lv], -1, mine);
    const int slot = old == -1 ? mine : old;
    const int slab = q / bc, col = q % bc;
    // vertices are unique within a query, so (slot, col) has a single writer
    tele[((size_t)slab * tele_rows + (size_t)slot) * bc + col] += seed_w[q * kMaxSeeds + j] * (qscale ? qscale[q] : 1.f);
}

template <int MODE>
hrag_status sweep_mode(const Ppr16Args &a, int n_slabs, int nt, bool main_only, hipStream_t s) {
    (void)main_only;   // long rows are finished inside the sweep kernel (last-arriving segment)
    if (a.n_chunks <= 0) return HRAG_OK;
    dim3 grid((unsigned)ceil_div(a.n_chunks, 4), (unsigned)n_slabs);
    if constexpr (MODE == kPprModeH) {
        if (a.colmask) {   // first sweep
            if (nt & 2) hipLaunchKernelGGL((ppr16_kernel<MODE, true, true, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((ppr16_kernel<MODE, true, false, true>), grid, dim3(256), 0, s, a);
            HRAG_LAUNCH_CHECK();
            return HRAG_OK;
        }
    }
    switch (nt & 3) {   // bit0: non-temporal pair loads, bit1: non-temporal state stores
        case 0: hipLaunchKernelGGL((ppr16_kernel<MODE, false, false>), grid, dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL((ppr16_kernel<MODE, true, false>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((ppr16_kernel<MODE, false, true>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((ppr16_kernel<MODE, true, true>), grid, dim3(256), 0, s, a); break;
    }
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace

hrag_status launch_ppr16_sweep(const Ppr16Args &a, int mode, int n_slabs, int nt_pairs, bool main_only,
                               hipStream_t s) {
    switch (mode) {
        case kPprModeH: return sweep_mode<kPprModeH>(a, n_slabs, nt_pairs, main_only, s);
        case kPprModeR: return sweep_mode<kPprModeR>(a, n_slabs, nt_pairs, main_only, s);
        case kPprModeC: return sweep_mode<kPprModeC>(a, n_slabs, nt_pairs, main_only, s);
        case kPprModeF:
            HRAG_TRY(sweep_mode<kPprModeF>(a, n_slabs, nt_pairs, main_only, s));
            if (a.est && a.n_chunks > 0)
                return launch_est_reduce(a.est_ws, a.n_chunks, 64, 0, n_slabs, a.batch, a.est, a.gate, a.gate_want, s);
            return HRAG_OK;
        default: set_error("bad ppr16 mode %d", mode); return HRAG_EINVAL;
    }
}

hrag_status launch_ppr16_init(const Ppr16Args &a, int n_slabs, hipStream_t s) {
    dim3 grid((unsigned)ceil_div(a.num_vertices * 8, 256), (unsigned)n_slabs);
    hipLaunchKernelGGL(ppr16_init_kernel, grid, dim3(256), 0, s, a);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr16_scale(const float *mn, const float *mx, const float *ssum, int64_t n_passages,
                               float passage_weight, const float *seed_w, const int32_t *seed_cnt,
                               const int32_t *flags, int32_t batch, float *qscale, hipStream_t s) {
    hipLaunchKernelGGL(ppr16_scale_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s, mn, mx,
                       ssum, n_passages, passage_weight, seed_w, seed_cnt, flags, batch, qscale);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr16_seed_rows(const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt,
                                   const float *qscale, int32_t batch, int64_t n_passages,
                                   int64_t num_vertices, int32_t *row_slot, float *tele,
                                   int64_t tele_rows, int32_t bc, hipStream_t s, int64_t row_offset,
                                   int64_t n_rows) {
    const int total = batch * kMaxSeeds;
    hipLaunchKernelGGL(ppr16_seed_rows_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s,
                       seed_vtx, seed_w, seed_cnt, qscale, batch, n_passages, num_vertices, row_slot,
                       tele, tele_rows, bc, row_offset, n_rows < 0 ? num_vertices : n_rows);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
