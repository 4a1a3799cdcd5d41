// K3b -- PPR power iteration with an fp8 (OCP e4m3) state and an fp32 true residual.
//
// Replaces igraph/PRPACK behind HippoRAG.run_ppr (reference src/hipporag/HippoRAG.py:1736-1743)
// for batches wider than 64 queries: one gathered line = 128 queries (two adjacent lines per gather at B > 128).
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
// from P and the weighted degrees: at_ij = p_ij d_j / d_i).  Long rows: segments whose partial sums the
// last-arriving wavefront adds up in a fixed order inside the sweep kernel (no second launch, no atomics on
// the data, bit-reproducible).  All arithmetic is fp32; fp8 -> fp32 is exact.
//
// Two kernels run the same sweep: ppr8_pair_kernel -- a wavefront carries TWO adjacent slabs of its 8 rows, so a
// gathered row is one 256-byte piece (the slabs of a vertex are adjacent: state [group][V + 1][spg][128] with spg
// even) -- takes the slab pairs, ppr8_kernel (one slab per wavefront) an odd last slab.
//
// Row shards (multi-GPU, the layout BASELINE.json's north star names): an engine owns rows
// [row_offset, row_offset + n_rows); its SELL-8 matrix, R, the stage copies and v cover those rows only,
// the e4m3 iterate is replicated and the owners' blocks are exchanged between sweeps (1 byte per
// vertex and query on the wire).  A single GPU is the shard that owns everything.
//
// What the sweeps do NOT compute (both follow from the structure of the path, HippoRAG.py:1745):
//   * only the passage rows of the result are ever read, so the last sweep (mode F) runs over the passage
//     rows of the matrix alone (~6 % of the entries on the benchmark graph), and the normalisation
//     sum(x) is not summed up: P is column-stochastic apart from isolated vertices, whose share is known
//     sweep by sweep, so the mass of the K-sweep iterate is a closed form of sum(v) (ppr8_scale_kernel);
//   * c_0 = Q(v/d) is zero outside the passage and seed vertices, so the first sweep (mode B0) tests a
//     column bitmap and issues only the gathers of those columns.
#include <algorithm>

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

// true when a value lies outside the e4m3 range (the static scale bound of its stage was violated)
__device__ __forceinline__ bool any_sat16(const f32x2_t (&f)[8]) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fmaxf(fabsf(f[j].x), fabsf(f[j].y)));
    return m > kE4m3Max;
}
// rare path: name the queries whose value was clamped (flags bit 3); lane gl owns queries 16 gl .. 16 gl + 15
__device__ __forceinline__ void flag_sat16(const f32x2_t (&f)[8], int slab, int gl, int32_t batch, int32_t *flags) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = slab * 128 + gl * 16 + 2 * j;
        if (fabsf(f[j].x) > kE4m3Max && q < batch) atomicOr(&flags[q], kFlagFp8Saturated);
        if (fabsf(f[j].y) > kE4m3Max && q + 1 < batch) atomicOr(&flags[q + 1], kFlagFp8Saturated);
    }
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
                                                const char *xs, unsigned stride, unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        wk[K] = __int_as_float(bcast8<K>(wbits));
        // `stride` bytes per vertex, 16 per lane; V + 1 <= 2^24 and (V + 1) * stride <= 2^32 are guaranteed by
        // the state layout, so the full-rate 24-bit multiply-add forms the offset
        xv[K] = *reinterpret_cast<const v4i_t *>(xs + (size_t)(__umul24(ck, stride) + lane_off));
        if constexpr (K + 1 < 8) Gather8<K + 1>::load(xv, wk, c, wbits, xs, stride, lane_off);
    }
};
__device__ __forceinline__ void gather_step(f32x2_t (&acc)[8], int c, int wbits, const char *xs,
                                            unsigned stride, unsigned lane_off) {
    v4i_t xv[8];
    float wk[8];
    Gather8<0>::load(xv, wk, c, wbits, xs, stride, lane_off);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) fma16(acc, wk[k], xv[k]);
}

// Mode B0: the gather of a column whose c_0 row is known to be zero is not issued at all (its lanes are
// masked off for the load): c_0 is non-zero only at the passage and seed vertices, ~6 % of the entries.
template <int K>
struct Gather8M {
    __device__ __forceinline__ static void load(v4i_t (&xv)[8], float (&wk)[8], int c, int wbits, int on,
                                                const char *xs, unsigned stride, unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        wk[K] = __int_as_float(bcast8<K>(wbits));
        const int onk = bcast8<K>(on);
        v4i_t v = {0, 0, 0, 0};
        if (onk) v = *reinterpret_cast<const v4i_t *>(xs + (size_t)(__umul24(ck, stride) + lane_off));
        xv[K] = v;
        if constexpr (K + 1 < 8) Gather8M<K + 1>::load(xv, wk, c, wbits, on, xs, stride, lane_off);
    }
};
__device__ __forceinline__ void gather_step_masked(f32x2_t (&acc)[8], int c, int wbits, int on, const char *xs,
                                                   unsigned stride, unsigned lane_off) {
    v4i_t xv[8];
    float wk[8];
    Gather8M<0>::load(xv, wk, c, wbits, on, xs, stride, lane_off);
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
// the same rows with sc1 (aux bit 4): write-through to memory / served past the CU's L1 -- the forms that make
// data written by one workgroup readable by another INSIDE a launch (per-XCD L2s are not coherent)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld16i_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned row_off, int gl, f32x2_t (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row_off + (unsigned)(8 * i + gl) * 16u, 0, 16);
        f[2 * i] = f32x2_t{__uint_as_float(v.x), __uint_as_float(v.y)};
        f[2 * i + 1] = f32x2_t{__uint_as_float(v.z), __uint_as_float(v.w)};
    }
}
__device__ __forceinline__ void st16i_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned row_off, int gl, const f32x2_t (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const v4u_t v = {__float_as_uint(f[2 * i].x), __float_as_uint(f[2 * i].y), __float_as_uint(f[2 * i + 1].x),
                         __float_as_uint(f[2 * i + 1].y)};
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, row_off + (unsigned)(8 * i + gl) * 16u, 0, 16);
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

// fp16 rows (the remainder rho of the 3-byte residual form, see finish_row): 128 halfs = 16 chunks of 16 bytes,
// chunk 8 * i + gl holds queries 16 * gl + 8 * i .. + 7 (lane gl's i-th half): every store instruction of an
// 8-lane group writes one whole 128-byte line, like the fp32 rows above.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void ld16h(const uint16_t *row, int gl, f32x2_t (&f)[8]) {
    const half8_t *p = reinterpret_cast<const half8_t *>(row) + gl;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const half8_t v = __builtin_nontemporal_load(p + 8 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * i + j] = f32x2_t{(float)v[2 * j], (float)v[2 * j + 1]};
    }
}
__device__ __forceinline__ void st16h(uint16_t *row, int gl, const f32x2_t (&f)[8]) {
    half8_t *p = reinterpret_cast<half8_t *>(row) + gl;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        half8_t v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = (_Float16)f[4 * i + j].x;
            v[2 * j + 1] = (_Float16)f[4 * i + j].y;
        }
        __builtin_nontemporal_store(v, p + 8 * i);
    }
}

// z_v = v / d for one owned row (lrow: LOCAL index, grow: global vertex id): v comes from the teleport
// rows (fp32 [n_slabs64][tele_rows][64] + row_slot), already scaled per query.
__device__ __forceinline__ void load_zv(const float *__restrict__ tele, int64_t tele_rows,
                                        const int32_t *__restrict__ row_slot, const float *__restrict__ deg,
                                        int32_t n_slabs64, int slab, int64_t lrow, int64_t grow, int gl,
                                        f32x2_t (&z)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = f32x2_t{0.f, 0.f};
    const int slot = row_slot[lrow];
    const int slab64 = 2 * slab + (gl >> 2);
    if (slot >= 0 && slab64 < n_slabs64) {
        const f32x4_t *tp = reinterpret_cast<const f32x4_t *>(
            tele + ((size_t)slab64 * tele_rows + (size_t)slot) * 64 + (size_t)(gl & 3) * 16);
        const float invd = __fdiv_rn(1.0f, deg[grow]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4_t v = tp[i];
            z[2 * i] = f32x2_t{v.x, v.y} * invd;
            z[2 * i + 1] = f32x2_t{v.z, v.w} * invd;
        }
    }
}

// byte offset of (slab, vertex, lane) in a state buffer [n_groups][V + 1][spg][128]
__device__ __forceinline__ size_t state_off(const Ppr8Args &a, int slab, int64_t grow, int gl) {
    const int g = slab / a.spg, k = slab - g * a.spg;
    return (size_t)g * (size_t)a.group_bytes + (size_t)grow * a.row_stride + (size_t)k * 128 + (size_t)gl * 16;
}

// Stage scales.  Plain plan: static powers of two in the kernel arguments.  HRAG_OPT_ACCEL (a.dyn != nullptr): MEASURED
// per stage on the device -- dyn[k] = cs of stage k, dyn[kP8DynInv + k] = 1 / cs -- because a Chebyshev stage is not a
// max-norm contraction by its spectral factor and a static chain built on its rigorous max-norm bound (7 / T_3) sinks
// the values by ~2.5x per stage below where the e4m3 range resolves them (cfg 3: 1.8e-6 instead of 4.4e-7, and the
// contract then needs 26 sweeps).  Every boundary reports max |R cs'| of the batch (mq, per wavefront -> a.mmax_ws);
// ppr8_next_scale_kernel turns it into the scale of the stage after next.  Wave-uniform scalar loads.
__device__ __forceinline__ float scale_inv(const Ppr8Args &a, int stage, float fallback) {
    return a.dyn ? a.dyn[kP8DynInv + stage] : fallback;
}
__device__ __forceinline__ float scale_cs(const Ppr8Args &a, int stage, float fallback) {
    return a.dyn ? a.dyn[stage] : fallback;
}

// Finish one output row (lrow: LOCAL row): lane gl of its group owns queries 16*gl .. 16*gl+15 of the slab.
// RIO (boundary / final modes): how the true residual R travels between two boundaries.
//   bit 0: R_in  = (rt + rho) / cs  -- the stage's quantised right-hand side rt = Q(R cs) (still in its state
//          buffer: it is only ever read at the own row) plus the fp16 remainder rho = f16(R cs - rt) that
//          the previous boundary stored: 3 bytes instead of 4, |error| <= 2^-11 |rho| <= 2^-15 |R|;
//   bit 1: R_out is stored in that form (rho only: the new rt is written anyway).
// The host switches a boundary to this form once damping^k <= 2^-6 (k = sweeps done), where 2^-15 |R| is
// below 5e-7 of the solution (csrc/shard.hip; emulation: 2.5e-7 -> 3.9e-7 on the benchmark graph against 2^-9,
// 4e-6 at 2^-3); the early boundaries keep the fp32 R.
// er (mode F with a.est): on return |R_p| / z_p of this lane's 16 queries when the row is an owned passage (the
// relative size of the update the final sweep applies to the passage score), untouched otherwise.
// mq (modes B / B0): raised to max |R_new cs'| of this lane's 16 queries (what the dynamic stage scales are measured on).
template <int MODE, int RIO, bool EST>
__device__ __forceinline__ void finish_row(const Ppr8Args &a, int slab, int lrow, int gl,
                                           const f32x2_t (&acc)[8], f32x2_t (&er)[8], float &mq) {
    const int64_t grow = a.row_offset + lrow;
    const float inv_cs = scale_inv(a, a.dyn_stage, a.inv_cs), cs_next = scale_cs(a, a.dyn_stage + 1, a.cs_next);
    const size_t off = state_off(a, slab, grow, gl);
    f32x2_t out[8];
    if constexpr (MODE == kP8ModeC) {
        f32x2_t r[8];
        decode16(__builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(a.rt + off)), r);
#pragma unroll
        for (int j = 0; j < 8; ++j)   // r_mul = 1 on a plain stage: that product is exact, one rounding as before
            out[j] = __builtin_elementwise_fma(acc[j], f32x2_t{a.c_mul, a.c_mul}, r[j] * f32x2_t{a.r_mul, a.r_mul});
        if (__builtin_expect(any_sat16(out), 0)) flag_sat16(out, slab, gl, a.batch, a.flags);
        __builtin_nontemporal_store(encode16(out), reinterpret_cast<v4i_t *>(a.y + off));
    } else {
        f32x2_t c[8], rin[8];
        const int slot = a.row_slot[lrow];
        const bool is_passage = slot >= 0 && slot < a.p_rows;
        v4i_t cv = {0, 0, 0, 0};
        // B0: c_0 = Q(v/d) exists only on the rows with a teleport row (ppr8_init_kernel writes no others)
        if (MODE != kP8ModeB0 || slot >= 0) cv = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(a.x + off));
        decode16(cv, c);
        float *rrow = a.R + ((size_t)slab * a.n_rows + (size_t)lrow) * 128;
        uint16_t *hrow = a.rho + ((size_t)slab * a.n_rows + (size_t)lrow) * 128;
        if constexpr (MODE == kP8ModeB0) {
            load_zv(a.tele, a.tele_rows, a.row_slot, a.deg, a.n_slabs64, slab, lrow, grow, gl, rin);   // R_in = b v/d
#pragma unroll
            for (int j = 0; j < 8; ++j) rin[j] *= a.beta;
        } else if constexpr ((RIO & 1) != 0) {
            f32x2_t r8[8];
            decode16(__builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(a.rt + off)), r8);
            ld16h(hrow, gl, rin);
#pragma unroll
            for (int j = 0; j < 8; ++j) rin[j] = (rin[j] + r8[j]) * inv_cs;   // inv_cs is a power of two: exact
        } else {
            ld16i(rrow, gl, rin);
        }
        const f32x2_t al = {a.alpha, a.alpha}, inv = {inv_cs, inv_cs};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2_t t = __builtin_elementwise_fma(acc[j], al, -c[j]);   // a (At c) - c, one rounding
            out[j] = __builtin_elementwise_fma(t, inv, rin[j]);              // inv is a power of two: exact
        }
        if constexpr (MODE == kP8ModeB || MODE == kP8ModeB0) {
            f32x2_t q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = out[j] * cs_next;
            {
                float m = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fmaxf(fabsf(q[j].x), fabsf(q[j].y)));
                mq = fmaxf(mq, m);
                if (__builtin_expect(m > kE4m3Max, 0)) flag_sat16(q, slab, gl, a.batch, a.flags);
            }
            const v4i_t enc = encode16(q);
            __builtin_nontemporal_store(enc, reinterpret_cast<v4i_t *>(a.y + off));
            if constexpr ((RIO & 2) != 0) {
                f32x2_t back[8];
                decode16(enc, back);
#pragma unroll
                for (int j = 0; j < 8; ++j) back[j] = q[j] - back[j];          // exact: |q - Q(q)| <= 2^-4 |q|
                st16h(hrow, gl, back);
            } else {
                st16i(rrow, gl, out);
            }
            // the final combine needs every stage's c at the passage rows only: keep a compact copy
            if (is_passage) {
                const size_t poff = ((size_t)slab * a.p_rows + (size_t)slot) * 128 + (size_t)gl * 16;
                *reinterpret_cast<v4i_t *>(a.stage_out + poff) = cv;
            }
        } else {   // kP8ModeF: z = R' + sum_s c_s / cs_s (earliest stage first), x = d z
            if (!is_passage) return;
            const size_t poff = ((size_t)slab * a.p_rows + (size_t)slot) * 128 + (size_t)gl * 16;
            f32x2_t z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = f32x2_t{0.f, 0.f};
            for (int s = 0; s + 1 < a.n_stage; ++s) {
                f32x2_t cs[8];
                decode16(*reinterpret_cast<const v4i_t *>(a.stage[s] + poff), cs);
                const float sinv = scale_inv(a, s, a.stage_inv[s]);
                const f32x2_t si = {sinv, sinv};
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = __builtin_elementwise_fma(cs[j], si, z[j]);
            }
            const float linv = scale_inv(a, a.n_stage - 1, a.stage_inv[a.n_stage - 1]);
            const f32x2_t sl = {linv, linv};
            const float dg = a.deg[grow];
            f32x2_t xs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                z[j] = __builtin_elementwise_fma(c[j], sl, z[j]);
                const f32x2_t zz = z[j] + out[j];
                xs[j] = zz * dg;
                if constexpr (EST) {
                    er[j].x = zz.x > 0.f ? fabsf(out[j].x) / zz.x : 0.f;
                    er[j].y = zz.y > 0.f ? fabsf(out[j].y) / zz.y : 0.f;
                }
            }
            // passage order: xp is [n_slabs64][p_rows][64] fp32, the layout slab_to_rows reads without a gather
            const int slab64 = 2 * slab + (gl >> 2);
            if (slab64 < a.n_slabs64)
                st16f(a.out + ((size_t)slab64 * a.p_rows + (size_t)slot) * 64 + (size_t)(gl & 3) * 16, xs);
        }
    }
}

// The wavefront's maximum of er over its 8 rows, 128 queries, goes to its own 512-byte row of the scratch array
// est_ws[slab][pslot[chunk]][128] with plain stores (every wavefront whose chunk holds a passage row writes, zeros
// included: nothing to initialise; the other chunks have no row -- the rows are sorted by length with the passages
// last inside a length class, so ~ Np / 8 of the chunks do); est_reduce_kernel takes the column maxima afterwards.  No atomics on this path: 4 M atomicMax on 256 addresses cost
// 0.3 ms per sweep when tried, and a pre-check needs a coherent (memory-latency) load per value.
// WAVE = false (a long row finished by the last-arriving wavefront, group 0 only; a handful of rows): atomicMax.
template <bool WAVE>
__device__ __forceinline__ void est_commit(const Ppr8Args &a, int slab, int chunk, int gl, int grp, f32x2_t (&er)[8]) {
    if constexpr (WAVE) {
        const int slot = a.m.pslot[chunk];   // wave-uniform
        if (slot < 0) return;
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                er[j].x = fmaxf(er[j].x, __shfl_xor(er[j].x, o, 64));
                er[j].y = fmaxf(er[j].y, __shfl_xor(er[j].y, o, 64));
            }
        if (grp != 0) return;
        st16f(a.est_ws + ((size_t)slab * a.m.n_pchunks + (size_t)slot) * 128 + (size_t)gl * 16, er);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = slab * 128 + gl * 16 + 2 * j;
            if (q < a.batch && er[j].x > 0.f) atomicMax(&a.est[q], __float_as_int(er[j].x));
            if (q + 1 < a.batch && er[j].y > 0.f) atomicMax(&a.est[q + 1], __float_as_int(er[j].y));
        }
    }
}

// Dynamic stage scales (HRAG_OPT_ACCEL): the wavefront's maximum of |R_new cs'| goes to ITS slot of a.mmax_ws (plain
// store, every slot is rewritten by every boundary launch: nothing to initialise); WAVE = false (a long row finished by
// the last-arriving wavefront: a handful of rows): integer atomicMax on the bits of the non-negative float.
template <bool WAVE>
__device__ __forceinline__ void mmax_commit(const Ppr8Args &a, int slot, float mq) {
    if (!a.mmax_ws) return;   // wave-uniform
    if constexpr (WAVE) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mq = fmaxf(mq, __shfl_xor(mq, o, 64));
        if ((threadIdx.x & 63) == 0) a.mmax_ws[slot] = mq;
    } else {
        if (mq > 0.f) atomicMax(a.mmax_atomic, __float_as_int(mq));
    }
}

__device__ __forceinline__ int ld_mask(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0);
}

// 1-D grid, XCD-aware: workgroups are dealt to the 8 XCDs round-robin, so ids 8 apart run back to back
// on the same XCD.  They are given the SAME chunk group for consecutive slabs: the second reader of
// a (col, val) block then finds it in that XCD's L2 (measured: C sweep 0.822 -> 0.802 ms at cfg 3;
// with non-temporal pair loads the remap alone changes nothing).
template <int MODE, int RIO, bool EST>
__global__ __launch_bounds__(256, 4) void ppr8_kernel(const Ppr8Args a) {
    if (a.gate && *a.gate != a.gate_want) return;   // a conditional step the device decided not to run
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7, grp = lane >> 3;
    // a workgroup = 4 wavefronts = (4 / wps) chunks x wps slabs: the wavefronts that work on the same chunk for
    // different slabs read the same (col, val) blocks at about the same time from the same CU
    const int id = blockIdx.x, wps = a.wps, nsg = a.n_slabs / wps, wave = threadIdx.x >> 6;
    const int slab = a.slab0 + ((id >> 3) % nsg) * wps + (wave & (wps - 1));
    const int cg = a.cg_per_xcd > 0 ? (id & 7) * a.cg_per_xcd + id / (8 * nsg) : (id / (8 * nsg)) * 8 + (id & 7);
    const int chunk = __builtin_amdgcn_readfirstlane(cg * (4 / wps) + wave / wps);
    if (chunk >= a.m.n_chunks) return;
    const int2 meta = a.m.chunk_meta[chunk];  // (first step, number of steps)
    const int n_steps = meta.y;
    const int g = slab / a.spg, k = slab - g * a.spg;
    const char *xs = reinterpret_cast<const char *>(a.x) + (size_t)g * (size_t)a.group_bytes + (size_t)k * 128;
    const unsigned stride = a.row_stride;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int2 *>(a.m.pairs), 0, (int)a.m.pairs_bytes, 0x00020000);
    const unsigned pbase = (unsigned)meta.x * 512u;   // scalar: first byte of this chunk's pairs
    const unsigned poff = (unsigned)lane * 8u;
    const unsigned lane_off = (unsigned)gl * 16u;
    f32x2_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x2_t{0.f, 0.f};
    // pair stream read two steps ahead, unconditionally (the array carries the padding)
    int2 p0 = ld_pair(prs, poff, pbase);
    int2 p1 = ld_pair(prs, poff + 512u, pbase);
    if constexpr (MODE == kP8ModeB0) {
        // c_0 is zero outside the passage / seed vertices: test the column bitmap (one step ahead) and issue
        // only the gathers of those columns
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(a.colmask), 0, (int)a.colmask_bytes, 0x00020000);
        int m0 = ld_mask(mrs, ((unsigned)p0.x >> 5) * 4u);
        for (int s = 0; s < n_steps; ++s) {
            const int2 p2 = ld_pair(prs, poff + (unsigned)(s + 2) * 512u, pbase);
            const int m1 = ld_mask(mrs, ((unsigned)p1.x >> 5) * 4u);
            gather_step_masked(acc, p0.x, p0.y, (m0 >> (p0.x & 31)) & 1, xs, stride, lane_off);
            p0 = p1;
            p1 = p2;
            m0 = m1;
        }
    } else {
        for (int s = 0; s < n_steps; ++s) {
            const int2 p2 = ld_pair(prs, poff + (unsigned)(s + 2) * 512u, pbase);
            gather_step(acc, p0.x, p0.y, xs, stride, lane_off);
            p0 = p1;
            p1 = p2;
        }
    }
    const int tgt = a.m.vrow[chunk * 8 + grp];
    const bool seg = tgt < 0 && tgt != kVrowNone;
    // partial sums travel write-through / L1-bypassing (sc1): writer and reader may sit on different XCDs
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        a.partial + (size_t)slab * a.m.n_partial * 128, 0, a.m.n_partial * 512, 0x00020000);
    constexpr bool kEst = EST && MODE == kP8ModeF;
    f32x2_t er[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) er[j] = f32x2_t{0.f, 0.f};
    float mq = 0.f;
    if (tgt >= 0) {
        finish_row<MODE, RIO, EST>(a, slab, tgt, gl, acc, er, mq);
    } else if (seg) {
        st16i_sc1(qrs, (unsigned)(-(tgt + 1)) * 512u, gl, acc);
    }
    if constexpr (kEst) est_commit<true>(a, slab, chunk, gl, grp, er);   // every lane takes part in the reduction
    if constexpr (MODE == kP8ModeB || MODE == kP8ModeB0) mmax_commit<true>(a, chunk * a.mmax_units + (slab - a.mmax_slab0), mq);
    // Long rows arrive as segments in different wavefronts; the segment that arrives LAST (agent-scope
    // counter) lends its whole wavefront to the row: the 8 lane groups stride over the row's partial sums, the 8
    // group totals are added with xor-shuffles -- a fixed summation order, whoever comes last -- and the row is
    // finished.  No second kernel per sweep.
    if (__builtin_amdgcn_ballot_w64(seg) == 0) return;   // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's partial sums have left the CU
    int m = -1;
    bool last = false;
    int32_t *cnts = a.m.lcount + (size_t)slab * a.m.n_lrow;
    if (seg && gl == 0) {
        m = a.m.seg_lrow[-(tgt + 1)];
        const int before = __hip_atomic_fetch_add(cnts + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = before == a.m.lrow_cnt[m] - 1;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(last);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int mm = __builtin_amdgcn_readlane(m, l);
        const int first = a.m.lrow_first[mm], cnt = a.m.lrow_cnt[mm];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x2_t{0.f, 0.f};
        for (int sg = grp; sg < cnt; sg += 8) {
            f32x2_t v[8];
            ld16i_sc1(qrs, (unsigned)(first + sg) * 512u, gl, v);
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
#pragma unroll
            for (int j = 0; j < 8; ++j) er[j] = f32x2_t{0.f, 0.f};
            float m2 = 0.f;
            finish_row<MODE, RIO, EST>(a, slab, a.m.lrow_row[mm], gl, acc, er, m2);
            if constexpr (kEst) est_commit<false>(a, slab, chunk, gl, 0, er);
            if constexpr (MODE == kP8ModeB || MODE == kP8ModeB0) mmax_commit<false>(a, 0, m2);
        }
        if (lane == 0) __hip_atomic_store(cnts + mm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Two slabs per wavefront ("pair" kernel).  With the slabs of a vertex adjacent in memory (spg even: vertex-major
// state) a lane issues the 16-byte loads of slab s and slab s + 1 of the same vertex back to back: the memory system
// sees 256 contiguous bytes per gathered row instead of two unrelated 128-byte lines (tools/membench.hip: random
// 256-byte pieces stream 7 % faster than random 128-byte lines), and the (col, val) stream is read once per pair.
// Twice the accumulators and loads in flight per wavefront, fewer wavefronts per SIMD.
template <int K>
struct Gather8P {
    __device__ __forceinline__ static void load(v4i_t (&x0)[8], v4i_t (&x1)[8], float (&wk)[8], int c, int wbits,
                                                const char *xs, unsigned stride, unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        wk[K] = __int_as_float(bcast8<K>(wbits));
        const char *p = xs + (size_t)(__umul24(ck, stride) + lane_off);
        x0[K] = *reinterpret_cast<const v4i_t *>(p);
        x1[K] = *reinterpret_cast<const v4i_t *>(p + 128);
        if constexpr (K + 1 < 8) Gather8P<K + 1>::load(x0, x1, wk, c, wbits, xs, stride, lane_off);
    }
};

template <int K>
struct Gather8PM {   // mode B0: masked (see Gather8M)
    __device__ __forceinline__ static void load(v4i_t (&x0)[8], v4i_t (&x1)[8], float (&wk)[8], int c, int wbits, int on,
                                                const char *xs, unsigned stride, unsigned lane_off) {
        const unsigned ck = (unsigned)bcast8<K>(c);
        wk[K] = __int_as_float(bcast8<K>(wbits));
        const int onk = bcast8<K>(on);
        v4i_t v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
        if (onk) {
            const char *p = xs + (size_t)(__umul24(ck, stride) + lane_off);
            v0 = *reinterpret_cast<const v4i_t *>(p);
            v1 = *reinterpret_cast<const v4i_t *>(p + 128);
        }
        x0[K] = v0;
        x1[K] = v1;
        if constexpr (K + 1 < 8) Gather8PM<K + 1>::load(x0, x1, wk, c, wbits, on, xs, stride, lane_off);
    }
};

template <int MODE, int RIO, bool EST>
__device__ __forceinline__ void ppr8_pair_body(const Ppr8Args &a) {
    if (a.gate && *a.gate != a.gate_want) return;   // a conditional step the device decided not to run
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7, grp = lane >> 3;
    // a workgroup = 4 wavefronts = (4 / wps) chunks x wps slab PAIRS
    const int id = blockIdx.x, wps = a.wps, nsg = (a.n_slabs >> 1) / wps, wave = threadIdx.x >> 6;
    const int slab = a.slab0 + 2 * (((id >> 3) % nsg) * wps + (wave & (wps - 1)));
    const int cg = a.cg_per_xcd > 0 ? (id & 7) * a.cg_per_xcd + id / (8 * nsg) : (id / (8 * nsg)) * 8 + (id & 7);
    const int chunk = __builtin_amdgcn_readfirstlane(cg * (4 / wps) + wave / wps);
    if (chunk >= a.m.n_chunks) return;
    const int2 meta = a.m.chunk_meta[chunk];  // (first step, number of steps)
    const int n_steps = meta.y;
    const int g = slab / a.spg, k = slab - g * a.spg;
    const char *xs = reinterpret_cast<const char *>(a.x) + (size_t)g * (size_t)a.group_bytes + (size_t)k * 128;
    const unsigned stride = a.row_stride;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int2 *>(a.m.pairs), 0, (int)a.m.pairs_bytes, 0x00020000);
    const unsigned pbase = (unsigned)meta.x * 512u;
    const unsigned poff = (unsigned)lane * 8u;
    const unsigned lane_off = (unsigned)gl * 16u;
    f32x2_t acc0[8], acc1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc0[j] = acc1[j] = f32x2_t{0.f, 0.f};
    int2 p0 = ld_pair(prs, poff, pbase);
    int2 p1 = ld_pair(prs, poff + 512u, pbase);
    if constexpr (MODE == kP8ModeB0) {
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(a.colmask), 0, (int)a.colmask_bytes, 0x00020000);
        int m0 = ld_mask(mrs, ((unsigned)p0.x >> 5) * 4u);
        for (int s = 0; s < n_steps; ++s) {
            const int2 p2 = ld_pair(prs, poff + (unsigned)(s + 2) * 512u, pbase);
            const int m1 = ld_mask(mrs, ((unsigned)p1.x >> 5) * 4u);
            v4i_t x0[8], x1[8];
            float wk[8];
            Gather8PM<0>::load(x0, x1, wk, p0.x, p0.y, (m0 >> (p0.x & 31)) & 1, xs, stride, lane_off);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                fma16(acc0, wk[kk], x0[kk]);
                fma16(acc1, wk[kk], x1[kk]);
            }
            p0 = p1;
            p1 = p2;
            m0 = m1;
        }
    } else {
        for (int s = 0; s < n_steps; ++s) {
            const int2 p2 = ld_pair(prs, poff + (unsigned)(s + 2) * 512u, pbase);
            v4i_t x0[8], x1[8];
            float wk[8];
            Gather8P<0>::load(x0, x1, wk, p0.x, p0.y, xs, stride, lane_off);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                fma16(acc0, wk[kk], x0[kk]);
                fma16(acc1, wk[kk], x1[kk]);
            }
            p0 = p1;
            p1 = p2;
        }
    }
    const int tgt = a.m.vrow[chunk * 8 + grp];
    const bool seg = tgt < 0 && tgt != kVrowNone;
    const __amdgpu_buffer_rsrc_t q0 = __builtin_amdgcn_make_buffer_rsrc(
        a.partial + (size_t)slab * a.m.n_partial * 128, 0, a.m.n_partial * 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t q1 = __builtin_amdgcn_make_buffer_rsrc(
        a.partial + (size_t)(slab + 1) * a.m.n_partial * 128, 0, a.m.n_partial * 512, 0x00020000);
    constexpr bool kEst = EST && MODE == kP8ModeF;
    f32x2_t er[8];
    float mq = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int j = 0; j < 8; ++j) er[j] = f32x2_t{0.f, 0.f};
        if (tgt >= 0) {
            finish_row<MODE, RIO, EST>(a, slab + half, tgt, gl, half ? acc1 : acc0, er, mq);
        } else if (seg) {
            st16i_sc1(half ? q1 : q0, (unsigned)(-(tgt + 1)) * 512u, gl, half ? acc1 : acc0);
        }
        if constexpr (kEst) est_commit<true>(a, slab + half, chunk, gl, grp, er);
    }
    if constexpr (MODE == kP8ModeB || MODE == kP8ModeB0) mmax_commit<true>(a, chunk * a.mmax_units + (slab - a.mmax_slab0), mq);
    // long rows: as in ppr8_kernel; one arrival (the first slab's counter) covers both slabs of the pair
    if (__builtin_amdgcn_ballot_w64(seg) == 0) return;   // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int m = -1;
    bool last = false;
    int32_t *cnts = a.m.lcount + (size_t)slab * a.m.n_lrow;
    if (seg && gl == 0) {
        m = a.m.seg_lrow[-(tgt + 1)];
        const int before = __hip_atomic_fetch_add(cnts + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = before == a.m.lrow_cnt[m] - 1;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(last);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int mm = __builtin_amdgcn_readlane(m, l);
        const int first = a.m.lrow_first[mm], cnt = a.m.lrow_cnt[mm];
        for (int half = 0; half < 2; ++half) {
            const __amdgpu_buffer_rsrc_t qh = half ? q1 : q0;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc0[j] = f32x2_t{0.f, 0.f};
            for (int sg = grp; sg < cnt; sg += 8) {
                f32x2_t v[8];
                ld16i_sc1(qh, (unsigned)(first + sg) * 512u, gl, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc0[j] += v[j];
            }
#pragma unroll
            for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc0[j].x += __shfl_xor(acc0[j].x, o, 64);
                    acc0[j].y += __shfl_xor(acc0[j].y, o, 64);
                }
            if (grp == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) er[j] = f32x2_t{0.f, 0.f};
                float m2 = 0.f;
                finish_row<MODE, RIO, EST>(a, slab + half, a.m.lrow_row[mm], gl, acc0, er, m2);
                if constexpr (kEst) est_commit<false>(a, slab + half, chunk, gl, 0, er);
                if constexpr (MODE == kP8ModeB || MODE == kP8ModeB0) mmax_commit<false>(a, 0, m2);
            }
        }
        if (lane == 0) __hip_atomic_store(cnts + mm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE, int RIO, bool EST>
__global__ __launch_bounds__(256, 2) void ppr8_pair_kernel(const Ppr8Args a) { ppr8_pair_body<MODE, RIO, EST>(a); }

// Measurement only (hrag_ppr_sweeps flag 256; bench.py `roofline.gather_replay_ms`): the GATHERS of a stage sweep and
// nothing else -- the same workgroup -> (chunk, slab pair) map, the same (col, val) stream read two steps ahead, the
// same sixteen 16-byte loads per lane and step (one 256-byte piece of the state per matrix slot), no right-hand side,
// no residual, no arithmetic beyond an XOR that keeps the loads alive, nothing stored.  What it takes is the floor of
// ANY sweep that fetches the state row of every slot of this matrix once: the ceiling of the formulation, measured on
// the engine's own matrix and state.
__global__ __launch_bounds__(256, 2) void ppr8_pair_replay_kernel(const Ppr8Args a) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7;
    const int id = blockIdx.x, wps = a.wps, nsg = (a.n_slabs >> 1) / wps, wave = threadIdx.x >> 6;
    const int slab = a.slab0 + 2 * (((id >> 3) % nsg) * wps + (wave & (wps - 1)));
    const int cg = a.cg_per_xcd > 0 ? (id & 7) * a.cg_per_xcd + id / (8 * nsg) : (id / (8 * nsg)) * 8 + (id & 7);
    const int chunk = __builtin_amdgcn_readfirstlane(cg * (4 / wps) + wave / wps);
    if (chunk >= a.m.n_chunks) return;
    const int2 meta = a.m.chunk_meta[chunk];
    const int n_steps = meta.y;
    const int g = slab / a.spg, k = slab - g * a.spg;
    const char *xs = reinterpret_cast<const char *>(a.x) + (size_t)g * (size_t)a.group_bytes + (size_t)k * 128;
    const unsigned stride = a.row_stride;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int2 *>(a.m.pairs), 0, (int)a.m.pairs_bytes, 0x00020000);
    const unsigned pbase = (unsigned)meta.x * 512u;
    const unsigned poff = (unsigned)lane * 8u;
    const unsigned lane_off = (unsigned)gl * 16u;
    v4i_t sink = {0, 0, 0, 0};
    int2 p0 = ld_pair(prs, poff, pbase);
    int2 p1 = ld_pair(prs, poff + 512u, pbase);
    for (int s = 0; s < n_steps; ++s) {
        const int2 p2 = ld_pair(prs, poff + (unsigned)(s + 2) * 512u, pbase);
        v4i_t x0[8], x1[8];
        float wk[8];
        Gather8P<0>::load(x0, x1, wk, p0.x, p0.y, xs, stride, lane_off);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) sink ^= x0[kk] ^ x1[kk];
        p0 = p1;
        p1 = p2;
    }
    // never true for an e4m3 state (0x7f / 0xff bytes are NaN and are never stored): keeps the loads, stores nothing
    if ((sink.x ^ sink.y ^ sink.z ^ sink.w) == 0x7f7f7f7f && a.flags) atomicOr(a.flags, 0);
}

// c_0 = Q(v/d * c0_scale) on the OWNED rows of the launch's slabs that carry a teleport row (passages, seeds): it is
// zero elsewhere, and its only reader, the first boundary sweep (mode B0: R_0 = b v/d is formed on the fly), neither
// gathers a column outside the bitmap of those rows nor reads its own c there -- 7/8 of the state is not written.
__global__ __launch_bounds__(256) void ppr8_init_kernel(const Ppr8Args a, float c0_scale) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t lrow = t >> 3;
    const int gl = (int)(t & 7);
    const int slab = a.slab0 + blockIdx.y;
    if (lrow >= a.n_rows || a.row_slot[lrow] < 0) return;
    const int64_t grow = a.row_offset + lrow;
    f32x2_t z[8], q[8];
    load_zv(a.tele, a.tele_rows, a.row_slot, a.deg, a.n_slabs64, slab, lrow, grow, gl, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = z[j] * c0_scale;
    if (__builtin_expect(any_sat16(q), 0)) flag_sat16(q, slab, gl, a.batch, a.flags);
    *reinterpret_cast<v4i_t *>(a.y + state_off(a, slab, grow, gl)) = encode16(q);
}

// Per-query statistics of the passage prior over the owned passages, one streaming pass:
//   zmax = max_p minmax(score_qp) / d_p  (the passage part of the bound that fixes the query's scale),
//   mass = sum_p v_p with v_p = float(minmax(score_qp) * weight) exactly as rows_to_slab forms it, and
//   the same sum over the passages whose vertex has no edges.
// Grid (kP8PriorSplit, B).  The maxima are combined with an integer atomicMax on the bits of the
// non-negative floats (order-preserving: independent of the arrival order); the sums go through
// per-block partials that ppr8_prior_final_kernel adds in a fixed order (no atomics on doubles).
__global__ __launch_bounds__(256) void ppr8_prior_kernel(const float *__restrict__ scores, int64_t ld,
                                                         int64_t p_rows, const float *__restrict__ mn,
                                                         const float *__restrict__ mx, float weight,
                                                         const float *__restrict__ pinvdeg,
                                                         const uint8_t *__restrict__ piso,
                                                         const int32_t *__restrict__ flags, int32_t batch,
                                                         int32_t *zmax_bits, double *part) {
    __shared__ float red[256];
    __shared__ double reds[256], redi[256];
    const int q = blockIdx.y, tid = threadIdx.x;
    float best = 0.f;
    double sum = 0.0, siso = 0.0;
    if (!(flags[q] & 1)) {
        const float lo = mn[q], range = mx[q] - mn[q];
        const float *row = scores + (size_t)q * ld;
        for (int64_t p = (int64_t)blockIdx.x * 256 + tid; p < p_rows; p += (int64_t)kP8PriorSplit * 256) {
            const float nrm = range == 0.f ? 1.f : __fdiv_rn(row[p] - lo, range);
            best = fmaxf(best, nrm * pinvdeg[p]);
            const float v = nrm * weight;
            sum += (double)v;
            if (piso[p]) siso += (double)v;
        }
    }
    red[tid] = best; reds[tid] = sum; redi[tid] = siso;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] = fmaxf(red[tid], red[tid + s]);
            reds[tid] += reds[tid + s];
            redi[tid] += redi[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        atomicMax(&zmax_bits[q], __float_as_int(fmaxf(red[0], 0.f)));
        part[((size_t)blockIdx.x * batch + q) * 2 + 0] = reds[0];
        part[((size_t)blockIdx.x * batch + q) * 2 + 1] = redi[0];
    }
}
__global__ void ppr8_prior_final_kernel(const int32_t *__restrict__ zmax_bits, const double *__restrict__ part,
                                        int32_t batch, float *zmax, double *mass) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    double s = 0.0, si = 0.0;
    for (int b = 0; b < kP8PriorSplit; ++b) {
        s += part[((size_t)b * batch + q) * 2 + 0];
        si += part[((size_t)b * batch + q) * 2 + 1];
    }
    zmax[q] = __int_as_float(zmax_bits[q]);
    mass[q] = s;
    mass[batch + q] = si;
}

// Per-query power-of-two scale s_q with  max_i v_i / d_i * s_q  in (1/2, 1]:
//   bound_q = passage_weight * zmax_q  +  max_j seed_w / d(seed_j)
// (the sum of the two maxima covers a seed that is also a passage vertex), and the mass of the result:
// x_{k+1} = a P x_k + b v with x_0 = v keeps sum(x) = sum(v) except for what sits on isolated vertices
// (P has an empty column there), and an isolated vertex i holds x_0[i] = v_i, x_k[i] = b v_i (k >= 1):
//   m_0 = M,  m_{k+1} = a (m_k - iso_k) + b M,  iso_0 = S, iso_k = b S     (M = sum v, S = sum over isolated)
// -- the value the reference's final division by sum(x) uses (PRPACK normalises, HippoRAG.py:1745), without
// a column sum over all vertices.  zmax / mass are GLOBAL (all-reduced over the row shards).
__global__ void ppr8_scale_kernel(const float *__restrict__ zmax, const double *__restrict__ mass,
                                  float passage_weight, const int32_t *__restrict__ seed_vtx,
                                  const float *__restrict__ seed_w, const int32_t *__restrict__ seed_cnt,
                                  const float *__restrict__ deg, const uint8_t *__restrict__ iso,
                                  int64_t num_vertices, const int32_t *__restrict__ flags, int32_t batch,
                                  float damping, int32_t iters, float *qscale, double *sums, int32_t n_tab,
                                  int64_t tab_stride) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    float bound = 0.f;
    double M = 0.0, S = 0.0;
    if (!(flags[q] & 1)) {
        bound = fmaxf(passage_weight, 0.f) * zmax[q];
        M = mass[q];
        S = mass[batch + q];
        float sb = 0.f;
        for (int j = 0; j < seed_cnt[q]; ++j) {
            const int64_t v = seed_vtx[q * kMaxSeeds + j];
            if (v < 0 || v >= num_vertices) continue;
            const float w = seed_w[q * kMaxSeeds + j];
            sb = fmaxf(sb, __fdiv_rn(fmaxf(w, 0.f), deg[v]));
            M += (double)w;
            if (iso[v]) S += (double)w;
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
    M *= (double)s;   // powers of two: exact
    S *= (double)s;
    const double al = (double)damping, be = (double)(1.0f - damping);
    // sums[j * tab_stride + q] = the mass after iters + p8_ext_sweeps(j) sweeps (j > 0: the extension stages of the
    // convergence contract, csrc/shard.hip)
    double m = M;
    int k = 0;
    for (int j = 0; j < n_tab; ++j) {
        for (; k < iters + p8_ext_sweeps(j); ++k) m = al * (m - (k == 0 ? S : be * S)) + be * M;
        sums[(size_t)j * tab_stride + q] = m;
    }
}

// est[slab * W + w] = max(est[...], max over the chunks of ws[slab][chunk][w]): the column maxima of the per-wavefront
// scratch a sweep with est left behind (W = queries per slab row: 128 fp8 state, 64 fp16 state, bp small batches).
// Grid (kEstSplit, n_slabs); one atomicMax per (block, query).  gate: the sweep it follows was conditional.
constexpr int kEstSplit = 128;
__global__ __launch_bounds__(256) void est_reduce_kernel(const float *__restrict__ ws, int32_t n_chunks, int32_t w,
                                                         int32_t slab0, int32_t batch, int32_t *est,
                                                         const int32_t *gate, int32_t gate_want) {
    if (gate && *gate != gate_want) return;
    __shared__ float red[256];
    const int slab = slab0 + blockIdx.y, tid = threadIdx.x;
    const int col = tid % w, sub = tid / w, nsub = 256 / w;      // w divides 256
    const float *base = ws + (size_t)slab * n_chunks * w;
    float m = 0.f;
    for (int64_t c = (int64_t)blockIdx.x * nsub + sub; c < n_chunks; c += (int64_t)gridDim.x * nsub)
        m = fmaxf(m, base[(size_t)c * w + col]);
    red[tid] = m;
    __syncthreads();
    if (tid < w) {
        for (int k = 1; k < nsub; ++k) m = fmaxf(m, red[tid + k * w]);
        const int q = slab * w + tid;
        if (q < batch && m > 0.f) atomicMax(&est[q], __float_as_int(m));
    }
}

// Convergence contract, decision number j (after final sweep variant j ran: the stage it closes may be the last).
// est_f[q] holds the relative size of the update that final sweep applied to the passage scores (max over the
// passages, float bits) -- MEASURED, not predicted: g = damping / (1 - damping) turns it into the error left.  While
// it exceeds tol for some query (and an extension stage is left) ctl[j] = 1: the boundary that closes the stage for
// real, extension stage j + 1 and final sweep variant j + 1 run (their launches are gated on ctl[j]), and est_f starts
// from zero for that sweep; otherwise ctl[j] stays 0, every later launch skips itself, est_f is what
// ppr8_finalize_kernel reports.  A decision whose predecessor stopped does nothing.
__global__ __launch_bounds__(256) void ppr8_decide_kernel(int32_t *est_f, const int32_t *__restrict__ flags,
                                                          int32_t batch, float g, float tol, int32_t j, int32_t e_max,
                                                          int32_t *ctl) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const bool alive = j == 0 || ctl[j - 1] == 1;
    float m = 0.f;
    if (alive)
        for (int q = tid; q < batch; q += 256)
            if (!(flags[q] & 1)) m = fmaxf(m, __int_as_float(est_f[q]));
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const bool go = alive && tol > 0.f && j < e_max && g * red[0] > tol;
    if (go)
        for (int q = tid; q < batch; q += 256) est_f[q] = 0;
    if (tid == 0 && alive) ctl[j] = go ? 1 : 0;
}

// Results of the contract per query: resid = g * (relative size of the final sweep's update of the passage scores),
// the sweeps that ran, flags bit 4 when tol > 0 and resid > tol, and the mass of the iterate that was computed.
__global__ void ppr8_finalize_kernel(const int32_t *__restrict__ est_f, int32_t *flags, int32_t batch, float g,
                                     float tol, int32_t iters, const int32_t *__restrict__ ctl, int32_t e_max,
                                     const double *__restrict__ mass_tab, int64_t tab_stride, double *sums,
                                     float *resid, int32_t *iters_used) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    int n_ext = 0;
    for (int j = 0; j < e_max; ++j) n_ext += ctl[j] == 1 ? 1 : 0;
    const bool fallback = (flags[q] & 1) != 0;
    const float r = fallback ? 0.f : g * __int_as_float(est_f[q]);
    resid[q] = r;
    iters_used[q] = fallback ? 0 : iters + p8_ext_sweeps(n_ext);
    if (mass_tab) sums[q] = mass_tab[(size_t)n_ext * tab_stride + q];
    if (tol > 0.f && r > tol) flags[q] |= kFlagNotConverged;
}

// Dynamic stage scales (HRAG_OPT_ACCEL), after the boundary that closed stage `stage` with cs' = dyn[stage + 1]:
//   M = max over the batch of |R_new| = (max of the wavefront slots and of the atomic word) / cs';
//   the NEXT boundary's residual is bounded by kappa M (kappa: the max-norm contraction of stage `stage + 1`), and the
//   stage after it grows its iterate by at most `growth`: dyn[stage + 2] = the power of two that maps kappa M growth to
//   <= 224 (half the e4m3 range, like the static chain).  M = 0 (nothing left) keeps the previous scale.
// seed = 1: write dyn[0] = cs0, dyn[1] = cs1 (the two scales known before anything was measured) and return.
// finalize = 0 (a boundary step that is launched per exchange group, every group but the last): the launch only folds
// its group's maximum into word[0]; the launch that follows the LAST group of the step finds the maximum of the whole
// batch there and turns it into the scale (a scale taken from one group's slots alone would saturate or starve the
// other groups' queries).
__global__ __launch_bounds__(256) void ppr8_next_scale_kernel(const float *__restrict__ ws, int32_t n_slots, int32_t *word,
                                                              float *dyn, int32_t stage, float kappa_growth, int32_t seed,
                                                              float cs0, float cs1, const int32_t *gate, int32_t gate_want,
                                                              int32_t finalize) {
    if (gate && *gate != gate_want) return;
    if (seed) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            dyn[0] = cs0; dyn[kP8DynInv + 0] = 1.0f / cs0;
            dyn[1] = cs1; dyn[kP8DynInv + 1] = 1.0f / cs1;
            word[0] = 0; word[1] = 0;
        }
        return;
    }
    // word[0]: running maximum (bits of a non-negative float: integer atomicMax is order-preserving and independent of
    // the arrival order); word[1]: blocks that have arrived -- the last one turns the maximum into the scale
    __shared__ float red[256];
    float m = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_slots; i += gridDim.x * 256) m = fmaxf(m, ws[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    if (red[0] > 0.f) __hip_atomic_fetch_max(&word[0], __float_as_int(red[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!finalize) return;
    const int before = __hip_atomic_fetch_add(&word[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before != (int)gridDim.x - 1) return;
    const float mq = __int_as_float(__hip_atomic_load(&word[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __hip_atomic_store(&word[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&word[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float cs_used = dyn[stage + 1];
    float cs = cs_used;                       // nothing measured: keep the scale
    const float bound = mq / cs_used * kappa_growth;
    if (bound > 0.f && bound < 3e38f) {
        int ex = (int)floorf(log2f(224.0f / bound));
        // floorf(log2f()) can be one off at exact powers of two: never let the bound exceed 224
        if (ldexpf(1.0f, ex) * bound > 224.0f) --ex;
        ex = ex < -60 ? -60 : (ex > 60 ? 60 : ex);
        cs = ldexpf(1.0f, ex);
    }
    dyn[stage + 2] = cs;
    dyn[kP8DynInv + stage + 2] = 1.0f / cs;
}

__global__ void ppr8_mask_seeds_kernel(const int32_t *__restrict__ seed_vtx, const int32_t *__restrict__ seed_cnt,
                                       int32_t batch, int64_t num_vertices, uint32_t *colmask) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = t / kMaxSeeds, j = t % kMaxSeeds;
    if (q >= batch || j >= seed_cnt[q]) return;
    const int64_t v = seed_vtx[q * kMaxSeeds + j];
    if (v < 0 || v >= num_vertices) return;
    atomicOr(&colmask[v >> 5], 1u << (v & 31));
}

// The pair kernel takes the slabs that come in adjacent pairs (groups of even width), ppr8_kernel an odd last slab.
// HRAG_P8_PAIR=0 keeps everything on ppr8_kernel (A/B measurements).
static bool p8_pair_enabled() {
    static const bool v = [] { const char *e = experiment_env("HRAG_P8_PAIR"); return !(e && e[0] == '0'); }();
    return v;
}

template <int MODE, int RIO, bool EST = false>
hrag_status sweep_mode(const Ppr8Args &a_in, bool main_only, hipStream_t s) {
    Ppr8Args a = a_in;
    if (a.m.n_chunks > 0 && p8_pair_enabled() && a.spg % 2 == 0 && a.slab0 % 2 == 0 && a.n_slabs >= 2) {
        Ppr8Args b = a;
        b.n_slabs = a.n_slabs & ~1;
        const int np = b.n_slabs / 2;
        b.wps = (a.wps == 4 && np % 4 == 0) ? 4 : (a.wps >= 2 && np % 2 == 0) ? 2 : 1;
        const unsigned ncg = (unsigned)ceil_div(a.m.n_chunks, 4 / b.wps);
        if (a.cg_per_xcd) b.cg_per_xcd = (int32_t)(round_up(ncg, 8) / 8);
        hipLaunchKernelGGL((ppr8_pair_kernel<MODE, RIO, EST>), dim3((unsigned)round_up(ncg, 8) * (unsigned)(np / b.wps)),
                           dim3(256), 0, s, b);
        HRAG_LAUNCH_CHECK();
        a.slab0 += b.n_slabs;
        a.n_slabs -= b.n_slabs;
        if (a.n_slabs == 0) return HRAG_OK;
    }
    if (a.m.n_chunks > 0) {
        // (86 VGPRs = 5 wavefronts per SIMD; 6 measured the same: the sweep is bandwidth-bound)
        Ppr8Args b = a;
        b.wps = (a.wps == 4 && a.n_slabs % 4 == 0) ? 4 : (a.wps >= 2 && a.n_slabs % 2 == 0) ? 2 : 1;
        const unsigned ncg = (unsigned)ceil_div(a.m.n_chunks, 4 / b.wps);
        if (a.cg_per_xcd) b.cg_per_xcd = (int32_t)(round_up(ncg, 8) / 8);
        hipLaunchKernelGGL((ppr8_kernel<MODE, RIO, EST>), dim3((unsigned)round_up(ncg, 8) * (unsigned)(a.n_slabs / b.wps)),
                           dim3(256), 0, s, b);
        HRAG_LAUNCH_CHECK();
    }
    (void)main_only;   // long rows are finished inside the sweep kernel (last-arriving segment)
    return HRAG_OK;
}

}  // namespace

static hrag_status sweep_dispatch(const Ppr8Args &a, int mode, bool main_only, hipStream_t s);

hrag_status launch_ppr8_sweep(const Ppr8Args &a, int mode, bool main_only, hipStream_t s) {
    if (a.n_slabs <= 0) return HRAG_OK;
    HRAG_TRY(sweep_dispatch(a, mode, main_only, s));
    if (a.est && mode == kP8ModeF)   // column maxima of the per-wavefront scratch
        return launch_est_reduce(a.est_ws, a.m.n_pchunks, 128, a.slab0, a.n_slabs, a.batch, a.est, a.gate, a.gate_want, s);
    return HRAG_OK;
}

hrag_status launch_ppr8_gather_replay(const Ppr8Args &a_in, hipStream_t s) {
    Ppr8Args b = a_in;
    if (!(b.m.n_chunks > 0 && b.spg % 2 == 0 && b.slab0 % 2 == 0 && b.n_slabs >= 2 && b.n_slabs % 2 == 0)) {
        set_error("gather replay: needs a slab-pair state (an even number of 128-query slabs, batch > 128)");
        return HRAG_EINVAL;
    }
    const int np = b.n_slabs / 2;
    b.wps = (a_in.wps == 4 && np % 4 == 0) ? 4 : (a_in.wps >= 2 && np % 2 == 0) ? 2 : 1;
    const unsigned ncg = (unsigned)ceil_div(b.m.n_chunks, 4 / b.wps);
    if (b.cg_per_xcd) b.cg_per_xcd = (int32_t)(round_up(ncg, 8) / 8);
    hipLaunchKernelGGL(ppr8_pair_replay_kernel, dim3((unsigned)round_up(ncg, 8) * (unsigned)(np / b.wps)), dim3(256), 0, s, b);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

static hrag_status sweep_dispatch(const Ppr8Args &a, int mode, bool main_only, hipStream_t s) {
    const int rio = a.rio & 3;
    switch (mode) {
        case kP8ModeC: return sweep_mode<kP8ModeC, 0>(a, main_only, s);
        case kP8ModeB0: return sweep_mode<kP8ModeB0, 0>(a, main_only, s);
        case kP8ModeB:
            if (rio == 0) return sweep_mode<kP8ModeB, 0>(a, main_only, s);
            if (rio == 2) return sweep_mode<kP8ModeB, 2>(a, main_only, s);
            if (rio == 3) return sweep_mode<kP8ModeB, 3>(a, main_only, s);
            break;
        case kP8ModeF:
            if (a.est) {
                if (rio == 0) return sweep_mode<kP8ModeF, 0, true>(a, main_only, s);
                if (rio == 1) return sweep_mode<kP8ModeF, 1, true>(a, main_only, s);
                break;
            }
            if (rio == 0) return sweep_mode<kP8ModeF, 0>(a, main_only, s);
            if (rio == 1) return sweep_mode<kP8ModeF, 1>(a, main_only, s);
            break;
        default: break;
    }
    set_error("bad ppr8 mode %d / residual form %d", mode, rio);
    return HRAG_EINVAL;
}

hrag_status launch_ppr8_init(const Ppr8Args &a, float c0_scale, hipStream_t s) {
    if (a.n_rows <= 0 || a.n_slabs <= 0) return HRAG_OK;
    dim3 grid((unsigned)ceil_div(a.n_rows * 8, 256), (unsigned)a.n_slabs);
    hipLaunchKernelGGL(ppr8_init_kernel, grid, dim3(256), 0, s, a, c0_scale);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_prior(const float *scores, int64_t ld, int64_t p_rows, const float *mn, const float *mx,
                              float passage_weight, const float *pinvdeg, const uint8_t *piso,
                              const int32_t *flags, int32_t batch, int32_t *zmax_bits, double *part,
                              float *zmax, double *mass, hipStream_t s) {
    HRAG_HIP_TRY(hipMemsetAsync(zmax_bits, 0, (size_t)batch * sizeof(int32_t), s));
    hipLaunchKernelGGL(ppr8_prior_kernel, dim3(kP8PriorSplit, (unsigned)batch), dim3(256), 0, s, scores, ld,
                       p_rows, mn, mx, passage_weight, pinvdeg, piso, flags, batch, zmax_bits, part);
    HRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ppr8_prior_final_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s, zmax_bits,
                       part, batch, zmax, mass);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_scale(const float *zmax, const double *mass, float passage_weight,
                              const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt,
                              const float *deg, const uint8_t *iso, int64_t num_vertices, const int32_t *flags,
                              int32_t batch, float damping, int32_t iters, float *qscale, double *sums,
                              hipStream_t s, int32_t n_tab, int64_t tab_stride) {
    hipLaunchKernelGGL(ppr8_scale_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s, zmax, mass,
                       passage_weight, seed_vtx, seed_w, seed_cnt, deg, iso, num_vertices, flags, batch, damping,
                       iters, qscale, sums, n_tab, tab_stride);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_est_reduce(const float *ws, int32_t n_chunks, int32_t w, int32_t slab0, int32_t n_slabs, int32_t batch,
                              int32_t *est, const int32_t *gate, int32_t gate_want, hipStream_t s) {
    if (n_chunks <= 0 || n_slabs <= 0) return HRAG_OK;
    if (!(w == 128 || w == 64 || w == 8 || w == 4 || w == 2 || w == 1)) {
        set_error("est_reduce: unsupported row width %d", w);
        return HRAG_EINVAL;
    }
    const unsigned split = (unsigned)std::min<int64_t>(kEstSplit, std::max<int64_t>(1, (int64_t)n_chunks * w / 4096));
    hipLaunchKernelGGL(est_reduce_kernel, dim3(split, (unsigned)n_slabs), dim3(256), 0, s, ws, n_chunks, w, slab0,
                       batch, est, gate, gate_want);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_decide(int32_t *est_f, const int32_t *flags, int32_t batch, float g, float tol, int32_t j,
                               int32_t e_max, int32_t *ctl, hipStream_t s) {
    hipLaunchKernelGGL(ppr8_decide_kernel, dim3(1), dim3(256), 0, s, est_f, flags, batch, g, tol, j, e_max, ctl);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_finalize(const int32_t *est_f, int32_t *flags, int32_t batch, float g, float tol,
                                 int32_t iters, const int32_t *ctl, int32_t e_max, const double *mass_tab,
                                 int64_t tab_stride, double *sums, float *resid, int32_t *iters_used, hipStream_t s) {
    hipLaunchKernelGGL(ppr8_finalize_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s, est_f, flags, batch,
                       g, tol, iters, ctl, e_max, mass_tab, tab_stride, sums, resid, iters_used);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_mask_seeds(const int32_t *seed_vtx, const int32_t *seed_cnt, int32_t batch,
                                   int64_t num_vertices, uint32_t *colmask, hipStream_t s) {
    const int total = batch * kMaxSeeds;
    hipLaunchKernelGGL(ppr8_mask_seeds_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, seed_vtx,
                       seed_cnt, batch, num_vertices, colmask);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr8_next_scale(const float *ws, int32_t n_slots, int32_t *word, float *dyn, int32_t stage,
                                   float kappa_growth, int32_t seed, float cs0, float cs1, const int32_t *gate,
                                   int32_t gate_want, hipStream_t s, bool finalize) {
    const unsigned blocks = seed ? 1u : (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, ceil_div(n_slots, 4096)));
    hipLaunchKernelGGL(ppr8_next_scale_kernel, dim3(blocks), dim3(256), 0, s, ws, n_slots, word, dyn, stage, kappa_growth, seed,
                       cs0, cs1, gate, gate_want, finalize ? 1 : 0);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
