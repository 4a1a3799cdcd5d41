// K3s -- PPR sweep for latency-bound batches (B <= 8): "sparse matrix x a few vectors".
//
// Replaces igraph/PRPACK behind HippoRAG.run_ppr (reference src/hipporag/HippoRAG.py:1736-1743) for
// the B = 1 callers of the path: the run_ppr seam itself and retrieve_ircot, which calls
// retrieve([thought]) once per reasoning step (:526,539).
//
// At B = 1 the state x is V * 4 bytes (4 MB at 1 M vertices): it lives in every XCD's L2, the
// random gathers are L2 hits and the sweep is bound by the (col, val) stream -- the textbook SpMV
// roofline, nnz * 8 + V * 12 bytes.  Layout and matrix format follow that:
//   * state fp32 [V][BP], BP = batch padded to 1, 2, 4 or 8 (one gather = BP * 4 contiguous bytes);
//   * the SELL-8 matrix of ppr16.hip, read with ONE coalesced, non-temporal 512-byte load per
//     wavefront step; here every lane keeps its own (col, val) pair -- lane l of an 8-lane group
//     walks entries l, l+8, ... of the group's row -- and the 8 partial sums are combined with a
//     fixed xor-butterfly.  No atomics; bit-reproducible.
//   * rows longer than kSell8SegLen entries are the same segments as in ppr16.hip; the segment that arrives
//     last adds the partial sums up (fixed order) and finishes the row.
// v (passage prior + seed rows) is one fp32 array [tele_rows][BP] addressed through row_slot, like
// on the fp16 path; row_slot == nullptr means "dense v" (slot = vertex), used by hrag_ppr.
//
// Round 2 (hrag_retrieve with >= 16 sweeps; hrag_ppr keeps the plain fp32 iteration over all rows):
//   * TWO-STAGE fp16 STATE, the scheme of ppr16.hip at [V][BP] halfs: h <- f16(a P h + b v) for K/2 sweeps,
//     one residual sweep r = f16(64 ((a P h + b v) - h)) in fp32 arithmetic, then c <- f16(a P c + r),
//     x = h + c / 64.  At 1 M vertices x and y are 2 MB each instead of 4 MB: the gathers stay in the 4 MB
//     L2 (the fp32 state missed 30 % of them and the sweep ran on the miss traffic: 0.86 GB per sweep).
//   * the FIRST sweep gathers only where x_0 = v is non-zero (passage and seed columns, column bitmap);
//   * the LAST sweep runs over the passage rows only (a second SELL-8 matrix) and writes x = h + c / 64 in
//     fp32 there -- nothing else is read afterwards (HippoRAG.py:1745) -- and the normalisation is the closed
//     form of ppr8.hip (mass of the K-sweep iterate from sum(v) and the mass on isolated vertices).
#include "common.h"

namespace hrag {
namespace {

template <int BP>
struct XVec {
    float v[BP];
};

template <int BP>
__device__ __forceinline__ XVec<BP> ld_x(const float *x, unsigned col) {
    XVec<BP> r;
    const float *p = x + (size_t)col * BP;
    if constexpr (BP == 1) {
        r.v[0] = *p;
    } else if constexpr (BP == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        r.v[0] = t.x; r.v[1] = t.y;
    } else if constexpr (BP == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        const float4 t0 = reinterpret_cast<const float4 *>(p)[0], t1 = reinterpret_cast<const float4 *>(p)[1];
        r.v[0] = t0.x; r.v[1] = t0.y; r.v[2] = t0.z; r.v[3] = t0.w;
        r.v[4] = t1.x; r.v[5] = t1.y; r.v[6] = t1.z; r.v[7] = t1.w;
    }
    return r;
}

// the same for a fp16 state: BP halfs = 2 .. 16 contiguous bytes
template <int BP>
__device__ __forceinline__ XVec<BP> ld_x(const _Float16 *x, unsigned col) {
    typedef _Float16 hvec __attribute__((ext_vector_type(BP < 2 ? 2 : BP)));
    XVec<BP> r;
    const _Float16 *p = x + (size_t)col * BP;
    if constexpr (BP == 1) {
        r.v[0] = (float)*p;
    } else {
        const hvec t = *reinterpret_cast<const hvec *>(p);
#pragma unroll
        for (int b = 0; b < BP; ++b) r.v[b] = (float)t[b];
    }
    return r;
}

typedef int v2i_t __attribute__((ext_vector_type(2)));

// non-temporal buffer load (aux = 2): the stream must not evict x from L2; see ppr16.hip on why a
// buffer load and not a plain / __builtin_nontemporal_load
template <bool NT>
__device__ __forceinline__ int2 ld_pair_nt(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, NT ? 2 : 0);
    return make_int2(v.x, v.y);
}

constexpr float kSvHalfMax = 65504.f;
__device__ __forceinline__ _Float16 to_half(float v) { return (_Float16)fminf(fmaxf(v, -kSvHalfMax), kSvHalfMax); }

// Sweep modes.  fp32 state: kSvPlain only.  fp16 state: H = kSvPlain, then kSvResid, kSvCorr; kSvFinal = the last
// correction sweep, over the passage rows, writing x = h + c / cscale in fp32.
enum SvMode { kSvPlain = 0, kSvResid = 1, kSvCorr = 2, kSvFinal = 3 };

// lane gl (< BP) of the group finishes column gl of the row
// returns the relative size of the update when a.est asks for it (last sweep), else 0
template <int BP, int MODE, typename T>
__device__ __forceinline__ float sv_finish(const PprSvArgs &a, int row, int gl, float sum) {
    float er = 0.f;
    if (gl >= BP) return er;
    const size_t at = (size_t)row * BP + gl;
    if constexpr (MODE == kSvPlain || MODE == kSvResid) {
        float t = 0.f;
        const int64_t slot = a.row_slot ? (int64_t)a.row_slot[row] : (int64_t)row;
        if (slot >= 0) t = a.tele[(size_t)slot * BP + gl];
        float out = fmaf(a.alpha, sum, a.beta * t);
        if constexpr (MODE == kSvResid) out = (out - (float)static_cast<const T *>(a.x)[at]) * a.cscale;
        if constexpr (MODE == kSvPlain && sizeof(T) == 2) {
            if (a.omega != 1.f)   // Chebyshev step (HRAG_OPT_ACCEL): omega (plain result - prev) + prev
                out = fmaf(a.omega, out, (1.f - a.omega) * (a.prev ? (float)reinterpret_cast<const _Float16 *>(a.prev)[at] : 0.f));
        }
        if constexpr (MODE == kSvPlain && sizeof(T) == 4) {
            if (a.est && gl < a.batch)    // last sweep of the fp32 state: relative size of the update
                er = out > 0.f ? fabsf(out - static_cast<const float *>(a.x)[at]) / out : 0.f;
        }
        if constexpr (sizeof(T) == 2) static_cast<_Float16 *>(a.y)[at] = to_half(out);
        else static_cast<float *>(a.y)[at] = out;
    } else {
        float c = fmaf(a.alpha, sum, (float)reinterpret_cast<const _Float16 *>(a.aux16)[at]);
        if constexpr (MODE == kSvCorr) {
            if (a.omega != 1.f)
                c = fmaf(a.omega, c, (1.f - a.omega) * (a.prev ? (float)reinterpret_cast<const _Float16 *>(a.prev)[at] : 0.f));
            static_cast<_Float16 *>(a.y)[at] = to_half(c);
        } else {
            const float x = fmaf(c, 1.0f / a.cscale, (float)reinterpret_cast<const _Float16 *>(a.h16)[at]);
            a.xout[at] = x;
            if (a.est && gl < a.batch)    // relative size of this (last) sweep's update
                er = x > 0.f ? fabsf(c - (float)static_cast<const _Float16 *>(a.x)[at]) / (a.cscale * x) : 0.f;
        }
    }
    return er;
}

// MASK: the gathers of columns whose bit is clear in a.colmask are skipped (first sweep: x_0 = v is zero there)
template <int BP, bool NT, int MODE, typename T, bool MASK>
__global__ __launch_bounds__(256) void ppr_sv_kernel(const PprSvArgs a) {
    if (a.gate && *a.gate != a.gate_want) return;   // a conditional step the device decided not to run
    const int lane = threadIdx.x & 63;
    const int gl = lane & 7, grp = lane >> 3;
    const int chunk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (chunk >= a.n_chunks) return;
    const int2 meta = a.chunk_meta[chunk];
    const int n_steps = meta.y;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int2 *>(a.pairs), 0, (int)a.pairs_bytes, 0x00020000);
    const unsigned pbase = (unsigned)meta.x * 512u, poff = (unsigned)lane * 8u;
    const T *xs = static_cast<const T *>(a.x);
    float acc[BP];
#pragma unroll
    for (int b = 0; b < BP; ++b) acc[b] = 0.f;
    // pairs are read two steps ahead (the array is padded); two gathers are in flight per lane.
    // Padding entries are (col 0, val 0): an odd tail step adds 0 * x[0].
    int2 p0 = ld_pair_nt<NT>(prs, poff, pbase), p1 = ld_pair_nt<NT>(prs, poff + 512u, pbase);
    for (int s = 0; s < n_steps; s += 2) {
        const int2 p2 = ld_pair_nt<NT>(prs, poff + (unsigned)(s + 2) * 512u, pbase);
        const int2 p3 = ld_pair_nt<NT>(prs, poff + (unsigned)(s + 3) * 512u, pbase);
        const bool tail = s + 1 >= n_steps;      // wave-uniform
        XVec<BP> xa, xb;
        if constexpr (MASK) {
            const bool ona = (a.colmask[(unsigned)p0.x >> 5] >> (p0.x & 31)) & 1u;
            const bool onb = !tail && ((a.colmask[(unsigned)p1.x >> 5] >> (p1.x & 31)) & 1u);
#pragma unroll
            for (int b = 0; b < BP; ++b) xa.v[b] = xb.v[b] = 0.f;
            if (ona) xa = ld_x<BP>(xs, (unsigned)p0.x);
            if (onb) xb = ld_x<BP>(xs, (unsigned)p1.x);
        } else {
            xa = ld_x<BP>(xs, (unsigned)p0.x);
            xb = ld_x<BP>(xs, tail ? 0u : (unsigned)p1.x);
        }
        const float wa = __int_as_float(p0.y), wb = tail ? 0.f : __int_as_float(p1.y);
#pragma unroll
        for (int b = 0; b < BP; ++b) acc[b] = fmaf(wa, xa.v[b], acc[b]);
#pragma unroll
        for (int b = 0; b < BP; ++b) acc[b] = fmaf(wb, xb.v[b], acc[b]);
        p0 = p2;
        p1 = p3;
    }
    // fixed-order butterfly over the 8 lanes of the group: every lane ends with the row total
#pragma unroll
    for (int o = 1; o < 8; o <<= 1)
#pragma unroll
        for (int b = 0; b < BP; ++b) acc[b] += __shfl_xor(acc[b], o, 64);
    float mine = acc[0];
#pragma unroll
    for (int b = 1; b < BP; ++b) mine = gl == b ? acc[b] : mine;
    const int tgt = a.vrow[chunk * 8 + grp];
    const bool seg = tgt < 0 && tgt != kVrowNone;
    float er = 0.f;
    if (tgt >= 0) {
        er = sv_finish<BP, MODE, T>(a, tgt, gl, mine);
    } else if (seg && gl < BP) {
        // write-through (sc1): another XCD's reader must find the value in memory, not in this XCD's L2
        __hip_atomic_store(&a.partial[(size_t)(-(tgt + 1)) * BP + gl], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Long rows (> kSell8SegLen entries) arrive as segments in different wavefronts.  The segment that
    // arrives LAST (agent-scope counter) adds the partial sums up -- always in segment order, with the whole
    // wavefront -- and finishes the row: no second kernel, and the result does not depend on who came last.
    if (a.est) {   // wave-uniform.  The wavefront's maximum per column -> its slot of est_ws[chunk][BP] (plain stores;
                   // launch_est_reduce takes the column maxima: no atomics per row)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) er = fmaxf(er, __shfl_xor(er, o, 64));
        if (grp == 0 && gl < BP) a.est_ws[(size_t)chunk * BP + gl] = er;
    }
    if (__builtin_amdgcn_ballot_w64(seg) == 0) return;   // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's partial sums have left the CU
    int m = -1;
    bool last = false;
    if (seg && gl == 0) {
        m = a.seg_lrow[-(tgt + 1)];
        const int before = __hip_atomic_fetch_add(a.lcount + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = before == a.lrow_cnt[m] - 1;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(last);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int mm = __builtin_amdgcn_readlane(m, l);
        const int first = a.lrow_first[mm], cnt = a.lrow_cnt[mm];
        float tot = 0.f;
#pragma unroll
        for (int b = 0; b < BP; ++b) {
            float v = 0.f;
            for (int i = lane; i < cnt; i += 64)
                v += __hip_atomic_load(&a.partial[(size_t)(first + i) * BP + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            tot = gl == b ? v : tot;
        }
        if (grp == 0) {
            const float e2 = sv_finish<BP, MODE, T>(a, a.lrow_row[mm], gl, tot);
            if (a.est && gl < a.batch && e2 > 0.f) atomicMax(&a.est[gl], __float_as_int(e2));   // a handful of rows
        }
        if (lane == 0) __hip_atomic_store(a.lcount + mm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// x_0 = v  (fp32 or fp16 state)
template <int BP, typename T>
__global__ __launch_bounds__(256) void ppr_sv_init_kernel(const PprSvArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / BP;
    const int gl = (int)(t % BP);
    if (row >= a.num_vertices) return;
    float v = 0.f;
    const int64_t slot = a.row_slot ? (int64_t)a.row_slot[row] : row;
    if (slot >= 0) v = a.tele[(size_t)slot * BP + gl];
    if constexpr (sizeof(T) == 2) static_cast<_Float16 *>(a.y)[(size_t)row * BP + gl] = to_half(v);
    else static_cast<float *>(a.y)[(size_t)row * BP + gl] = v;
}

// The mass of the `iters`-sweep iterate in closed form (see ppr8_scale_kernel): M = sum over ALL teleport rows
// (passages with the seeds that are passages folded in, then the seed rows), S = the part on isolated vertices.
// Pass 1: grid (kSvMassSplit, batch) partial sums in double; pass 2: one thread per query adds them in a fixed order.
constexpr int kSvMassSplit = 64;
// tele: [slab][slab_rows][stride] with query b in slab b / stride, column b % stride (small batches: one slab)
__global__ __launch_bounds__(256) void ppr_sv_mass_kernel(const float *__restrict__ tele, int stride, int64_t slab_rows,
                                                          int64_t n_passages, int64_t tele_rows,
                                                          const uint8_t *__restrict__ piso, double *__restrict__ part) {
    __shared__ double rm[256], rs[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float *col = tele + (size_t)(b / stride) * (size_t)slab_rows * stride + (b % stride);
    double m = 0.0, si = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + tid; r < tele_rows; r += (int64_t)kSvMassSplit * 256) {
        const double v = (double)col[(size_t)r * stride];
        m += v;
        if (r < n_passages && piso[r]) si += v;
    }
    rm[tid] = m; rs[tid] = si;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rm[tid] += rm[tid + o]; rs[tid] += rs[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        part[((size_t)b * kSvMassSplit + blockIdx.x) * 2 + 0] = rm[0];
        part[((size_t)b * kSvMassSplit + blockIdx.x) * 2 + 1] = rs[0];
    }
}
__global__ void ppr_sv_mass_final_kernel(const double *__restrict__ part, const uint8_t *__restrict__ iso,
                                         const int32_t *__restrict__ passage_of_vertex,
                                         const int32_t *__restrict__ seed_vtx, const float *__restrict__ seed_w,
                                         const int32_t *__restrict__ seed_cnt, const float *__restrict__ qscale,
                                         int64_t num_vertices, int32_t batch, float damping, int32_t iters,
                                         double *sums) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double M = 0.0, S = 0.0;
    for (int i = 0; i < kSvMassSplit; ++i) {
        M += part[((size_t)b * kSvMassSplit + i) * 2 + 0];
        S += part[((size_t)b * kSvMassSplit + i) * 2 + 1];
    }
    const double qs = qscale ? (double)qscale[b] : 1.0;
    for (int j = 0; j < seed_cnt[b]; ++j) {       // isolated seed vertices that are not passages (own seed row)
        const int64_t v = seed_vtx[b * kMaxSeeds + j];
        if (v >= 0 && v < num_vertices && iso[v] && passage_of_vertex[v] < 0) S += (double)(seed_w[b * kMaxSeeds + j]) * qs;
    }
    const double al = (double)damping, be = (double)(1.0f - damping);
    double mk = M;
    for (int k = 0; k < iters; ++k) mk = al * (mk - (k == 0 ? S : be * S)) + be * M;
    sums[b] = mk;
}

__device__ __forceinline__ float sv_minmax_norm(float s, float mn, float mx) {
    const float range = mx - mn;
    return range == 0.f ? 1.f : __fdiv_rn(s - mn, range);   // misc_utils.py:130-139
}

// passage prior (HippoRAG.py:1626-1635) into tele[p][b]; columns >= batch and DPR-fallback queries are 0
template <int BP>
__global__ __launch_bounds__(256) void ppr_sv_tele_kernel(const float *scores, int64_t ld, int64_t n,
                                                          int32_t batch, const float *mn, const float *mx,
                                                          float weight, const int32_t *flags, float *tele,
                                                          const float *qscale) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
#pragma unroll
    for (int b = 0; b < BP; ++b) {
        float v = 0.f;
        if (b < batch && !(flags && (flags[b] & 1))) {
            v = sv_minmax_norm(scores[(size_t)b * ld + p], mn[b], mx[b]) * weight;
            if (qscale) v *= qscale[b];      // a power of two: exact
        }
        tele[(size_t)p * BP + b] = v;
    }
}

// reset_prob rows [B][V] -> v [V][BP] with NaN / negative -> 0 (HippoRAG.py:1735)
template <int BP>
__global__ __launch_bounds__(256) void ppr_sv_reset_kernel(const float *reset, int64_t n, int32_t batch,
                                                           float *tele) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int b = 0; b < BP; ++b) {
        float v = 0.f;
        if (b < batch) {
            v = reset[(size_t)b * n + i];
            v = (v != v || v < 0.f) ? 0.f : v;
        }
        tele[(size_t)i * BP + b] = v;
    }
}

constexpr int kSvSumBlocks = 256;

template <int BP>
__global__ __launch_bounds__(256) void ppr_sv_colsum_kernel(const float *x, int64_t n, double *partial) {
    __shared__ double red[256];
    double s[BP];
#pragma unroll
    for (int b = 0; b < BP; ++b) s[b] = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const XVec<BP> v = ld_x<BP>(x, (unsigned)r);
#pragma unroll
        for (int b = 0; b < BP; ++b) s[b] += (double)v.v[b];
    }
#pragma unroll
    for (int b = 0; b < BP; ++b) {
        red[threadIdx.x] = s[b];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * BP + b] = red[0];
        __syncthreads();
    }
}

template <int BP>
__global__ void ppr_sv_colsum_final_kernel(const double *partial, int nblk, double *sums) {
    const int b = threadIdx.x;
    if (b >= BP) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[(size_t)i * BP + b];
    sums[b] = s;
}

// out[b][i] = x[gather ? gather[i] : i][b] / sums[b]   (or the normalised DPR scores on fallback)
template <int BP>
__global__ __launch_bounds__(256) void ppr_sv_rows_kernel(const float *x, const int32_t *gather, int64_t n,
                                                          int32_t batch, const double *sums, float *out,
                                                          int64_t ld, const float *alt, int64_t alt_ld,
                                                          const float *mn, const float *mx,
                                                          const int32_t *flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const XVec<BP> v = ld_x<BP>(x, (unsigned)(gather ? gather[i] : (int32_t)i));
#pragma unroll
    for (int b = 0; b < BP; ++b) {
        if (b >= batch) break;
        float r;
        if (alt && flags && (flags[b] & 1)) {
            r = sv_minmax_norm(alt[(size_t)b * alt_ld + i], mn[b], mx[b]);
        } else {
            const double sm = sums[b];
            r = sm > 0.0 ? (float)((double)v.v[b] / sm) : 0.f;
        }
        out[(size_t)b * ld + i] = r;
    }
}

template <int BP, int MODE, typename T, bool MASK>
hrag_status sv_sweep_one(const PprSvArgs &a, bool main_only, hipStream_t s) {
    if (a.n_chunks > 0) {
        const dim3 grid((unsigned)ceil_div(a.n_chunks, 4));
        if (a.nt) hipLaunchKernelGGL((ppr_sv_kernel<BP, true, MODE, T, MASK>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ppr_sv_kernel<BP, false, MODE, T, MASK>), grid, dim3(256), 0, s, a);
        HRAG_LAUNCH_CHECK();
        if (a.est) HRAG_TRY(launch_est_reduce(a.est_ws, a.n_chunks, BP, 0, 1, a.batch, a.est, a.gate, a.gate_want, s));
    }
    (void)main_only;   // long rows are finished inside the sweep kernel (last-arriving segment)
    return HRAG_OK;
}

template <int BP>
hrag_status sv_sweep(const PprSvArgs &a, bool main_only, hipStream_t s) {
    const bool mask = a.colmask != nullptr;
    if (!a.half_state) {
        HRAG_REQUIRE(a.mode == kSvPlain, "the fp32 small-batch state only has the plain sweep");
        return mask ? sv_sweep_one<BP, kSvPlain, float, true>(a, main_only, s)
                    : sv_sweep_one<BP, kSvPlain, float, false>(a, main_only, s);
    }
    switch (a.mode) {
        case kSvPlain:
            return mask ? sv_sweep_one<BP, kSvPlain, _Float16, true>(a, main_only, s)
                        : sv_sweep_one<BP, kSvPlain, _Float16, false>(a, main_only, s);
        case kSvResid: return sv_sweep_one<BP, kSvResid, _Float16, false>(a, main_only, s);
        case kSvCorr: return sv_sweep_one<BP, kSvCorr, _Float16, false>(a, main_only, s);
        case kSvFinal: return sv_sweep_one<BP, kSvFinal, _Float16, false>(a, main_only, s);
        default: set_error("bad small-batch sweep mode %d", a.mode); return HRAG_EINVAL;
    }
}

}  // namespace

#define HRAG_DISPATCH_BP(bp, CALL)                                   \
    switch (bp) {                                                    \
        case 1: CALL(1); break;                                      \
        case 2: CALL(2); break;                                      \
        case 4: CALL(4); break;                                      \
        case 8: CALL(8); break;                                      \
        default:                                                     \
            set_error("unsupported small-batch width %d", (int)(bp)); \
            return HRAG_EINVAL;                                      \
    }

hrag_status launch_ppr_sv_sweep(const PprSvArgs &a, int bp, bool main_only, hipStream_t s) {
#define CALL(BP) return sv_sweep<BP>(a, main_only, s)
    HRAG_DISPATCH_BP(bp, CALL)
#undef CALL
    return HRAG_OK;
}

hrag_status launch_ppr_sv_init(const PprSvArgs &a, int bp, hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(a.num_vertices * bp, 256));
#define CALL(BP)                                                                                  \
    if (a.half_state) hipLaunchKernelGGL((ppr_sv_init_kernel<BP, _Float16>), grid, dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((ppr_sv_init_kernel<BP, float>), grid, dim3(256), 0, s, a)
    HRAG_DISPATCH_BP(bp, CALL)
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr_sv_tele(const float *scores, int64_t ld, int64_t n, int32_t batch, const float *mn,
                               const float *mx, float weight, const int32_t *flags, float *tele, int bp,
                               hipStream_t s, const float *qscale) {
    if (n == 0) return HRAG_OK;
    const dim3 grid((unsigned)ceil_div(n, 256));
#define CALL(BP) hipLaunchKernelGGL(ppr_sv_tele_kernel<BP>, grid, dim3(256), 0, s, scores, ld, n, batch, mn, mx, weight, flags, tele, qscale)
    HRAG_DISPATCH_BP(bp, CALL)
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr_sv_reset(const float *reset, int64_t n, int32_t batch, float *tele, int bp,
                                hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(n, 256));
#define CALL(BP) hipLaunchKernelGGL(ppr_sv_reset_kernel<BP>, grid, dim3(256), 0, s, reset, n, batch, tele)
    HRAG_DISPATCH_BP(bp, CALL)
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr_sv_colsum(const float *x, int64_t n, int bp, double *partial, double *sums,
                                 hipStream_t s) {
#define CALL(BP)                                                                                         \
    hipLaunchKernelGGL(ppr_sv_colsum_kernel<BP>, dim3(kSvSumBlocks), dim3(256), 0, s, x, n, partial);      \
    hipLaunchKernelGGL(ppr_sv_colsum_final_kernel<BP>, dim3(1), dim3(64), 0, s, partial, kSvSumBlocks, sums)
    HRAG_DISPATCH_BP(bp, CALL)
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr_sv_mass(const float *tele, int stride, int64_t slab_rows, int64_t n_passages, int64_t tele_rows,
                               const uint8_t *piso, const uint8_t *iso, const int32_t *passage_of_vertex,
                               const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt, const float *qscale,
                               int64_t num_vertices, int32_t batch, float damping, int32_t iters, double *part,
                               double *sums, hipStream_t s) {
    hipLaunchKernelGGL(ppr_sv_mass_kernel, dim3(kSvMassSplit, (unsigned)batch), dim3(256), 0, s, tele, stride, slab_rows,
                       n_passages, tele_rows, piso, part);
    HRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ppr_sv_mass_final_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s, part, iso,
                       passage_of_vertex, seed_vtx, seed_w, seed_cnt, qscale, num_vertices, batch, damping, iters, sums);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_ppr_sv_rows(const float *x, const int32_t *gather, int64_t n, int32_t batch,
                               const double *sums, float *out, int64_t ld, const float *alt, int64_t alt_ld,
                               const float *mn, const float *mx, const int32_t *flags, int bp,
                               hipStream_t s) {
    if (n == 0) return HRAG_OK;
    const dim3 grid((unsigned)ceil_div(n, 256));
#define CALL(BP) hipLaunchKernelGGL(ppr_sv_rows_kernel<BP>, grid, dim3(256), 0, s, x, gather, n, batch, sums, out, ld, alt, alt_ld, mn, mx, flags)
    HRAG_DISPATCH_BP(bp, CALL)
#undef CALL
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
