// K4 -- row-wise top-k selection with the library ranking rule (score desc, index desc), fused
// with the row min / max that min_max_normalize needs.
//
// Replaces the full np.argsort(...)[::-1] of the reference wherever only a prefix is consumed:
//   rerank_facts      np.argsort(query_fact_scores)[-link_top_k:][::-1]   HippoRAG.py:1683-1688
//   run_ppr           np.argsort(doc_scores)[::-1] (top retrieval_top_k)   HippoRAG.py:1746, :503
//   dense_passage_retrieval  np.argsort(query_doc_scores)[::-1]            HippoRAG.py:1500
// and min_max_normalize (utils/misc_utils.py:130-139) for the selected entries.
//
// One 1024-thread workgroup per row, two streaming passes, everything else in LDS:
//   pass 1: every thread keeps its 2 largest 64-bit keys (ordered(score) << 32 | index; keys are
//           unique, so ties need no special casing) and the running min / max.  The k-th largest
//           of the 2048 thread-local maxima is a lower bound L of the true k-th key (at least k
//           elements are >= L).
//   pass 2: keys >= L are appended to an LDS candidate buffer (expected k..~1.3k entries),
//           bitonic-sorted descending, and the first k are written out.
// If the candidates overflow the buffer (an adversarial distribution, or simply a large k on a long row) the bound is
// refined by radix histograms over the key space (10 bits per streaming pass, round 6; bisection before) until they fit,
// and pass 2 is repeated; correctness never depends on luck.
#include <algorithm>

#include "common.h"

namespace hrag {
namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_CAP = 4096;  // LDS candidate capacity (32 KiB of keys)

__device__ __forceinline__ void bitonic_sort_desc(uint64_t *buf, int n_pow2, int tid) {
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n_pow2 >> 1); t += TK_THREADS) {
                const int i = ((t / stride) * (stride << 1)) + (t % stride);
                const int j = i + stride;
                const bool first_half = (i & size) == 0;
                const uint64_t a = buf[i], b = buf[j];
                if ((a < b) == first_half) {
                    buf[i] = b;
                    buf[j] = a;
                }
            }
            __syncthreads();
        }
    }
}

template <typename F>
__device__ __forceinline__ void for_each_in_row(const float *s, int64_t n, int tid, F f) {
    const bool vec = (reinterpret_cast<uintptr_t>(s) & 15) == 0;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        int64_t i = tid;
        // four independent 16-byte loads in flight per lane (one per iteration left a 1024-thread workgroup with 16 KB
        // outstanding: latency-bound at ~1/3 of the stream rate on 125k-long rows); the SET of elements a thread sees is
        // unchanged, and nothing downstream depends on their order
        for (; i + 3 * TK_THREADS < n4; i += 4 * TK_THREADS) {
            const float4 v0 = s4[i], v1 = s4[i + TK_THREADS], v2 = s4[i + 2 * TK_THREADS], v3 = s4[i + 3 * TK_THREADS];
            const float4 vv[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t b = (uint32_t)(4 * (i + u * TK_THREADS));
                f(vv[u].x, b);
                f(vv[u].y, b + 1);
                f(vv[u].z, b + 2);
                f(vv[u].w, b + 3);
            }
        }
        for (; i < n4; i += TK_THREADS) {
            const float4 v = s4[i];
            f(v.x, (uint32_t)(4 * i));
            f(v.y, (uint32_t)(4 * i + 1));
            f(v.z, (uint32_t)(4 * i + 2));
            f(v.w, (uint32_t)(4 * i + 3));
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += TK_THREADS) f(s[i], (uint32_t)i);
    } else {
        for (int64_t i = tid; i < n; i += TK_THREADS) f(s[i], (uint32_t)i);
    }
}

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float minmax_norm(float s, float mn, float mx) {
    const float range = mx - mn;
    return range == 0.f ? 1.f : __fdiv_rn(s - mn, range);  // misc_utils.py:130-139
}

// parts > 1: workgroup (row_in * parts + part) selects from the slice [part * plen, +plen) of input
// row row_in and writes its k candidates (global positions, raw values) as output row blockIdx.x;
// topk_merge_kernel then picks the final k.
__global__ __launch_bounds__(TK_THREADS) void row_topk_kernel(
    const float *__restrict__ scores, int64_t n_total, int64_t ld, int32_t k, int32_t idx_offset_in,
    int32_t norm, int32_t *__restrict__ idx_out, float *__restrict__ val_out,
    float *__restrict__ mn_out, float *__restrict__ mx_out, int32_t parts, int64_t plen) {
    __shared__ uint64_t buf[TK_CAP];
    __shared__ float red_mn[TK_THREADS / 64], red_mx[TK_THREADS / 64];
    __shared__ unsigned int s_count;

    const int tid = threadIdx.x;
    const int row = blockIdx.x;          // output row
    const int row_in = row / parts, part = row % parts;
    const float *s = scores + (size_t)row_in * ld + (size_t)part * plen;
    int64_t n = parts > 1 ? n_total - (int64_t)part * plen : n_total;
    n = n < 0 ? 0 : (parts > 1 && n > plen ? plen : n);
    const int32_t idx_offset = parts > 1 ? (int32_t)(part * plen) : idx_offset_in;
    const int kk = (int)((int64_t)k < n ? (int64_t)k : n);

    // ---- pass 1: thread-local top-2 + min / max
    uint64_t t0 = 0, t1 = 0;
    float mn = INFINITY, mx = -INFINITY;
    for_each_in_row(s, n, tid, [&](float v, uint32_t i) {
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        const uint64_t key = rank_key(v, i);
        if (key > t1) {
            if (key > t0) {
                t1 = t0;
                t0 = key;
            } else {
                t1 = key;
            }
        }
    });
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((tid & 63) == 0) {
        red_mn[tid >> 6] = mn;
        red_mx[tid >> 6] = mx;
    }
    buf[tid] = t0;
    buf[TK_THREADS + tid] = t1;
    __syncthreads();
    mn = red_mn[0];
    mx = red_mx[0];
    for (int w = 1; w < TK_THREADS / 64; ++w) {
        mn = fminf(mn, red_mn[w]);
        mx = fmaxf(mx, red_mx[w]);
    }
    if (tid == 0) {
        if (mn_out) mn_out[row] = mn;
        if (mx_out) mx_out[row] = mx;
    }
    if (kk == 0) {
        for (int j = tid; j < k; j += TK_THREADS) {
            idx_out[(size_t)row * k + j] = -1;
            val_out[(size_t)row * k + j] = 0.f;
        }
        return;
    }
    // A lower bound of the kk-th largest key without sorting the 2048 thread-local maxima (66 barrier stages, ~25 us --
    // most of the kernel on short rows): every wavefront finds the r-th largest of ITS 128 keys, r = ceil(kk / 16), by
    // rank counting over shuffles (keys are unique); the minimum over the 16 wavefronts has at least 16 r >= kk keys
    // above or at it.  (A wavefront with fewer than r real keys answers 0: everything is collected, and the refinement
    // below takes over if that does not fit.)
    __shared__ uint64_t wave_rth[TK_THREADS / 64];
    {
        const int r_target = (kk + TK_THREADS / 64 - 1) / (TK_THREADS / 64) - 1;     // 0-based rank inside the wavefront
        int ra = 0, rb = 0;
        for (int j = 0; j < 64; ++j) {
            const uint64_t oa = __shfl(t0, j, 64), ob = __shfl(t1, j, 64);
            ra += (oa > t0 ? 1 : 0) + (ob > t0 ? 1 : 0);
            rb += (oa > t1 ? 1 : 0) + (ob > t1 ? 1 : 0);
        }
        uint64_t mine = (ra == r_target && t0) ? t0 : ((rb == r_target && t1) ? t1 : 0);
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(mine, o, 64);
            mine = other > mine ? other : mine;
        }
        if ((tid & 63) == 0) wave_rth[tid >> 6] = mine;
    }
    __syncthreads();
    uint64_t thresh = wave_rth[0];
    for (int w = 1; w < TK_THREADS / 64; ++w) thresh = wave_rth[w] < thresh ? wave_rth[w] : thresh;
    __syncthreads();

    // ---- pass 2: collect candidates >= thresh (refine the bound first if they would not fit)
    // collect straight away (the usual case: k .. ~1.3 k candidates); only when they do NOT fit is the
    // bound refined (below) and the collection repeated -- one streaming pass less than counting first
    auto collect = [&](uint64_t t) -> unsigned int {
        if (tid == 0) s_count = 0;
        __syncthreads();
        for_each_in_row(s, n, tid, [&](float v, uint32_t i) {
            const uint64_t key = rank_key(v, i);
            if (key >= t) {
                const unsigned int slot = atomicAdd(&s_count, 1u);
                if (slot < TK_CAP) buf[slot] = key;
            }
        });
        __syncthreads();
        const unsigned int r = s_count;
        __syncthreads();
        return r;
    };
    unsigned int cnt = collect(thresh);
    if (cnt > TK_CAP) {
        // More than TK_CAP keys at or above the bound (large k on a long row: k = 2047 of 875 k in the index-time KNN puts
        // the bound at the weakest of all thread-local maxima, ~1 - 2 % of the row).  Until round 6 the exact kk-th key was
        // then found by bisection on the 64-bit key space -- up to 64 counting passes over the row, 35 of the 44 ms of a
        // 1000-query KNN block.  Radix refinement instead: a 1024-bin histogram of the keys inside [lo, hi] (LDS atomics; the
        // candidate buffer lends its memory), the bin that holds the kk-th largest is the next [lo, hi]; stop as soon as
        // count(key >= lo) fits the buffer -- 10 bits per pass and no need for the exact key: 2 - 3 passes.
        // Invariant: count(key > hi) = above < kk <= above + count(lo <= key <= hi).
        unsigned int *hist = reinterpret_cast<unsigned int *>(buf);
        static_assert(TK_THREADS == 1024, "one thread per histogram bin");
        uint64_t lo = thresh, hi = rank_key(mx, 0xffffffffu);
        unsigned int above = 0;
        for (;;) {
            const uint64_t width = hi - lo;
            int shift = 0;
            while ((width >> shift) >= 1024) ++shift;              // bins 0 .. width >> shift <= 1023
            hist[tid] = 0;
            __syncthreads();
            for_each_in_row(s, n, tid, [&](float v, uint32_t i) {
                const uint64_t key = rank_key(v, i);
                if (key >= lo && key <= hi) atomicAdd(&hist[(unsigned int)((key - lo) >> shift)], 1u);
            });
            __syncthreads();
            for (int o = 1; o < TK_THREADS; o <<= 1) {              // hist[b] := keys in the bins >= b
                const unsigned int add = tid + o < TK_THREADS ? hist[tid + o] : 0u;
                __syncthreads();
                hist[tid] += add;
                __syncthreads();
            }
            // the counts fall with b: exactly one bin b has above + hist[b] >= kk > above + hist[b + 1]
            const bool here = above + hist[tid] >= (unsigned int)kk;
            const bool next = tid + 1 < TK_THREADS && above + hist[tid + 1] >= (unsigned int)kk;
            if (here && !next) s_count = (unsigned int)tid;
            __syncthreads();
            const int b = (int)s_count;
            const unsigned int c_ge = above + hist[b];
            const unsigned int c_gt = b + 1 < TK_THREADS ? above + hist[b + 1] : above;
            __syncthreads();
            const uint64_t blo = lo + ((uint64_t)b << shift);
            if (c_ge <= (unsigned int)TK_CAP || shift == 0) {       // (shift 0: bins are single keys, c_ge = c_gt + 1 <= kk)
                thresh = blo;
                break;
            }
            const uint64_t span = ((uint64_t)1 << shift) - 1;
            above = c_gt;
            hi = hi - blo > span ? blo + span : hi;
            lo = blo;
        }
        cnt = collect(thresh);
    }
    cnt = cnt < (unsigned int)TK_CAP ? cnt : (unsigned int)TK_CAP;
    int p2 = 2;
    while (p2 < (int)cnt) p2 <<= 1;
    for (int j = (int)cnt + tid; j < p2; j += TK_THREADS) buf[j] = 0;
    __syncthreads();
    bitonic_sort_desc(buf, p2, tid);

    for (int j = tid; j < k; j += TK_THREADS) {
        int32_t idx = -1;
        float val = 0.f;
        if (j < kk) {
            const uint64_t key = buf[j];
            idx = (int32_t)(uint32_t)key + idx_offset;
            val = ordered_to_f32((uint32_t)(key >> 32));
            if (norm == kNormMinMax) val = minmax_norm(val, mn, mx);
        }
        idx_out[(size_t)row * k + j] = idx;
        val_out[(size_t)row * k + j] = val;
    }
}

__global__ __launch_bounds__(TK_THREADS) void row_minmax_kernel(const float *__restrict__ scores,
                                                                int64_t n, int64_t ld,
                                                                float *__restrict__ mn_out,
                                                                float *__restrict__ mx_out,
                                                                float *__restrict__ sum_out) {
    __shared__ float red_mn[TK_THREADS / 64], red_mx[TK_THREADS / 64];
    __shared__ double red_sum[TK_THREADS / 64];
    double sum = 0.0;  // only used to pick a scale (ppr16): accuracy is irrelevant
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const float *s = scores + (size_t)row * ld;
    float mn = INFINITY, mx = -INFINITY;
    for_each_in_row(s, n, tid, [&](float v, uint32_t) {
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        sum += (double)v;
    });
    mn = wave_min(mn);
    mx = wave_max(mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63) == 0) {
        red_mn[tid >> 6] = mn;
        red_mx[tid >> 6] = mx;
        red_sum[tid >> 6] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < TK_THREADS / 64; ++w) {
            mn = fminf(mn, red_mn[w]);
            mx = fmaxf(mx, red_mx[w]);
            sum += red_sum[w];
        }
        mn_out[row] = mn;
        mx_out[row] = mx;
        if (sum_out) sum_out[row] = (float)sum;
    }
}

// second level of the split selection: parts * k candidates per row (keys are unique), parts * k <= TK_CAP
__global__ __launch_bounds__(TK_THREADS) void topk_merge_kernel(
    const int32_t *__restrict__ cand_idx, const float *__restrict__ cand_val,
    const float *__restrict__ part_mn, const float *__restrict__ part_mx, int32_t parts, int32_t k,
    int64_t n_total, int32_t idx_offset, int32_t norm, int32_t *__restrict__ idx_out,
    float *__restrict__ val_out, float *__restrict__ mn_out, float *__restrict__ mx_out) {
    __shared__ uint64_t buf[TK_CAP];
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const int n_cand = parts * k;
    int p2 = 2;
    while (p2 < n_cand) p2 <<= 1;
    for (int j = tid; j < p2; j += TK_THREADS) {
        uint64_t key = 0;
        if (j < n_cand) {
            const int32_t ci = cand_idx[(size_t)row * n_cand + j];
            if (ci >= 0) key = rank_key(cand_val[(size_t)row * n_cand + j], (uint32_t)ci);
        }
        buf[j] = key;
    }
    float mn = INFINITY, mx = -INFINITY;
    for (int p = 0; p < parts; ++p) {
        mn = fminf(mn, part_mn[(size_t)row * parts + p]);
        mx = fmaxf(mx, part_mx[(size_t)row * parts + p]);
    }
    __syncthreads();
    bitonic_sort_desc(buf, p2, tid);
    if (tid == 0) {
        if (mn_out) mn_out[row] = mn;
        if (mx_out) mx_out[row] = mx;
    }
    const int kk = (int)((int64_t)k < n_total ? (int64_t)k : n_total);
    for (int j = tid; j < k; j += TK_THREADS) {
        int32_t idx = -1;
        float val = 0.f;
        if (j < kk) {
            const uint64_t key = buf[j];
            idx = (int32_t)(uint32_t)key + idx_offset;
            val = ordered_to_f32((uint32_t)(key >> 32));
            if (norm == kNormMinMax) val = minmax_norm(val, mn, mx);
        }
        idx_out[(size_t)row * k + j] = idx;
        val_out[(size_t)row * k + j] = val;
    }
}

}  // namespace

hrag_status launch_row_topk(const float *scores, int32_t batch, int64_t n, int64_t ld, int32_t k,
                            int32_t idx_offset, TopkNorm norm, int32_t *idx_out, float *val_out,
                            float *mn_out, float *mx_out, hipStream_t s, void *ws, size_t ws_bytes) {
    HRAG_REQUIRE(k >= 1 && k <= kTopkMax, "top-k k=%d outside [1, %d]", k, kTopkMax);
    HRAG_REQUIRE(n >= 0 && n < (int64_t)0xffffffffll, "row length %lld not supported", (long long)n);
    if (batch == 0) return HRAG_OK;
    // one workgroup per row leaves the chip idle for a handful of rows: split long rows
    int parts = 1;
    if (ws && batch < 64 && n >= 32768) {
        // the merge kernel bitonic-sorts parts * k keys (barrier stages ~ log^2): keep that at <= 1024 keys when k allows
        const int64_t merge_cap = k <= 512 ? 1024 : TK_CAP;
        parts = (int)std::min<int64_t>(std::min<int64_t>(merge_cap / k, ceil_div(n, 8192)), std::max(1, 256 / batch));
        parts = std::min(parts, 64);
        while (parts > 1 && (size_t)batch * parts * ((size_t)k * 8 + 8) > ws_bytes) --parts;
    }
    if (parts <= 1) {
        hipLaunchKernelGGL(row_topk_kernel, dim3((unsigned)batch), dim3(TK_THREADS), 0, s, scores, n, ld, k,
                           idx_offset, (int32_t)norm, idx_out, val_out, mn_out, mx_out, 1, (int64_t)0);
        HRAG_LAUNCH_CHECK();
        return HRAG_OK;
    }
    const int64_t plen = round_up(ceil_div(n, parts), 4);
    const size_t n_cand = (size_t)batch * parts * k;
    int32_t *c_idx = static_cast<int32_t *>(ws);
    float *c_val = reinterpret_cast<float *>(c_idx + n_cand);
    float *p_mn = c_val + n_cand;
    float *p_mx = p_mn + (size_t)batch * parts;
    hipLaunchKernelGGL(row_topk_kernel, dim3((unsigned)(batch * parts)), dim3(TK_THREADS), 0, s, scores, n, ld,
                       k, 0, (int32_t)kNormNone, c_idx, c_val, p_mn, p_mx, parts, plen);
    HRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)batch), dim3(TK_THREADS), 0, s, c_idx, c_val, p_mn, p_mx,
                       parts, k, n, idx_offset, (int32_t)norm, idx_out, val_out, mn_out, mx_out);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

hrag_status launch_row_minmax(const float *scores, int32_t batch, int64_t n, int64_t ld,
                              float *mn_out, float *mx_out, hipStream_t s, float *sum_out) {
    if (batch == 0) return HRAG_OK;
    hipLaunchKernelGGL(row_minmax_kernel, dim3((unsigned)batch), dim3(TK_THREADS), 0, s, scores, n, ld,
                       mn_out, mx_out, sum_out);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
