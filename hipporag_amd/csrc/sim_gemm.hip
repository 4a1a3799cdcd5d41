// K1 -- cosine-similarity scores S[b][m] = sum_k Q[b][k] * E[m][k] on the matrix cores (bf16, or IEEE
// fp16 for BASELINE configs[4]: `fp16 embeddings`; same kernel, v_mfma_f32_16x16x32_f16).
//
// Replaces np.dot(self.fact_embeddings, q.T) / np.dot(self.passage_embeddings, q.T)
// (reference src/hipporag/HippoRAG.py:1459, :1496; StandardRAG.py:422) for a batch of B queries.
// Embeddings are unit-norm rows rounded to bf16; products are exact in fp32, accumulation is fp32
// (v_mfma_f32_16x16x32_bf16), so the only difference to an fp64 dot product is summation order.
//
// Both operands are K-contiguous ("B^T" form): A = E tile (MFMA rows i <-> embedding rows m),
// B = Q tile (MFMA cols j <-> queries b).  Each lane feeds 8 consecutive k of one row to the MFMA
// for A and B alike, so the result does not depend on how the hardware numbers k inside a
// fragment.  C/D layout (cdna_hip_programming.md section 3): col = lane & 15, row = 4*(lane>>4)+reg,
// i.e. a lane owns 4 consecutive embedding rows of one query -> one float4 store into S[b][m..m+3].
//
// Tile: 128 embedding rows x BN queries per 256-thread workgroup, BK = 64 (one full 128-byte line
// of every row per step), global -> registers -> LDS with the next tile's loads issued before the
// MFMA block (T14 split), LDS rows padded to 144 bytes (conflict-free ds_read_b128 for 16 rows).
// The embedding matrix is streamed from HBM once: the BN-tiles of one row tile are adjacent in
// dispatch order so that they meet in the Infinity Cache.
#include <cstdlib>

#include "common.h"

namespace hrag {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// one 16x16x32 MFMA on 16-byte fragments of bf16 (F16 = false) or IEEE fp16 (F16 = true) elements
template <bool F16>
__device__ __forceinline__ f32x4 mfma32(const uint4 &ua, const uint4 &ub, f32x4 acc) {
    if constexpr (F16) {
        f16x8 a, b;
        __builtin_memcpy(&a, &ua, 16);
        __builtin_memcpy(&b, &ub, 16);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    } else {
        bf16x8 a, b;
        __builtin_memcpy(&a, &ua, 16);
        __builtin_memcpy(&b, &ub, 16);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
}

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int LDS_LD = BK + 8;  // bf16 elements per LDS row (144 bytes)

// TILEMAX: instead of the scores, write per (row tile, query) the max and min score of the tile
// (tmax / tmin: [n_tiles_m][batch]) -- pass 1 of the fused fact top-k below.
template <int BN, int WM, int WN, bool TILEMAX = false, bool F16 = false>
__global__ __launch_bounds__(256) void sim_gemm_kernel(const uint16_t *__restrict__ emb,
                                                       int64_t rows, int32_t dim,
                                                       const uint16_t *__restrict__ q,
                                                       int32_t batch, float *__restrict__ out,
                                                       int64_t ld, int32_t n_tiles_n, int32_t accumulate,
                                                       float *__restrict__ tmax = nullptr,
                                                       float *__restrict__ tmin = nullptr) {
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    constexpr int MI = BM / (WM * 16);
    constexpr int NJ = BN / (WN * 16);
    constexpr int A_PASSES = BM / 32;
    constexpr int B_PASSES = (BN + 31) / 32;
    __shared__ __attribute__((aligned(16))) uint16_t As[BM][LDS_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BN][LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin, so ids 8 apart run back to
    // back on one XCD; they get the n_tiles_n query tiles of the SAME row tile, whose embedding rows the
    // later ones then find in that XCD's L2 (PMC: the fact GEMM fetched 2.7 GB for a 1.34 GB matrix
    // when the two query tiles of a row tile sat on neighbouring XCDs).
    const int64_t tile = blockIdx.x;
    const int64_t jj = tile >> 3;
    const int nt = (int)(jj % n_tiles_n);
    const int64_t mt = (jj / n_tiles_n) * 8 + (tile & 7);
    if (mt * BM >= rows) return;     // the grid is padded to a multiple of 8 row tiles
    const int64_t m0 = mt * BM;
    const int b0 = nt * BN;

    const int ld_row = tid >> 3;        // 0..31
    const int ld_chunk = (tid & 7) * 8; // element offset inside the BK slice

    uint4 ra[A_PASSES], rb[B_PASSES];
    auto load_tile = [&](int k0) {
        const int kk = k0 + ld_chunk;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            const int64_t gm = m0 + ld_row + 32 * p;
            ra[p] = make_uint4(0, 0, 0, 0);
            if (gm < rows && kk < dim)
                ra[p] = *reinterpret_cast<const uint4 *>(emb + (size_t)gm * dim + kk);
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int r = ld_row + 32 * p;
            const int gb = b0 + r;
            rb[p] = make_uint4(0, 0, 0, 0);
            if (r < BN && gb < batch && kk < dim)
                rb[p] = *reinterpret_cast<const uint4 *>(q + (size_t)gb * dim + kk);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            *reinterpret_cast<uint4 *>(&As[ld_row + 32 * p][ld_chunk]) = ra[p];
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int r = ld_row + 32 * p;
            if (r < BN) *reinterpret_cast<uint4 *>(&Bs[r][ld_chunk]) = rb[p];
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    for (int k0 = 0; k0 < dim; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < dim) load_tile(k0 + BK);  // in flight while the MFMAs below run
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            uint4 a[MI], b[NJ];
            const int kcol = s * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const uint4 *>(&As[(wm * MI + i) * 16 + (lane & 15)][kcol]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const uint4 *>(&Bs[(wn * NJ + j) * 16 + (lane & 15)][kcol]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<F16>(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }

    if constexpr (TILEMAX) {
        // lane: 4 * MI rows of NJ queries; lanes l, l+16, l+32, l+48 share a query; the WM wavefronts
        // with the same wn share it too (LDS).  Rows beyond `rows` (zero-filled) are excluded.
        __shared__ float red_mx[WM][BN], red_mn[WM][BN];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float mx = -INFINITY, mn = INFINITY;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int64_t m = m0 + (wm * MI + i) * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) {
                        mx = fmaxf(mx, acc[i][j][r]);
                        mn = fminf(mn, acc[i][j][r]);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            mn = fminf(mn, __shfl_xor(mn, 16, 64));
            mn = fminf(mn, __shfl_xor(mn, 32, 64));
            if (lane < 16) {
                red_mx[wm][(wn * NJ + j) * 16 + lane] = mx;
                red_mn[wm][(wn * NJ + j) * 16 + lane] = mn;
            }
        }
        __syncthreads();
        if (tid < BN && b0 + tid < batch) {
            float mx = red_mx[0][tid], mn = red_mn[0][tid];
#pragma unroll
            for (int w = 1; w < WM; ++w) {
                mx = fmaxf(mx, red_mx[w][tid]);
                mn = fminf(mn, red_mn[w][tid]);
            }
            tmax[(size_t)mt * batch + b0 + tid] = mx;
            tmin[(size_t)mt * batch + b0 + tid] = mn;
        }
        return;
    }
    // epilogue: lane owns rows m..m+3 of query gb for every (i, j) fragment
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + (wm * MI + i) * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gb = b0 + (wn * NJ + j) * 16 + (lane & 15);
            if (gb >= batch || m >= rows) continue;
            float *dst = out + (size_t)gb * ld + m;
            f32x4 v = acc[i][j];
            if (accumulate) {   // multi-pass products (bf16x3 KNN): out += this pass
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) v[r] += dst[r];
            }
            if (m + 3 < rows && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int r = 0; r < 4; ++r)
                    if (m + r < rows) dst[r] = v[r];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fused fact top-k (get_fact_scores + rerank_facts' argsort prefix, HippoRAG.py:1427-1465, :1683-1688)
// without materialising the [B, F] score matrix:
//   pass 1  sim_gemm_kernel<.., TILEMAX>: max / min score of every 128-row tile, per query;
//   pass 2  tile_select_kernel: the k tiles with the largest maxima (key = score desc, tile desc).
//           The k largest scores of a row live in those tiles: the k best tile maxima are k distinct
//           scores >= the k-th best maximum t_k, hence the k-th largest score is >= t_k, and any
//           score >= t_k sits in a tile whose maximum is >= t_k.  Ties follow the ranking rule
//           (larger index first = larger tile first).
//   pass 3  tile_rescore_kernel: recompute the k x 128 scores of those tiles with the SAME MFMA
//           fragments in the SAME k order as pass 1 (bit-identical values), exact top-k of them,
//           min-max normalisation with the global min / max from pass 1.
// Output is bit-identical to launch_sim_gemm + launch_row_topk (tests/test_gpu_parity.py).
constexpr int kFusedMaxK = 16;

__device__ __forceinline__ uint64_t block_max_u64(uint64_t v, uint64_t *red, int tid) {
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t u = __shfl_xor((unsigned long long)v, o, 64);
        v = u > v ? u : v;
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    uint64_t r = red[0];
    for (int w = 1; w < 4; ++w) r = red[w] > r ? red[w] : r;
    return r;
}

// One workgroup per query.  tmax is [tile][batch], so a query's maxima are a strided column: they are
// read ONCE into registers (kSelRegs per thread covers 8192 tiles = 1M rows; beyond that the rounds
// re-read memory) and the k selection rounds run on the registers.
constexpr int kSelRegs = 32;
constexpr int kSelRecInts = 32 + 16 * 128 * 2;   // = kSelRec below: ints per query record of the `sel` workspace
__global__ __launch_bounds__(256) void tile_select_kernel(const float *__restrict__ tmax,
                                                          const float *__restrict__ tmin, int32_t n_tiles,
                                                          int32_t batch, int32_t k, int32_t *__restrict__ sel,
                                                          float *__restrict__ mn_out, float *__restrict__ mx_out,
                                                          float cut, int32_t *__restrict__ overflow) {
    // cut: tiles whose maximum is below it are not selected (their list entries are -1; the list is in descending order, so
    // the selected tiles stay a prefix); overflow != nullptr: overflow[b] = a (k + 1)-th tile reaches the cut as well
    // (the thresholded KNN: the caller takes the dense path for that query).  cut = -inf, overflow = nullptr: the plain top-k.
    __shared__ uint64_t red[4];
    __shared__ float redf[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool in_regs = n_tiles <= 256 * kSelRegs;
    uint64_t keys[kSelRegs];
    float mn = INFINITY;
#pragma unroll
    for (int i = 0; i < kSelRegs; ++i) {
        const int t = tid + 256 * i;
        keys[i] = 0;
        if (in_regs && t < n_tiles) {
            keys[i] = rank_key(tmax[(size_t)t * batch + b], (uint32_t)t);
            mn = fminf(mn, tmin[(size_t)t * batch + b]);
        }
    }
    if (!in_regs)
        for (int t = tid; t < n_tiles; t += 256) mn = fminf(mn, tmin[(size_t)t * batch + b]);
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
    if ((tid & 63) == 0) redf[tid >> 6] = mn;
    __syncthreads();
    mn = fminf(fminf(redf[0], redf[1]), fminf(redf[2], redf[3]));
    uint64_t prev = ~0ull;
    for (int r = 0; r < k + (overflow ? 1 : 0); ++r) {
        uint64_t best = 0;   // keys are > 0: ordered(x) of any non-NaN float is >= 0x00800000
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < kSelRegs; ++i)
                if (keys[i] < prev && keys[i] > best) best = keys[i];
        } else {
            for (int t = tid; t < n_tiles; t += 256) {
                const uint64_t key = rank_key(tmax[(size_t)t * batch + b], (uint32_t)t);
                if (key < prev && key > best) best = key;
            }
        }
        best = block_max_u64(best, red, tid);
        const bool keep = best && !(ordered_to_f32((uint32_t)(best >> 32)) < cut);      // (a NaN maximum is kept, as without a cut)
        if (r == k) {                                   // the extra round of the thresholded form
            if (tid == 0) overflow[b] = keep ? 1 : 0;
            break;
        }
        if (tid == 0) {
            sel[(size_t)b * kSelRecInts + r] = keep ? (int32_t)(uint32_t)best : -1;
            if (r == 0) {
                mx_out[b] = best ? ordered_to_f32((uint32_t)(best >> 32)) : -INFINITY;
                mn_out[b] = mn;
            }
        }
        if (!keep) {        // the maxima come in descending order: nothing further reaches the cut (or nothing is left)
            if (tid == 0) {
                for (int rr = r + 1; rr < k; ++rr) sel[(size_t)b * kSelRecInts + rr] = -1;
                if (overflow) overflow[b] = 0;
            }
            break;
        }
        prev = best;
    }
}

// Per-query record in the `sel` workspace (ints): [0, 16) the selected tiles, [16] the arrival counter of pass 3
// (zero between launches), then kFusedMaxK * BM 64-bit candidate keys.  The record stride does not depend on the
// batch, so the counters zeroed at allocation stay where every launch expects them.
constexpr int kSelRec = 32 + kFusedMaxK * BM * 2;

// Wide batches (the index-time KNN: 4096 queries x 16 tiles over 6.8 k tiles; round 6): many (tile, query) pairs name the
// SAME tile, and pass 3 read it once per pair -- 38.7 GB per launch at the synonymy call's shape, 6.9 ms beside a 16 ms
// GEMM (profiles/r06w_knn_*.json).  The pairs are bucketed by tile (histogram, one-workgroup scan, scatter: three small
// launches; the order INSIDE a bucket is whatever the atomics give, which no result depends on) and pass 3 becomes one
// workgroup per (tile, up to 16 of its queries): the tile is read once per 16 pairs and the 16 columns of the MFMA carry
// 16 different queries instead of 16 copies of one (tile_rescore_grouped_kernel).  pair id = b * k + r.
constexpr int kPairChunk = 16;
// "No tile" entries (id -1: fewer tiles than k, or -- the thresholded form -- tiles below the cut: nearly all of them in the
// index-time KNN) take no part: a query's arrival counter waits for its VALID tiles only, and a query with none gets its
// empty result row here.
__global__ __launch_bounds__(256) void pair_hist_kernel(const int32_t *__restrict__ rec, int32_t n_pairs, int32_t k,
                                                        int32_t *__restrict__ hist, int32_t *__restrict__ idx_out,
                                                        float *__restrict__ val_out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const int b = p / k, r = p % k;
    const int t = rec[(size_t)b * kSelRecInts + r];
    if (t >= 0) {
        atomicAdd(hist + t, 1);
    } else if (r == 0) {                 // the selected tiles are a prefix of the list: none at all
        for (int j = 0; j < k; ++j) {
            idx_out[(size_t)b * k + j] = -1;
            val_out[(size_t)b * k + j] = 0.f;
        }
    }
}
// one workgroup, in place over n buckets: hist[i] (count) -> first position of bucket i; chunk[i] = number of 16-pair
// chunks of the buckets before i, chunk[n] = all chunks
__global__ __launch_bounds__(1024) void pair_scan_kernel(int32_t *__restrict__ hist, int32_t *__restrict__ chunk, int32_t n) {
    __shared__ int32_t part[1024], partc[1024];
    const int tid = threadIdx.x;
    const int seg = (n + 1023) / 1024, lo = min(tid * seg, n), hi = min(lo + seg, n);
    int32_t sum = 0, sumc = 0;
    for (int i = lo; i < hi; ++i) {
        sum += hist[i];
        sumc += (hist[i] + kPairChunk - 1) / kPairChunk;
    }
    part[tid] = sum;
    partc[tid] = sumc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int32_t v = tid >= o ? part[tid - o] : 0, vc = tid >= o ? partc[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        partc[tid] += vc;
        __syncthreads();
    }
    int32_t run = part[tid] - sum, runc = partc[tid] - sumc;
    for (int i = lo; i < hi; ++i) {
        const int32_t c = hist[i];
        hist[i] = run;
        chunk[i] = runc;
        run += c;
        runc += (c + kPairChunk - 1) / kPairChunk;
    }
    if (tid == 1023) chunk[n] = partc[1023];
}
// cursor[t] (= first position of bucket t) advances to the END of bucket t
__global__ __launch_bounds__(256) void pair_scatter_kernel(const int32_t *__restrict__ rec, int32_t n_pairs, int32_t k,
                                                           int32_t *__restrict__ cursor, int32_t *__restrict__ order) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const int t = rec[(size_t)(p / k) * kSelRecInts + (p % k)];
    if (t >= 0) order[atomicAdd(cursor + t, 1)] = p;
}

// The exact top-k of one query's k x 128 candidate keys (written by the workgroups that rescored its tiles), by the
// workgroup that arrived last at the query's counter: shared by both forms of pass 3.
__device__ __forceinline__ void select_from_candidates(int32_t *my, int b, int32_t k, uint64_t *cand, uint64_t *red,
                                                       const float *__restrict__ mn_in, const float *__restrict__ mx_in,
                                                       int32_t idx_offset, int32_t normalize, int32_t *__restrict__ idx_out,
                                                       float *__restrict__ val_out, int tid) {
    unsigned long long *gc = reinterpret_cast<unsigned long long *>(my + 32);
    int n_cand = 0;
    for (int i = 0; i < k; ++i) n_cand += my[i] >= 0 ? BM : 0;   // selected tiles are a prefix of the list
    for (int i = tid; i < n_cand; i += 256)
        cand[i] = __hip_atomic_load(gc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(my + 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const float mn = mn_in[b], mx = mx_in[b];
    uint64_t prev = ~0ull;
    for (int rr = 0; rr < k; ++rr) {
        uint64_t best = 0;
        for (int i = tid; i < n_cand; i += 256) {
            const uint64_t key = cand[i];
            if (key < prev && key > best) best = key;
        }
        best = block_max_u64(best, red, tid);
        if (tid == 0) {
            int32_t idx = -1;
            float val = 0.f;
            if (best) {
                idx = (int32_t)(uint32_t)best + idx_offset;
                val = ordered_to_f32((uint32_t)(best >> 32));
                if (normalize) {
                    const float range = mx - mn;
                    val = range == 0.f ? 1.f : __fdiv_rn(val - mn, range);   // misc_utils.py:130-139
                }
            }
            idx_out[(size_t)b * k + rr] = idx;
            val_out[(size_t)b * k + rr] = val;
        }
        prev = best ? best : 0;
    }
}

// Pass 3, one workgroup per (selected tile, query) -- round 1 ran the k tiles of a query one after the other in one
// workgroup, a chain of ~240 dependent loads (0.10 ms at cfg 3, 69 us at cfg 2); now the k tiles are k workgroups, and
// the one that arrives LAST (agent-scope counter; candidate keys travel as sc1 stores / loads: the workgroups sit on
// different XCDs) selects the exact top-k of the k x 128 keys.  Same MFMA fragments in the same k order as pass 1.
template <bool F16>
__global__ __launch_bounds__(256) void tile_rescore_kernel(const uint16_t *__restrict__ emb, int64_t rows,
                                                           int32_t dim, const uint16_t *__restrict__ q,
                                                           int32_t *__restrict__ rec,
                                                           const float *__restrict__ mn_in,
                                                           const float *__restrict__ mx_in, int32_t k,
                                                           int32_t idx_offset, int32_t normalize,
                                                           int32_t *__restrict__ idx_out,
                                                           float *__restrict__ val_out) {
    __shared__ uint64_t cand[kFusedMaxK * BM];
    __shared__ uint64_t red[4];
    __shared__ int s_last;
    const int r = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *my = rec + (size_t)b * kSelRec;
    unsigned long long *gc = reinterpret_cast<unsigned long long *>(my + 32);
    const uint16_t *qrow = q + (size_t)b * dim;
    const int t = my[r];
    if (t >= 0) {
        const int64_t row0 = (int64_t)t * BM + wave * 32;
        const int64_t arow0 = row0 + (lane & 15), arow1 = arow0 + 16;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // the k order of sim_gemm_kernel: BK = 64 steps, two 32-wide MFMAs each, 8-element zero fill
        for (int k0 = 0; k0 < dim; k0 += BK) {
            uint4 ua0[BK / 32], ua1[BK / 32], ub[BK / 32];
#pragma unroll
            for (int s = 0; s < BK / 32; ++s) {
                const int kcol = k0 + s * 32 + 8 * (lane >> 4);
                ua0[s] = ua1[s] = ub[s] = make_uint4(0, 0, 0, 0);
                if (kcol < dim) {
                    ub[s] = *reinterpret_cast<const uint4 *>(qrow + kcol);
                    if (arow0 < rows) ua0[s] = *reinterpret_cast<const uint4 *>(emb + (size_t)arow0 * dim + kcol);
                    if (arow1 < rows) ua1[s] = *reinterpret_cast<const uint4 *>(emb + (size_t)arow1 * dim + kcol);
                }
            }
#pragma unroll
            for (int s = 0; s < BK / 32; ++s) {
                acc0 = mfma32<F16>(ua0[s], ub[s], acc0);
                acc1 = mfma32<F16>(ua1[s], ub[s], acc1);
            }
        }
        if ((lane & 15) == 0) {   // every column holds the same query: take column 0
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int at = wave * 32 + 4 * (lane >> 4) + reg;
                const int64_t m0 = row0 + 4 * (lane >> 4) + reg, m1 = m0 + 16;
                __hip_atomic_store(gc + r * BM + at, m0 < rows ? rank_key(acc0[reg], (uint32_t)m0) : 0ull,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gc + r * BM + at + 16, m1 < rows ? rank_key(acc1[reg], (uint32_t)m1) : 0ull,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wavefront: its keys have left the CU
    __syncthreads();
    if (tid == 0)
        s_last = __hip_atomic_fetch_add(my + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k - 1;
    __syncthreads();
    if (!s_last) return;
    select_from_candidates(my, b, k, cand, red, mn_in, mx_in, idx_offset, normalize, idx_out, val_out, tid);
}

// Pass 3 for wide batches: one workgroup per (tile, chunk of <= 16 of the queries that selected it).  Same fragments, same
// k order, same MFMA per score as tile_rescore_kernel -- a column of the 16 x 16 result never sees its neighbours -- so
// every key is bit-identical; only the column <-> query assignment differs (16 queries instead of one repeated).
// bucket_end / chunk_start / order: pair_scan_kernel, pair_scatter_kernel.  The grid is an upper bound of the chunks.
template <bool F16>
__global__ __launch_bounds__(256) void tile_rescore_grouped_kernel(const uint16_t *__restrict__ emb, int64_t rows, int32_t dim,
                                                                   const uint16_t *__restrict__ q, int32_t *__restrict__ rec,
                                                                   const float *__restrict__ mn_in,
                                                                   const float *__restrict__ mx_in, int32_t k,
                                                                   int32_t idx_offset, int32_t normalize,
                                                                   int32_t *__restrict__ idx_out, float *__restrict__ val_out,
                                                                   const int32_t *__restrict__ bucket_end,
                                                                   const int32_t *__restrict__ chunk_start,
                                                                   const int32_t *__restrict__ order, int32_t n_tiles) {
    __shared__ uint64_t cand[kFusedMaxK * BM];
    __shared__ uint64_t red[4];
    __shared__ int s_done[kPairChunk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x;
    if (w >= chunk_start[n_tiles]) return;
    int lo = 0, hi = n_tiles;                  // chunk_start[lo] <= w < chunk_start[hi]: the LAST bucket that starts at or
    while (hi - lo > 1) {                      // before w is the non-empty one
        const int mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= w) lo = mid;
        else hi = mid;
    }
    const int t = lo;
    const int first = (t == 0 ? 0 : bucket_end[t - 1]) + (w - chunk_start[t]) * kPairChunk;
    const int n = min(kPairChunk, bucket_end[t] - first);
    {
        const int col = lane & 15;
        const int p = order[first + min(col, n - 1)];        // columns beyond the chunk repeat its last query (not stored)
        const int bq = p / k, rq = p % k;
        const uint16_t *qrow = q + (size_t)bq * dim;
        const int64_t row0 = (int64_t)t * BM + wave * 32;
        const int64_t arow0 = row0 + (lane & 15), arow1 = arow0 + 16;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < dim; k0 += BK) {
            uint4 ua0[BK / 32], ua1[BK / 32], ub[BK / 32];
#pragma unroll
            for (int s = 0; s < BK / 32; ++s) {
                const int kcol = k0 + s * 32 + 8 * (lane >> 4);
                ua0[s] = ua1[s] = ub[s] = make_uint4(0, 0, 0, 0);
                if (kcol < dim) {
                    ub[s] = *reinterpret_cast<const uint4 *>(qrow + kcol);
                    if (arow0 < rows) ua0[s] = *reinterpret_cast<const uint4 *>(emb + (size_t)arow0 * dim + kcol);
                    if (arow1 < rows) ua1[s] = *reinterpret_cast<const uint4 *>(emb + (size_t)arow1 * dim + kcol);
                }
            }
#pragma unroll
            for (int s = 0; s < BK / 32; ++s) {
                acc0 = mfma32<F16>(ua0[s], ub[s], acc0);
                acc1 = mfma32<F16>(ua1[s], ub[s], acc1);
            }
        }
        if (col < n) {
            unsigned long long *gc = reinterpret_cast<unsigned long long *>(rec + (size_t)bq * kSelRec + 32) + rq * BM;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int at = wave * 32 + 4 * (lane >> 4) + reg;
                const int64_t m0 = row0 + 4 * (lane >> 4) + reg, m1 = m0 + 16;
                __hip_atomic_store(gc + at, m0 < rows ? rank_key(acc0[reg], (uint32_t)m0) : 0ull, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gc + at + 16, m1 < rows ? rank_key(acc1[reg], (uint32_t)m1) : 0ull, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wavefront: its keys have left the CU
    __syncthreads();
    if (tid < kPairChunk) {
        int done = -1;
        if (tid < n) {
            const int b = order[first + tid] / k;
            const int32_t *my = rec + (size_t)b * kSelRec;
            int n_valid = 0;                                 // the query's counter waits for its valid tiles (a prefix)
            for (int i = 0; i < k; ++i) n_valid += my[i] >= 0 ? 1 : 0;
            if (__hip_atomic_fetch_add(rec + (size_t)b * kSelRec + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_valid - 1)
                done = b;
        }
        s_done[tid] = done;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {            // the queries this chunk completed (each query completes exactly once)
        const int b = s_done[j];
        if (b < 0) continue;
        select_from_candidates(rec + (size_t)b * kSelRec, b, k, cand, red, mn_in, mx_in, idx_offset, normalize, idx_out,
                               val_out, tid);
        __syncthreads();
    }
}

}  // namespace

int64_t sim_fused_tiles(int64_t rows) { return ceil_div(rows, BM); }
int64_t sim_fused_sel_ints(int32_t batch) { static_assert(kSelRec == kSelRecInts, "record size"); return (int64_t)batch * kSelRec; }

// A/B switch for measurements and for the bit-identity test of the two GEMM kernels: HRAG_SIM_SMALL_TILES=1 in the
// environment keeps every shape on sim_gemm_kernel (read once)
bool sim_gemm_force_small_tiles() {
    static const bool v = [] { const char *e = experiment_env("HRAG_SIM_SMALL_TILES"); return e && e[0] == '1'; }();
    return v;
}

// A/B switch for measurements: HRAG_RESCORE_GROUPED=0 keeps pass 3 one workgroup per (tile, query) at every batch (read once)
bool rescore_grouping_disabled() {
    static const bool v = [] { const char *e = experiment_env("HRAG_RESCORE_GROUPED"); return e && e[0] == '0'; }();
    return v;
}

// ws: 2 * tiles * batch floats (tile max / min) ; sel: sim_fused_sel_ints(batch) ints, ZEROED at allocation (per-query
// records: selected tiles, arrival counter, candidate keys) ; mn / mx: batch floats
hrag_status launch_sim_topk_fused(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q,
                                  int32_t batch, int32_t k, int32_t idx_offset, int32_t normalize,
                                  float *ws, int32_t *sel, float *mn, float *mx, int32_t *idx_out,
                                  float *val_out, hipStream_t s, int32_t dtype, int32_t approx_dim, float cut,
                                  int32_t *overflow) {
    HRAG_REQUIRE(dim > 0 && dim % 8 == 0, "embedding dim %d must be a positive multiple of 8", dim);
    // Thresholded form (the index-time KNN, round 6): the caller only reads results at or above a score.  cut: tiles whose
    // maximum is below it are never rescored; approx_dim > 0: pass 1 runs over the first approx_dim elements of every row
    // only -- the caller guarantees |full product - prefix product| <= (its threshold - cut) -- and pass 3 rescores the tiles
    // that reach the cut over all `dim` elements, so every returned score is the exact chain and every row at or above the
    // caller's threshold is among the candidates unless overflow[b] says that more than k tiles reached the cut.
    HRAG_REQUIRE(approx_dim == 0 || (approx_dim > 0 && approx_dim <= dim && normalize == 0),
                 "fused top-k: approx_dim %d must lie in (0, dim] and needs normalize = 0", approx_dim);
    if (approx_dim > 0 && !(sim_gemm256_serves(rows, approx_dim, batch) && !sim_gemm_force_small_tiles())) approx_dim = 0;
    HRAG_REQUIRE(k >= 1 && k <= kFusedMaxK && rows >= 1 && batch >= 1, "fused top-k: bad k / rows / batch");
    const int64_t tiles_m = ceil_div(rows, BM);
    float *tmax = ws, *tmin = ws + (size_t)tiles_m * batch;
    const bool f16 = dtype == HRAG_FP16;
    // the same tile shapes as launch_sim_gemm (the scores do not depend on the shape: every element is
    // the same chain of 16x16x32 MFMAs in ascending k)
#define LAUNCH_TM(BN_, WM_, WN_)                                                                              \
    do {                                                                                                      \
        const int tn = (int)ceil_div(batch, BN_);                                                             \
        const dim3 grid((unsigned)(round_up(tiles_m, 8) * tn));                                               \
        if (f16)                                                                                              \
            hipLaunchKernelGGL((sim_gemm_kernel<BN_, WM_, WN_, true, true>), grid, dim3(256), 0, s, emb, rows, \
                               dim, q, batch, nullptr, 0, tn, 0, tmax, tmin);                                 \
        else                                                                                                  \
            hipLaunchKernelGGL((sim_gemm_kernel<BN_, WM_, WN_, true, false>), grid, dim3(256), 0, s, emb, rows, \
                               dim, q, batch, nullptr, 0, tn, 0, tmax, tmin);                                 \
    } while (0)
    if (approx_dim > 0)
        HRAG_TRY(launch_sim_gemm256(emb, rows, approx_dim, q, batch, nullptr, 0, tmax, tmin, s, dtype, dim));
    else if (sim_gemm256_serves(rows, dim, batch) && !(sim_gemm_force_small_tiles()))
        HRAG_TRY(launch_sim_gemm256(emb, rows, dim, q, batch, nullptr, 0, tmax, tmin, s, dtype));
    else if (batch > 64) LAUNCH_TM(128, 2, 2);
    else LAUNCH_TM(64, 4, 1);
#undef LAUNCH_TM
    HRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(tile_select_kernel, dim3((unsigned)batch), dim3(256), 0, s, tmax, tmin, (int32_t)tiles_m,
                       batch, k, sel, mn, mx, cut, overflow);
    HRAG_LAUNCH_CHECK();
    // wide batches: pass 3 grouped by tile (see pair_hist_kernel); the tile minima are dead after tile_select_kernel and
    // lend their memory: [n_tiles + 1] bucket cursors, [n_tiles + 2] chunk starts, then the ordered pair ids
    const int64_t n_pairs = (int64_t)batch * k;
    const bool grouped = tiles_m >= 64 && n_pairs >= 4 * tiles_m && n_pairs < (1ll << 30) &&
                         n_pairs + 2 * tiles_m + 3 <= tiles_m * (int64_t)batch && !rescore_grouping_disabled();
    if (grouped) {
        int32_t *cursor = reinterpret_cast<int32_t *>(tmin), *chunk = cursor + tiles_m + 1, *ord = chunk + tiles_m + 2;
        const unsigned pb = (unsigned)ceil_div(n_pairs, 256);
        const dim3 grid3((unsigned)(ceil_div(n_pairs, kPairChunk) + tiles_m));         // >= the number of chunks
        HRAG_HIP_TRY(hipMemsetAsync(cursor, 0, (size_t)(tiles_m + 1) * sizeof(int32_t), s));
        hipLaunchKernelGGL(pair_hist_kernel, dim3(pb), dim3(256), 0, s, sel, (int32_t)n_pairs, k, cursor, idx_out, val_out);
        HRAG_LAUNCH_CHECK();
        hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(1024), 0, s, cursor, chunk, (int32_t)tiles_m);
        HRAG_LAUNCH_CHECK();
        hipLaunchKernelGGL(pair_scatter_kernel, dim3(pb), dim3(256), 0, s, sel, (int32_t)n_pairs, k, cursor, ord);
        HRAG_LAUNCH_CHECK();
        if (dtype == HRAG_FP16)
            hipLaunchKernelGGL(tile_rescore_grouped_kernel<true>, grid3, dim3(256), 0, s, emb, rows, dim, q, sel, mn, mx, k,
                               idx_offset, normalize, idx_out, val_out, cursor, chunk, ord, (int32_t)tiles_m);
        else
            hipLaunchKernelGGL(tile_rescore_grouped_kernel<false>, grid3, dim3(256), 0, s, emb, rows, dim, q, sel, mn, mx, k,
                               idx_offset, normalize, idx_out, val_out, cursor, chunk, ord, (int32_t)tiles_m);
    } else if (dtype == HRAG_FP16)
        hipLaunchKernelGGL(tile_rescore_kernel<true>, dim3((unsigned)k, (unsigned)batch), dim3(256), 0, s, emb, rows, dim,
                           q, sel, mn, mx, k, idx_offset, normalize, idx_out, val_out);
    else
        hipLaunchKernelGGL(tile_rescore_kernel<false>, dim3((unsigned)k, (unsigned)batch), dim3(256), 0, s, emb, rows, dim,
                           q, sel, mn, mx, k, idx_offset, normalize, idx_out, val_out);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

namespace {
}  // namespace

hrag_status launch_sim_gemm(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q,
                            int32_t batch, float *out, int64_t ld, hipStream_t s, int32_t accumulate,
                            int32_t dtype) {
    HRAG_REQUIRE(dim > 0 && dim % 8 == 0, "embedding dim %d must be a positive multiple of 8", dim);
    if (rows == 0 || batch == 0) return HRAG_OK;
    // latency path: GEMV.  Measured at F = 875k, D = 768: B = 1 0.27 ms (MFMA kernel 0.70), B = 2 0.31;
    // from B = 4 the per-lane dot products make it VALU-bound (0.96 ms) and the MFMA kernel wins.
    if (batch <= 2 && !accumulate && launch_sim_gemv(emb, rows, dim, q, batch, out, ld, s, dtype)) {
        HRAG_LAUNCH_CHECK();
        return HRAG_OK;
    }
    if (!accumulate && sim_gemm256_serves(rows, dim, batch) && !sim_gemm_force_small_tiles())
        return launch_sim_gemm256(emb, rows, dim, q, batch, out, ld, nullptr, nullptr, s, dtype);
    const int64_t tiles_m = ceil_div(rows, BM);
    const bool f16 = dtype == HRAG_FP16;
#define LAUNCH(BN_, WM_, WN_, GRID, TN)                                                                       \
    do {                                                                                                      \
        if (f16)                                                                                              \
            hipLaunchKernelGGL((sim_gemm_kernel<BN_, WM_, WN_, false, true>), dim3((unsigned)(GRID)), dim3(256), \
                               0, s, emb, rows, dim, q, batch, out, ld, TN, accumulate);                     \
        else                                                                                                  \
            hipLaunchKernelGGL((sim_gemm_kernel<BN_, WM_, WN_, false, false>), dim3((unsigned)(GRID)), dim3(256), \
                               0, s, emb, rows, dim, q, batch, out, ld, TN, accumulate);                     \
    } while (0)
    if (batch > 64) {
        const int tn = (int)ceil_div(batch, 128);
        LAUNCH(128, 2, 2, round_up(tiles_m, 8) * tn, tn);
    } else if (batch > 16) {
        const int tn = (int)ceil_div(batch, 64);
        LAUNCH(64, 4, 1, round_up(tiles_m, 8) * tn, tn);
    } else {
        LAUNCH(16, 4, 1, round_up(tiles_m, 8), 1);
    }
#undef LAUNCH
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
