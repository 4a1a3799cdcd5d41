// K2 -- entity ("phrase") seeds of the reset vector, one thread per query.
//
// Restates the array-level effect of graph_search_with_fact_entities
// (reference src/hipporag/HippoRAG.py:1574-1623) + get_top_k_weights (:1505-1542):
//   for every kept fact (filter order) and for subject, then object:
//       vertex absent (-1)                                  -> skip              (:1595-1597)
//       w = fact_score / num_chunks[vertex] if num_chunks>0 -> fp32 division     (:1598-1601)
//       phrase_weights[vertex] += w ; number_of_occurs[vertex] += 1  (fp64)      (:1603-1604)
//   phrase_weights /= number_of_occurs                                           (:1608)
//   keep the link_top_k heaviest phrases (stable: ties keep first-occurrence order; the reference
//   order is that of a Python set, i.e. undefined)                               (:1528)
//   flag bit2 when a kept phrase has weight exactly 0 -- the reference's
//   assert np.count_nonzero(...) == len(linking_score_map) would fire            (:1541)
// At most 2 * kf <= 32 distinct vertices per query, so the "dict" is a linear probe over a
// per-thread array; the work is negligible next to the PPR sweep that consumes the result.
#include "common.h"

namespace hrag {
namespace {

__global__ void build_seeds_kernel(const int32_t *__restrict__ kept_idx,
                                   const float *__restrict__ kept_score,
                                   const int32_t *__restrict__ kept_count, int32_t kf,
                                   int32_t link_top_k, int32_t batch,
                                   const int32_t *__restrict__ subj,
                                   const int32_t *__restrict__ obj, int64_t n_facts,
                                   const int32_t *__restrict__ num_chunks, int64_t num_vertices,
                                   int32_t *__restrict__ seed_vtx, float *__restrict__ seed_w,
                                   int32_t *__restrict__ seed_cnt, int32_t *__restrict__ flags) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= batch) return;
    int32_t ids[kMaxSeeds];
    double wsum[kMaxSeeds];
    int32_t occ[kMaxSeeds];
    int m = 0;
    int cnt = kept_count[q];
    cnt = cnt < 0 ? 0 : (cnt > kf ? kf : cnt);
    int32_t flag = flags[q] & ~(1 | 4);
    if (cnt == 0) flag |= 1;  // DPR fallback (HippoRAG.py:467-469)
    // three rounds of independent loads (facts -> their two entities -> the entities' chunk counts) instead of a
    // chain of 3 dependent loads per fact and side: one thread per query is latency, not work (42 -> 12 us at cfg 3)
    int32_t fidx[kMaxKeptFacts], vtx[2 * kMaxKeptFacts], nch[2 * kMaxKeptFacts];
    float fsc[kMaxKeptFacts];
#pragma unroll
    for (int r = 0; r < kMaxKeptFacts; ++r) {
        const bool on = r < cnt;
        fidx[r] = on ? kept_idx[q * kf + r] : -1;
        fsc[r] = on ? kept_score[q * kf + r] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < kMaxKeptFacts; ++r) {
        const bool on = fidx[r] >= 0 && (int64_t)fidx[r] < n_facts;
        vtx[2 * r] = on ? subj[fidx[r]] : -1;
        vtx[2 * r + 1] = on ? obj[fidx[r]] : -1;
    }
#pragma unroll
    for (int i = 0; i < 2 * kMaxKeptFacts; ++i) {
        const bool on = vtx[i] >= 0 && (int64_t)vtx[i] < num_vertices;
        nch[i] = on ? num_chunks[vtx[i]] : 0;
        if (!on) vtx[i] = -1;
    }
#pragma unroll
    for (int i = 0; i < 2 * kMaxKeptFacts; ++i) {   // fact r = i / 2 in rank order, subject before object
        const int32_t v = vtx[i];
        if (v < 0) continue;
        const float score = fsc[i >> 1];
        const float w = nch[i] > 0 ? __fdiv_rn(score, (float)nch[i]) : score;
        int j = 0;
        while (j < m && ids[j] != v) ++j;
        if (j == m) {
            ids[m] = v;
            wsum[m] = 0.0;
            occ[m] = 0;
            ++m;
        }
        wsum[j] += (double)w;
        occ[j] += 1;
    }
    for (int j = 0; j < m; ++j) wsum[j] /= (double)occ[j];
    // stable selection of the link_top_k heaviest (link_top_k <= 0: keep all, sorted)
    const int keep = (link_top_k > 0 && link_top_k < m) ? link_top_k : m;
    bool used[kMaxSeeds];
    for (int j = 0; j < m; ++j) used[j] = false;
    int n_out = 0;
    for (int t = 0; t < keep; ++t) {
        int best = -1;
        for (int j = 0; j < m; ++j)
            if (!used[j] && (best < 0 || wsum[j] > wsum[best])) best = j;
        used[best] = true;
        if (wsum[best] == 0.0) flag |= 4;
        seed_vtx[q * kMaxSeeds + n_out] = ids[best];
        seed_w[q * kMaxSeeds + n_out] = (float)wsum[best];
        ++n_out;
    }
    seed_cnt[q] = n_out;
    flags[q] = flag;
}

}  // namespace

hrag_status launch_build_seeds(const int32_t *kept_idx, const float *kept_score,
                               const int32_t *kept_count, int32_t kf, int32_t link_top_k,
                               int32_t batch, const int32_t *subj, const int32_t *obj,
                               int64_t n_facts, const int32_t *num_chunks, int64_t num_vertices,
                               int32_t *seed_vtx, float *seed_w, int32_t *seed_cnt,
                               int32_t *flags, hipStream_t s) {
    HRAG_REQUIRE(kf >= 1 && kf <= kMaxKeptFacts, "kf=%d outside [1, %d]", kf, kMaxKeptFacts);
    if (batch == 0) return HRAG_OK;
    hipLaunchKernelGGL(build_seeds_kernel, dim3((unsigned)ceil_div(batch, 64)), dim3(64), 0, s,
                       kept_idx, kept_score, kept_count, kf, link_top_k, batch, subj, obj, n_facts,
                       num_chunks, num_vertices, seed_vtx, seed_w, seed_cnt, flags);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
