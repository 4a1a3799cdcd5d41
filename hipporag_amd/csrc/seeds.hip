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
    for (int r = 0; r < cnt; ++r) {
        const int64_t f = kept_idx[q * kf + r];
        if (f < 0 || f >= n_facts) continue;
        const float score = kept_score[q * kf + r];
        for (int side = 0; side < 2; ++side) {
            const int32_t v = side == 0 ? subj[f] : obj[f];
            if (v < 0 || v >= num_vertices) continue;
            float w = score;
            const int32_t nc = num_chunks[v];
            if (nc > 0) w = __fdiv_rn(score, (float)nc);
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
