// K2 -- entity ("phrase") seeds of the reset vector: 32 lanes per query, one per (kept fact, side).
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
// At most 2 * kf <= 32 (fact, side) entries per query: lane i of a query's half-wavefront holds entry i (fact i / 2,
// subject before object); the reference's dict is a pass of 32 shuffles (sum and count of the entries with the same
// vertex, in entry order = the order the reference adds them up), the stable selection another one (rank = entries
// of other phrases that are heavier, or equally heavy and earlier).  Round 2 ran one THREAD per query with
// dynamically indexed private arrays (scratch memory): 32 us at any batch size; this takes ~6.
#include "common.h"

namespace hrag {
namespace {

static_assert(kMaxSeeds == 32 && 2 * kMaxKeptFacts == kMaxSeeds, "one lane per (fact, side) entry");

__global__ __launch_bounds__(64) void build_seeds_kernel(const int32_t *__restrict__ kept_idx,
                                                         const float *__restrict__ kept_score,
                                                         const int32_t *__restrict__ kept_count, int32_t kf,
                                                         int32_t link_top_k, int32_t batch,
                                                         const int32_t *__restrict__ subj,
                                                         const int32_t *__restrict__ obj, int64_t n_facts,
                                                         const int32_t *__restrict__ num_chunks, int64_t num_vertices,
                                                         int32_t *__restrict__ seed_vtx, float *__restrict__ seed_w,
                                                         int32_t *__restrict__ seed_cnt, int32_t *__restrict__ flags) {
    const int lane = threadIdx.x & 63, half = lane >> 5, i = lane & 31, base = half * 32;
    const int q = blockIdx.x * 2 + half;
    const bool qv = q < batch;
    int cnt = qv ? kept_count[q] : 0;
    cnt = cnt < 0 ? 0 : (cnt > kf ? kf : cnt);
    const int r = i >> 1;                               // fact of this entry, in filter order
    const bool on = qv && r < cnt;
    const int32_t fidx = on ? kept_idx[q * kf + r] : -1;
    const float score = on ? kept_score[q * kf + r] : 0.f;
    int32_t v = -1;
    if (fidx >= 0 && (int64_t)fidx < n_facts) v = (i & 1) ? obj[fidx] : subj[fidx];
    int32_t nch = 0;
    if (v >= 0 && (int64_t)v < num_vertices) nch = num_chunks[v]; else v = -1;
    const float w = nch > 0 ? __fdiv_rn(score, (float)nch) : score;
    // the "dict": sum / count of the entries with my vertex, in entry order; `first` = its first entry
    double wsum = 0.0;
    int occ = 0, first = i;
    for (int j = 0; j < 32; ++j) {
        const int32_t vj = __shfl(v, base + j, 64);
        const float wj = __shfl(w, base + j, 64);
        if (v >= 0 && vj == v) {
            wsum += (double)wj;
            occ += 1;
            first = j < first ? j : first;
        }
    }
    const bool rep = v >= 0 && first == i;               // this lane speaks for its phrase
    const double mean = rep ? wsum / (double)occ : 0.0;
    // stable selection: heavier first, ties in first-occurrence order
    int rank = 0;
    for (int j = 0; j < 32; ++j) {
        const double mj = __shfl(mean, base + j, 64);
        const int rj = __shfl((int)rep, base + j, 64);
        if (rj && j != i && (mj > mean || (mj == mean && j < i))) rank += 1;
    }
    const unsigned long long reps = __builtin_amdgcn_ballot_w64(rep);
    const int m = __builtin_popcountll((reps >> base) & 0xffffffffull);
    const int keep = (link_top_k > 0 && link_top_k < m) ? link_top_k : m;
    const bool kept = rep && rank < keep;
    if (kept) {
        seed_vtx[q * kMaxSeeds + rank] = v;
        seed_w[q * kMaxSeeds + rank] = (float)mean;
    }
    const unsigned long long zero_kept = __builtin_amdgcn_ballot_w64(kept && mean == 0.0);
    if (qv && i == 0) {
        int32_t flag = flags[q] & ~(1 | 4);
        if (cnt == 0) flag |= 1;                          // DPR fallback (HippoRAG.py:467-469)
        if ((zero_kept >> base) & 0xffffffffull) flag |= 4;
        seed_cnt[q] = keep;
        flags[q] = flag;
    }
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
    hipLaunchKernelGGL(build_seeds_kernel, dim3((unsigned)ceil_div(batch, 2)), dim3(64), 0, s,
                       kept_idx, kept_score, kept_count, kf, link_top_k, batch, subj, obj, n_facts,
                       num_chunks, num_vertices, seed_vtx, seed_w, seed_cnt, flags);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace hrag
