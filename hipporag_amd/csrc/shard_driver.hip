// One-call drivers of the ROW-SHARDED hot path (round 6): hrag_shard_score_facts_all / hrag_shard_retrieve run the whole
// phase A / phase B of a row shard -- local similarity, candidate merge, prior statistics, the staged e4m3 PPR with its
// per-sweep state exchange pipelined over the exchange groups, the convergence contract's all-reduced measure, the merged
// top-k -- and call the host back ONLY for the collectives (hrag_comm: four function pointers).  What a multi-GPU host
// had to write around the hrag_shard_* steps before (hipporag_amd/dist.py ShardedRetriever, ~100 lines of orchestration
// per phase, kept as the reference implementation and compared bit for bit in tests/test_gpu_multi.py) is now behind the
// C ABI: a maintainer of the reference supplies ncclAllReduce / ncclAllGather one-liners and calls two functions.
//
// The reference has no analogue (src/hipporag/HippoRAG.py:459 is a serial loop over the queries); the layout is
// BASELINE.json's: CSR rows + embeddings sharded, the PPR iterate exchanged every sweep.
#include "engine_impl.h"

namespace hrag {
namespace {

// gathered per-shard top-k lists [world][B][k] (each: score desc, id desc; shard id ranges ascending in rank order) ->
// candidate rows [B][world * k], every list REVERSED and the lists in rank order, so that "later position" == "larger
// (score, id)" among equal scores and the library's positional tie rule reproduces the global order (dist.merge_ranked)
__global__ __launch_bounds__(256) void merge_prepare_kernel(const int32_t *__restrict__ gidx, const float *__restrict__ gval,
                                                            int32_t world, int32_t batch, int32_t k,
                                                            int32_t *__restrict__ cand_idx, float *__restrict__ cand_val) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)world * batch * k;
    if (t >= n) return;
    const int j = (int)(t % k);
    const int b = (int)((t / k) % batch);
    const int r = (int)(t / ((int64_t)k * batch));
    const int32_t id = gidx[t];
    const size_t at = (size_t)b * world * k + (size_t)r * k + (size_t)(k - 1 - j);
    cand_idx[at] = id;
    cand_val[at] = id < 0 ? -INFINITY : gval[t];
}

// positions of the merged top-k -> ids, values (-1 / 0 beyond the candidates); mn / mx non-null: min-max normalise the
// values with the GLOBAL row minimum / maximum (phase A: misc_utils.py:130-139, a zero range gives ones)
__global__ __launch_bounds__(256) void merge_finish_kernel(const int32_t *__restrict__ pos, const float *__restrict__ top_val,
                                                           const int32_t *__restrict__ cand_idx, int32_t batch, int32_t k,
                                                           int32_t n_cand, const float *__restrict__ mn,
                                                           const float *__restrict__ mx, int32_t *__restrict__ out_idx,
                                                           float *__restrict__ out_val) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)batch * k) return;
    const int b = (int)(t / k);
    const int32_t p = pos[t];
    const int32_t id = p < 0 ? -1 : cand_idx[(size_t)b * n_cand + p];
    float v = top_val[t];
    if (mn) {
        const float lo = mn[b], rng = mx[b] - lo;
        v = rng == 0.f ? 1.f : (v - lo) / rng;
    }
    out_idx[t] = id;
    out_val[t] = id < 0 ? 0.f : v;
}

__global__ __launch_bounds__(256) void sat_extract_kernel(const int32_t *__restrict__ flags, int32_t batch, int32_t *__restrict__ sat) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) sat[t] = flags[t] & kFlagFp8Saturated;
}
__global__ __launch_bounds__(256) void sat_merge_kernel(const int32_t *__restrict__ flags, const int32_t *__restrict__ sat,
                                                        int32_t batch, int32_t *__restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) out[t] = flags[t] | sat[t];
}

// the caller's workspace, carved in 256-byte steps
struct Carver {
    char *p;
    int64_t left;
    template <typename T>
    T *take(int64_t count) {
        const int64_t bytes = round_up(std::max<int64_t>(count, 1) * (int64_t)sizeof(T), 256);
        if (bytes > left) return nullptr;
        T *r = reinterpret_cast<T *>(p);
        p += bytes;
        left -= bytes;
        return r;
    }
};

#define COMM_TRY(call, what)                                                                      \
    do {                                                                                            \
        const int32_t _rc = (call);                                                                 \
        if (_rc != 0) {                                                                             \
            set_error("hrag_comm.%s returned %d", what, (int)_rc);                                  \
            return HRAG_EINVAL;                                                                     \
        }                                                                                           \
    } while (0)

hrag_status check_comm(const hrag_comm *c) {
    HRAG_REQUIRE(c != nullptr, "comm is NULL");
    HRAG_REQUIRE(c->world >= 1 && c->rank >= 0 && c->rank < c->world, "comm: rank %d outside [0, world=%d)", c->rank, c->world);
    HRAG_REQUIRE(c->world == 1 || (c->all_reduce && c->all_gather && c->exchange_begin && c->exchange_wait),
                 "comm: all four collectives must be set when world > 1");
    return HRAG_OK;
}

// merge the shards' local top-k lists (lidx / lval [B][k] on this shard) into the global ones
hrag_status merge_topk(const hrag_comm *c, const int32_t *lidx, const float *lval, int32_t batch, int32_t k, Carver &ws,
                       const float *mn, const float *mx, int32_t *out_idx, float *out_val, hipStream_t s) {
    const int32_t w = c->world;
    const int64_t per = (int64_t)batch * k;
    int32_t *gidx = ws.take<int32_t>(per * w), *cidx = ws.take<int32_t>(per * w), *pos = ws.take<int32_t>(per);
    float *gval = ws.take<float>(per * w), *cval = ws.take<float>(per * w), *tval = ws.take<float>(per);
    HRAG_REQUIRE(gidx && cidx && pos && gval && cval && tval, "workspace too small (hrag_shard_workspace_bytes)");
    if (w > 1) {
        COMM_TRY(c->all_gather(c->user, lidx, gidx, per * (int64_t)sizeof(int32_t), (hrag_stream)s), "all_gather");
        COMM_TRY(c->all_gather(c->user, lval, gval, per * (int64_t)sizeof(float), (hrag_stream)s), "all_gather");
    } else {
        HRAG_HIP_TRY(hipMemcpyAsync(gidx, lidx, (size_t)per * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        HRAG_HIP_TRY(hipMemcpyAsync(gval, lval, (size_t)per * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(merge_prepare_kernel, dim3((unsigned)ceil_div(per * w, 256)), dim3(256), 0, s, gidx, gval, w, batch, k,
                       cidx, cval);
    HRAG_LAUNCH_CHECK();
    HRAG_TRY(hrag_topk_rows(cval, batch, (int64_t)w * k, (int64_t)w * k, k, 0, 0, pos, tval, nullptr, nullptr, (hrag_stream)s));
    hipLaunchKernelGGL(merge_finish_kernel, dim3((unsigned)ceil_div(per, 256)), dim3(256), 0, s, pos, tval, cidx, batch, k,
                       w * k, mn, mx, out_idx, out_val);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // namespace
}  // namespace hrag

using namespace hrag;

extern "C" {

int64_t hrag_shard_workspace_bytes(hrag_engine *e, int32_t world, int32_t batch, int32_t k) {
    if (!e || world < 1 || batch < 1 || k < 1) return 0;
    const int64_t per = (int64_t)batch * k, b = batch;
    auto pad = [](int64_t bytes) { return round_up(std::max<int64_t>(bytes, 1), 256); };
    int64_t n = 0;
    n += 4 * pad(b * 4);                                   // mn, mx, zmax, est
    n += pad(2 * b * 8);                                   // mass
    n += 2 * pad(b * HRAG_SEED_STRIDE * 4) + 3 * pad(b * 4);   // seeds, seed count, flags, sat
    n += 2 * pad(per * 4);                                 // local top-k
    n += 4 * pad(per * world * 4) + 2 * pad(per * 4);      // gathered lists, candidates, merged positions / values
    return n;
}

hrag_status hrag_shard_score_facts_all(hrag_engine *e, const hrag_comm *comm, const uint16_t *q_fact, int32_t batch, int32_t k,
                                       void *workspace, int64_t workspace_bytes, int32_t *idx_out, float *score_out,
                                       hrag_stream stream) {
    HRAG_TRY(check_comm(comm));
    HRAG_REQUIRE(e && q_fact && workspace && idx_out && score_out && k >= 1, "bad argument");
    HRAG_REQUIRE(workspace_bytes >= hrag_shard_workspace_bytes(e, comm->world, batch, k), "workspace too small: %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)hrag_shard_workspace_bytes(e, comm->world, batch, k));
    hipStream_t s = (hipStream_t)stream;
    Carver ws{static_cast<char *>(workspace), workspace_bytes};
    float *mn = ws.take<float>(batch), *mx = ws.take<float>(batch);
    int32_t *lidx = ws.take<int32_t>((int64_t)batch * k);
    float *lval = ws.take<float>((int64_t)batch * k);
    HRAG_REQUIRE(mn && mx && lidx && lval, "workspace too small");
    HRAG_TRY(hrag_shard_score_facts(e, q_fact, batch, k, lidx, lval, mn, mx, stream));
    if (comm->world > 1) {
        COMM_TRY(comm->all_reduce(comm->user, mn, batch, HRAG_COMM_F32, HRAG_COMM_MIN, stream), "all_reduce");
        COMM_TRY(comm->all_reduce(comm->user, mx, batch, HRAG_COMM_F32, HRAG_COMM_MAX, stream), "all_reduce");
    }
    return merge_topk(comm, lidx, lval, batch, k, ws, mn, mx, idx_out, score_out, s);
}

hrag_status hrag_shard_retrieve(hrag_engine *e, const hrag_comm *comm, const uint16_t *q_pass, int32_t batch,
                                const int32_t *kept_idx, const float *kept_score, const int32_t *kept_count, int32_t kf,
                                int32_t link_top_k, float damping, float passage_node_weight, int32_t ppr_iters,
                                int32_t ppr_max_iters, float ppr_tol, int32_t k, int32_t n_groups, void *state0, void *state1,
                                void *state2, void *workspace, int64_t workspace_bytes, int32_t *doc_idx_out,
                                float *doc_score_out, int32_t *flags_out, float *residual_out, int32_t *iters_out,
                                hrag_stream stream) {
    HRAG_TRY(check_comm(comm));
    HRAG_REQUIRE(e && q_pass && kept_idx && kept_score && kept_count && state0 && state1 && state2 && workspace &&
                 doc_idx_out && doc_score_out && flags_out, "NULL argument");
    HRAG_REQUIRE(workspace_bytes >= hrag_shard_workspace_bytes(e, comm->world, batch, k), "workspace too small: %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)hrag_shard_workspace_bytes(e, comm->world, batch, k));
    const int32_t w = comm->world;
    hipStream_t s = (hipStream_t)stream;
    hrag_shard_layout lay;
    HRAG_TRY(hrag_shard_layout_query(e, batch, n_groups, &lay));
    if (w > 1) {
        HRAG_REQUIRE(lay.own_offset == (int64_t)comm->rank * lay.own_bytes,
                     "the state exchange needs equal-sized row shards in rank order (own_offset %lld != rank %d * own_bytes %lld)",
                     (long long)lay.own_offset, comm->rank, (long long)lay.own_bytes);
        HRAG_REQUIRE((int64_t)w * lay.own_bytes + (int64_t)lay.slabs_per_group * 128 <= lay.group_bytes,
                     "the state exchange would overwrite the zero row: the shards do not tile [0, V)");
    }
    Carver ws{static_cast<char *>(workspace), workspace_bytes};
    float *mn = ws.take<float>(batch), *mx = ws.take<float>(batch), *zmax = ws.take<float>(batch), *est = ws.take<float>(batch);
    double *mass = ws.take<double>(2 * (int64_t)batch);
    int32_t *sv = ws.take<int32_t>((int64_t)batch * HRAG_SEED_STRIDE);
    float *sw = ws.take<float>((int64_t)batch * HRAG_SEED_STRIDE);
    int32_t *sc = ws.take<int32_t>(batch), *flags = ws.take<int32_t>(batch), *sat = ws.take<int32_t>(batch);
    int32_t *lidx = ws.take<int32_t>((int64_t)batch * k);
    float *lval = ws.take<float>((int64_t)batch * k);
    HRAG_REQUIRE(mn && mx && zmax && est && mass && sv && sw && sc && flags && sat && lidx && lval, "workspace too small");
    auto reduce = [&](void *buf, int64_t n, int32_t dt, int32_t op) -> hrag_status {
        if (w > 1) COMM_TRY(comm->all_reduce(comm->user, buf, n, dt, op, stream), "all_reduce");
        return HRAG_OK;
    };
    // the seed arrays and the flag words start from zero (the seed and prior kernels OR into the flags and write only the
    // seeds a query has): the workspace is the caller's and holds whatever the last call left
    HRAG_HIP_TRY(hipMemsetAsync(sv, 0, (size_t)batch * HRAG_SEED_STRIDE * sizeof(int32_t), s));
    HRAG_HIP_TRY(hipMemsetAsync(sw, 0, (size_t)batch * HRAG_SEED_STRIDE * sizeof(float), s));
    HRAG_HIP_TRY(hipMemsetAsync(sc, 0, (size_t)batch * sizeof(int32_t), s));
    HRAG_HIP_TRY(hipMemsetAsync(flags, 0, (size_t)batch * sizeof(int32_t), s));
    // ---- similarity of the owned passages, the reset vector's statistics over ALL shards
    HRAG_TRY(hrag_shard_passage_scores(e, q_pass, batch, mn, mx, stream));
    HRAG_TRY(reduce(mn, batch, HRAG_COMM_F32, HRAG_COMM_MIN));
    HRAG_TRY(reduce(mx, batch, HRAG_COMM_F32, HRAG_COMM_MAX));
    HRAG_TRY(hrag_stage_seeds(e, kept_idx, kept_score, kept_count, kf, link_top_k, batch, sv, sw, sc, flags, stream));
    HRAG_TRY(hrag_shard_prior_stats(e, mn, mx, passage_node_weight, flags, batch, zmax, mass, stream));
    HRAG_TRY(reduce(zmax, batch, HRAG_COMM_F32, HRAG_COMM_MAX));
    HRAG_TRY(reduce(mass, 2 * (int64_t)batch, HRAG_COMM_F64, HRAG_COMM_SUM));
    const bool contract = ppr_tol > 0.f;
    int32_t n_steps = 0;
    HRAG_TRY(hrag_shard_ppr_begin(e, mn, mx, zmax, mass, passage_node_weight, sv, sw, sc, flags, batch, damping, ppr_iters,
                                  std::max(ppr_max_iters, ppr_iters), contract ? ppr_tol : 0.f, lay.n_groups, state0, state1,
                                  state2, &n_steps, stream));
    if (!contract) n_steps = ppr_iters;
    // ---- the sweeps: group g's exchange overlaps with the sweeps of the other groups; a sweep of group g only waits for
    //      group g's previous exchange
    void *bufs[3] = {state0, state1, state2};
    std::vector<char> pending((size_t)lay.n_groups, 0);
    // an error return between exchange_begin and exchange_wait must not leave the host's collective handles open (the
    // next call on the same hrag_comm would start from a half-finished exchange): wait for what was begun, ignore the rc
    struct Drain {
        const hrag_comm *c; std::vector<char> &p; hrag_stream st;
        ~Drain() {
            for (size_t g = 0; g < p.size(); ++g)
                if (p[g]) { (void)c->exchange_wait(c->user, (int32_t)g, st); p[g] = 0; }
        }
    } drain{comm, pending, stream};
    auto begin_x = [&](int buf, int g) -> hrag_status {
        if (w > 1) {
            char *region = static_cast<char *>(bufs[buf]) + (size_t)g * (size_t)lay.group_bytes;
            COMM_TRY(comm->exchange_begin(comm->user, region, lay.own_bytes, g, stream), "exchange_begin");
            pending[(size_t)g] = 1;
        }
        return HRAG_OK;
    };
    auto wait_x = [&](int g) -> hrag_status {
        if (pending[(size_t)g]) {
            COMM_TRY(comm->exchange_wait(comm->user, g, stream), "exchange_wait");
            pending[(size_t)g] = 0;
        }
        return HRAG_OK;
    };
    for (int g = 0; g < lay.n_groups; ++g) HRAG_TRY(begin_x(0, g));
    bool est_global = false;
    for (int i = 0; i < n_steps; ++i) {
        int32_t ck = 0;
        for (int g = 0; g < lay.n_groups; ++g) {
            HRAG_TRY(wait_x(g));
            int32_t xb = -1, ck_g = 0;
            HRAG_TRY(hrag_shard_ppr_sweep(e, i, g, &xb, &ck_g, stream));
            ck |= ck_g;
            if (xb >= 0) HRAG_TRY(begin_x(xb, g));
        }
        est_global = false;
        if (contract && ck) {   // a final sweep that measured: the residual over ALL passages, then the decision
            HRAG_TRY(hrag_shard_ppr_est(e, est, 0, stream));
            HRAG_TRY(reduce(est, batch, HRAG_COMM_F32, HRAG_COMM_MAX));
            HRAG_TRY(hrag_shard_ppr_est(e, est, 1, stream));
            HRAG_TRY(hrag_shard_ppr_decide(e, i, stream));
            est_global = true;
            int32_t open = 0;   // the same on every shard (the measure was all-reduced): once closed, every later step is
            HRAG_TRY(hrag_shard_ppr_gate(e, i + 1, &open, stream));
            if (!open) break;
        }
    }
    for (int g = 0; g < lay.n_groups; ++g) HRAG_TRY(wait_x(g));
    if (contract && !est_global) {
        HRAG_TRY(hrag_shard_ppr_est(e, est, 0, stream));
        HRAG_TRY(reduce(est, batch, HRAG_COMM_F32, HRAG_COMM_MAX));
        HRAG_TRY(hrag_shard_ppr_est(e, est, 1, stream));
    }
    HRAG_TRY(hrag_shard_finish(e, mn, mx, flags, batch, k, lidx, lval, contract ? residual_out : nullptr,
                               contract ? iters_out : nullptr, stream));
    HRAG_TRY(merge_topk(comm, lidx, lval, batch, k, ws, nullptr, nullptr, doc_idx_out, doc_score_out, s));
    // a saturated value is raised on the shard that owns the row: every shard reports it (never clipped scores unnoticed)
    hipLaunchKernelGGL(sat_extract_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, s, flags, batch, sat);
    HRAG_LAUNCH_CHECK();
    HRAG_TRY(reduce(sat, batch, HRAG_COMM_I32, HRAG_COMM_MAX));
    hipLaunchKernelGGL(sat_merge_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, s, flags, sat, batch, flags_out);
    HRAG_LAUNCH_CHECK();
    return HRAG_OK;
}

}  // extern "C"
