// libhrag.so C ABI (include/hrag.h): engine object, stage operators and the fused
// hrag_score_facts / hrag_retrieve / hrag_dense_retrieve / hrag_ppr entry points.
//
// hrag_engine_create stages what HippoRAG.prepare_retrieval_objects builds on the host
// (reference src/hipporag/HippoRAG.py:1287-1389) into device memory once; every compute call
// afterwards only enqueues kernels on the caller's stream (no allocation, no synchronisation).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "engine_impl.h"

namespace {

void free_store(Sell8Store &m) {
    void *ptrs[] = {m.pairs, m.pairs_at, m.chunk_meta, m.vrow, m.lrow_row, m.lrow_first, m.lrow_cnt, m.seg_lrow,
                    m.lcount, m.pslot};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    m = Sell8Store();
}

// the per-call buffers of an engine handle: everything hrag_workspace_create allocates afresh
static void workspace_ptrs(hrag_engine *e, std::vector<void **> &out) {
    void **ptrs[] = {(void **)&e->d_partial, (void **)&e->d_x, (void **)&e->d_y, (void **)&e->d_tele, (void **)&e->d_tele_dense,
                     (void **)&e->d_spass, (void **)&e->d_sfact, (void **)&e->d_doc, (void **)&e->d_mn_p, (void **)&e->d_mx_p,
                     (void **)&e->d_seed_vtx, (void **)&e->d_seed_cnt, (void **)&e->d_flags, (void **)&e->d_seed_w,
                     (void **)&e->d_colsum_partial, (void **)&e->d_sums, (void **)&e->d_partial16, (void **)&e->d_h16[0],
                     (void **)&e->d_h16[1], (void **)&e->d_h16[2], (void **)&e->d_h16[3], (void **)&e->d_tele16,
                     (void **)&e->d_row_slot, (void **)&e->d_qscale, (void **)&e->d_ssum, (void **)&e->d_tele_sv,
                     (void **)&e->d_partial_sv, (void **)&e->d_topk_ws, (void **)&e->d_R8, (void **)&e->d_rho8,
                     (void **)&e->d_partial8, (void **)&e->d_fused_ws, (void **)&e->d_mn_f, (void **)&e->d_mx_f,
                     (void **)&e->d_fused_sel, (void **)&e->d_xp8, (void **)&e->d_colmask, (void **)&e->d_stagep,
                     (void **)&e->d_pool8[0], (void **)&e->d_pool8[1], (void **)&e->d_pool8[2], (void **)&e->d_sv16[0],
                     (void **)&e->d_sv16[1], (void **)&e->d_sv16[2], (void **)&e->d_sv16[3], (void **)&e->d_zmax_bits,
                     (void **)&e->d_zmax, (void **)&e->d_mass, (void **)&e->d_prior_part, (void **)&e->d_est_f,
                     (void **)&e->d_ctl, (void **)&e->d_iters_used, (void **)&e->d_resid, (void **)&e->d_mass_tab,
                     (void **)&e->d_est_ws, (void **)&e->d_qsplit, (void **)&e->d_dyn, (void **)&e->d_mmax_ws,
                     (void **)&e->d_mmax_word, (void **)&e->sell.lcount, (void **)&e->fsell.lcount};
    out.assign(std::begin(ptrs), std::end(ptrs));
}

void free_engine(hrag_engine *e) {
    if (!e) return;
    std::vector<void **> ws;
    workspace_ptrs(e, ws);
    for (void **p : ws) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (!e->borrowed) {   // the immutable half: graph, embeddings, static tables, the SELL-8 matrices
        void *ptrs[] = {e->d_row_ptr, e->d_col, e->d_val, e->d_row_order, e->d_seg_row, e->d_seg_begin,
                        e->d_seg_end, e->d_seg_slot, e->d_mrow_row, e->d_mrow_first, e->d_mrow_cnt,
                        e->d_passage_vertex, e->d_row_to_tele, e->d_pemb, e->d_femb, e->d_subj, e->d_obj,
                        e->d_num_chunks, e->d_deg, e->d_pinvdeg, e->d_row_ptele, e->d_iso, e->d_piso, e->d_colmask_static};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        free_store(e->sell);
        free_store(e->fsell);
    } else if (e->parent) {
        e->parent->n_workspaces.v.fetch_sub(1);
    }
    for (auto &ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->ev_last) (void)hipEventDestroy(e->ev_last);
    delete e;
}

hrag_status check_batch(const hrag_engine *e, int32_t batch) {
    HRAG_REQUIRE(e != nullptr, "engine is NULL");
    if (batch < 1 || batch > e->max_batch) {
        set_error("batch %d outside [1, max_batch=%d]", batch, e->max_batch);
        return HRAG_ECAPACITY;
    }
    return HRAG_OK;
}

// seeds + teleport are applied the same way at init (scale 1) and in every sweep (scale 1-alpha)
hrag_status ppr_init(hrag_engine *e, const float *tele, int64_t tele_rows, const int32_t *r2t,
                     const int32_t *sv, const float *sw, const int32_t *sc, int batch, float *x,
                     SlabLayout lay, hipStream_t s) {
    SpmmArgs a = e->spmm_args(nullptr, x, tele, tele_rows, r2t, 0.f);
    HRAG_TRY(launch_ppr_init(a, lay, s));
    if (sv)
        HRAG_TRY(launch_seed_scatter(x, e->V, e->row_offset, e->n_rows, sv, sw, sc, kMaxSeeds, batch,
                                     1.0f, lay, s));
    return HRAG_OK;
}

hrag_status ppr_step(hrag_engine *e, const float *tele, int64_t tele_rows, const int32_t *r2t,
                     const int32_t *sv, const float *sw, const int32_t *sc, int batch, float damping,
                     const float *x, float *y, SlabLayout lay, bool main_only, hipStream_t s) {
    SpmmArgs a = e->spmm_args(x, y, tele, tele_rows, r2t, damping);
    HRAG_TRY(launch_ppr_spmm(a, lay, main_only, s));
    if (sv && !main_only)
        HRAG_TRY(launch_seed_scatter(y, e->V, e->row_offset, e->n_rows, sv, sw, sc, kMaxSeeds, batch,
                                     1.0f - damping, lay, s));
    return HRAG_OK;
}

// Longest (virtual) row of the SELL-8 matrix.  The wavefront that owns a row of n entries runs n / 8 dependent
// gather steps (~0.9 us each): a sweep cannot end before its longest row does, and rows are cut into segments of
// at most this many entries (the last-arriving segment adds the partial sums up inside the sweep kernel).
// Measured (profiles/r02k_segment_length.json): on the 2 M-entry graph of cfg 2 a 64-query sweep moves its
// 256 MB in ~38 us, and 512-entry rows (64 steps = 57 us) WERE the sweep: 41.3k queries/s at 512, 52.8k at 256,
// 60.0k at 128, 62.5k at 64, 51.8k at 32 (more virtual rows, partial sums and arrivals), 37.0k at 16; the same
// graph at 256 queries (two fp8 slabs) peaks at 128; 20 M entries x 2 slabs (cfg 3, 800 us sweeps) are flat from
// 256 to 1024 and 1 % slower at 128; one GPU's share of the hub-heavy 200 M-entry power-law graph of cfg 5 (25 M
// entries x 4 slabs, 18 ms sweeps) prefers 2048 (1294 queries/s; 1264 at 512: every segment pays ~4 us for its
// write-through and arrival; 1269 at 8192: imbalance).  Hence a rule in the work of one sweep, entries x 128-query slabs.
// HRAG_SELL8_SEG_LEN overrides (experiments).
int32_t sell8_seg_len(int64_t nnz_owned, int max_batch, int32_t asked = 0) {
    if (asked >= 8 && asked <= 1 << 20) return (int32_t)round_up(asked, 8);   // hrag_opts.sell_seg_len
    if (const char *env = experiment_env("HRAG_SELL8_SEG_LEN")) {
        const int v = std::atoi(env);
        if (v >= 8 && v <= 1 << 20) return (int32_t)round_up(v, 8);
    }
    const int64_t work = nnz_owned * (int64_t)n_slabs128(max_batch);
    return work <= (int64_t)3 << 20 ? 64 : work <= (int64_t)12 << 20 ? 128 : work <= (int64_t)30 << 20 ? 256
         : work <= (int64_t)64 << 20 ? kSell8SegLen : 2048;
}

// SELL-8 form of (a subset of) the owned CSR rows for ppr16.hip / ppr8.hip (see the header comments there).
// rows (may be null = all): LOCAL rows to include.  want_p: the (col, P value) pairs of ppr16 / ppr_sv.
// deg (may be null): weighted degrees by GLOBAL vertex id; when given, the pairs with the row-normalised
// values at_ij = p_ij d_j / d_i (the degree-scaled iteration of ppr8.hip) are built.
// row_passage (may be null): per LOCAL row, >= 0 when the row is a passage vertex -- the chunks that hold such a row get
// a dense slot number (Sell8Store::pslot): the scratch rows of the convergence contract's est measure.
hrag_status build_sell8(const hrag_engine *e, const std::vector<int32_t> &row_ptr, const int32_t *col,
                        const float *val, const double *deg, const std::vector<int32_t> *rows, bool want_p,
                        Sell8Store *out, const int32_t *row_passage = nullptr) {
    struct VRow { int32_t len, begin, target, row; };
    std::vector<VRow> vr;
    const int64_t n_sel = rows ? (int64_t)rows->size() : e->n_rows;
    vr.reserve((size_t)n_sel + 1024);
    std::vector<int32_t> lrow_row, lrow_first, lrow_cnt, seg_lrow;
    int32_t n_partial = 0;
    const int32_t max_len = sell8_seg_len(row_ptr[(size_t)e->n_rows], e->max_batch, e->sell_seg_len);
    for (int64_t k = 0; k < n_sel; ++k) {
        const int64_t r = rows ? (*rows)[(size_t)k] : k;
        const int32_t b0 = row_ptr[(size_t)r], len = row_ptr[(size_t)r + 1] - b0;
        if (len <= max_len) {
            vr.push_back({len, b0, (int32_t)r, (int32_t)r});
            continue;
        }
        // at most kSell8MaxSegs segments per row, each a multiple of 8 entries
        int32_t nseg = std::min<int32_t>((len + max_len - 1) / max_len, kSell8MaxSegs);
        const int32_t seg_len = (int32_t)round_up((len + nseg - 1) / nseg, 8);
        nseg = (len + seg_len - 1) / seg_len;
        lrow_row.push_back((int32_t)r);
        lrow_first.push_back(n_partial);
        lrow_cnt.push_back(nseg);
        for (int32_t i = 0; i < nseg; ++i) {
            seg_lrow.push_back((int32_t)lrow_row.size() - 1);
            vr.push_back({std::min(seg_len, len - i * seg_len), b0 + i * seg_len, -(n_partial++ + 1), (int32_t)r});
        }
    }
    // longest first (stable => deterministic); a chunk = 8 consecutive virtual rows.  hrag_opts.sell_sigma: the sort
    // stays inside windows of that many consecutive rows (SELL-C-sigma), so the processing order follows the vertex
    // numbering at window granularity -- what a numbering with locality needs for its gathers to hit the L2
    if (e->sell_sigma >= 8) {
        const size_t win = (size_t)round_up(e->sell_sigma, 8);
        for (size_t lo = 0; lo < vr.size(); lo += win)
            std::stable_sort(vr.begin() + (ptrdiff_t)lo, vr.begin() + (ptrdiff_t)std::min(vr.size(), lo + win),
                             [](const VRow &a, const VRow &b) { return a.len > b.len; });
    } else {
        std::stable_sort(vr.begin(), vr.end(), [](const VRow &a, const VRow &b) { return a.len > b.len; });
    }
    if (e->opt_flags & (HRAG_OPT_ROWS_BY_MINCOL | HRAG_OPT_ROWS_BFS)) {
        // EXPERIMENT ("row order and L2 reuse"): rows of equal length may be processed in any
        // order, so put rows that share in-neighbours next to each other -- they run on the same XCD at about the
        // same time and the second one could find the first one's gathered lines in that XCD's L2.
        //   BY_MINCOL: secondary key = the row's smallest column id;
        //   BFS:       secondary key = breadth-first rank of the row's vertex (from the vertex of largest degree,
        //              restarting at the next unvisited vertex), i.e. a Cuthill-McKee-like clustering.
        std::vector<int32_t> key((size_t)e->n_rows, 0);
        if (e->opt_flags & HRAG_OPT_ROWS_BFS) {
            std::vector<int32_t> rank((size_t)e->n_rows, -1), queue;
            queue.reserve((size_t)e->n_rows);
            std::vector<int32_t> by_deg((size_t)e->n_rows);
            std::iota(by_deg.begin(), by_deg.end(), 0);
            std::stable_sort(by_deg.begin(), by_deg.end(), [&](int32_t a, int32_t b) {
                return row_ptr[(size_t)a + 1] - row_ptr[(size_t)a] > row_ptr[(size_t)b + 1] - row_ptr[(size_t)b];
            });
            int32_t next_rank = 0;
            for (int32_t root : by_deg) {
                if (rank[(size_t)root] >= 0) continue;
                rank[(size_t)root] = next_rank++;
                queue.push_back(root);
                for (size_t h = queue.size() - 1; h < queue.size(); ++h) {
                    const int32_t u = queue[h];
                    for (int32_t k = row_ptr[(size_t)u]; k < row_ptr[(size_t)u + 1]; ++k) {
                        const int64_t w = (int64_t)col[k] - e->row_offset;
                        if (w >= 0 && w < e->n_rows && rank[(size_t)w] < 0) {
                            rank[(size_t)w] = next_rank++;
                            queue.push_back((int32_t)w);
                        }
                    }
                }
            }
            key = rank;
        } else {
            for (int64_t r = 0; r < e->n_rows; ++r)
                key[(size_t)r] = row_ptr[(size_t)r + 1] > row_ptr[(size_t)r] ? col[row_ptr[(size_t)r]] : 0;   // columns are sorted
        }
        std::stable_sort(vr.begin(), vr.end(), [&](const VRow &a, const VRow &b) {
            return a.len != b.len ? a.len > b.len : key[(size_t)a.row] < key[(size_t)b.row];
        });
    }
    const int64_t n_chunks = ceil_div((int64_t)vr.size(), 8);
    std::vector<int2> meta((size_t)n_chunks);
    std::vector<int32_t> vrow((size_t)n_chunks * 8, kVrowNone);
    int64_t steps = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int32_t ns = (int32_t)ceil_div(vr[(size_t)c * 8].len, 8);
        meta[(size_t)c] = make_int2((int)steps, ns);
        steps += ns;
    }
    HRAG_REQUIRE((steps + 4) * 512 < (int64_t)0x7fffffff, "graph too large for the SELL-8 buffer range (2 GiB)");
    std::vector<int2> pairs(want_p ? (size_t)(steps + 4) * 64 : 0, make_int2(0, 0));  // +4 steps: read-ahead padding
    std::vector<int2> pairs8(deg ? (size_t)(steps + 4) * 64 : 0, make_int2(0, 0));
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t base = (int64_t)meta[(size_t)c].x * 64;
        for (int g = 0; g < 8; ++g) {
            const int64_t vi = c * 8 + g;
            if (vi >= (int64_t)vr.size()) break;
            const VRow &v = vr[(size_t)vi];
            vrow[(size_t)vi] = v.target;
            for (int32_t i = 0; i < v.len; ++i) {
                int32_t bits;
                const size_t at = (size_t)(base + (int64_t)(i >> 3) * 64 + g * 8 + (i & 7));
                if (want_p) {
                    std::memcpy(&bits, &val[v.begin + i], 4);
                    pairs[at] = make_int2(col[v.begin + i], bits);
                }
                if (deg) {
                    const float atv = (float)((double)val[v.begin + i] * deg[col[v.begin + i]] /
                                              deg[(size_t)(e->row_offset + v.row)]);
                    std::memcpy(&bits, &atv, 4);
                    pairs8[at] = make_int2(col[v.begin + i], bits);
                }
            }
        }
    }
    if (row_passage) {
        // rows are sorted by length with ties in vertex order and the passages are the last vertices, so the passage
        // rows of a length class sit together: ~ Np / 8 chunks, not all of them
        std::vector<int32_t> pslot((size_t)n_chunks, -1);
        int32_t n_p = 0;
        for (int64_t c = 0; c < n_chunks; ++c)
            for (int g = 0; g < 8; ++g) {
                const int32_t t = vrow[(size_t)c * 8 + g];
                if (t >= 0 && row_passage[(size_t)t] >= 0) { pslot[(size_t)c] = n_p++; break; }
            }
        out->n_pchunks = n_p;
        HRAG_TRY(dev_upload(&out->pslot, pslot.data(), (int64_t)pslot.size()));
    }
    out->n_chunks = (int32_t)n_chunks;
    out->n_lrow = (int32_t)lrow_row.size();
    out->n_partial = n_partial;
    out->steps = steps;
    if (want_p) HRAG_TRY(dev_upload(&out->pairs, pairs.data(), (int64_t)pairs.size()));
    if (deg) HRAG_TRY(dev_upload(&out->pairs_at, pairs8.data(), (int64_t)pairs8.size()));
    HRAG_TRY(dev_upload(&out->chunk_meta, meta.data(), (int64_t)meta.size()));
    HRAG_TRY(dev_upload(&out->vrow, vrow.data(), (int64_t)vrow.size()));
    HRAG_TRY(dev_upload(&out->lrow_row, lrow_row.data(), (int64_t)lrow_row.size()));
    HRAG_TRY(dev_upload(&out->lrow_first, lrow_first.data(), (int64_t)lrow_first.size()));
    HRAG_TRY(dev_upload(&out->lrow_cnt, lrow_cnt.data(), (int64_t)lrow_cnt.size()));
    HRAG_TRY(dev_upload(&out->seg_lrow, seg_lrow.data(), (int64_t)seg_lrow.size()));
    return HRAG_OK;   // the arrival counters (lcount) are per workspace: alloc_workspace
}

Ppr16Args ppr16_args(const hrag_engine *e, const uint16_t *x, uint16_t *y, const uint16_t *aux,
                     float damping) {
    Ppr16Args a;
    a.pairs = e->sell.pairs; a.pairs_bytes = e->sell.pairs_bytes();
    a.chunk_meta = e->sell.chunk_meta; a.vrow = e->sell.vrow; a.n_chunks = e->sell.n_chunks;
    a.lrow_row = e->sell.lrow_row; a.lrow_first = e->sell.lrow_first; a.lrow_cnt = e->sell.lrow_cnt;
    a.n_lrow = e->sell.n_lrow; a.n_partial = e->sell.n_partial; a.partial = e->d_partial16;
    a.seg_lrow = e->sell.seg_lrow; a.lcount = e->sell.lcount;
    a.num_vertices = e->V; a.x = x; a.y = y; a.aux = aux; a.row_slot = e->d_row_slot;
    a.tele = e->d_tele16; a.tele_rows = e->tele16_rows;
    a.alpha = damping; a.beta = 1.0f - damping; a.cscale = kPpr16CScale;
    return a;
}

// HRAG_OPT_ACCEL on the two-stage fp16 states (ppr16.hip, ppr_sv.hip).  Both stages solve (I - G) y = rhs, G = a P, and
// on an undirected graph the spectrum of G is real, inside [-a, a]: the sweeps of a stage become Chebyshev steps
//     y_1 = G y_0 + rhs,   y_{k+1} = w_{k+1} (G y_k + rhs - y_{k-1}) + y_{k-1},   w_2 = 1 / (1 - a^2 / 2), w_{k+1} = 1 / (1 - a^2 w_k / 4)
// (y_{k+1} overwrites y_{k-1}: the ping-pong buffers are the history).  Stage 1: K1 steps on h from h_0 = v; the residual
// sweep is step 1 of stage 2 (c_0 = 0, c_1 = r); K2 more steps on c; then ONE plain correction sweep and the plain final
// sweep over the passage rows, so that (i) the mass of the result is the closed form of the plain iteration (the error
// polynomial vanishes at 0) and (ii) the convergence measure reads the update of a plain sweep that FOLLOWS a plain
// sweep -- the quantity it reads without the flag (csrc/shard.hip ppr8_plan_accel on why).  Error bound of the plan:
// 1 / (T_K1(1/a) T_{K2+1}(1/a)) a^2, each Chebyshev factor capped at 2^11 (the fp16 rounding of the stage's iterate);
// the smallest K1 + K2 that reaches a^iters / 16 is taken (accel_plan16 on the margin): a = 0.5, iters = 20 -> K1 = 7, K2 = 6,
// 16 sweeps (rounds 4 - 5: a^iters / 4, 14 sweeps).  Only with
// ppr_tol = 0 (`ppr_iters` names an accuracy).  Under a tolerance these states keep the plain plan (+ its device-side
// extension): after a Chebyshev stage h is less converged ELEMENTWISE than after as many plain sweeps (equi-oscillation
// puts error into every mode), the correction c is correspondingly larger, and its fp16 rounding (2^-11 |c| / 64) puts a
// floor of 3e-6 .. 1e-5 under the measured residual that no further correction sweep removes (measured: all four extension
// stages ran and the residual stayed there) -- above the mirror's tolerance, so every query would be flagged; the plain
// plan reads 2e-7 after 20 sweeps on the same graphs.
static double accel_omega(int k, double rho) {    // w_k of the recurrence above
    double w = 1.0;
    for (int j = 2; j <= k; ++j) w = j == 2 ? 1.0 / (1.0 - rho * rho / 2.0) : 1.0 / (1.0 - rho * rho * w / 4.0);
    return w;
}
// The two-stage fp16 states (ppr16.hip, ppr_sv.hip) split `iters` sweeps into K1 on h + the residual sweep + K2 on the
// correction c.  >= 16 sweeps: halves, as in every round.  11 .. 15 sweeps (round 6: the BASE count of a tolerance-driven
// call -- the contract's measure adds stages when the graph needs them, so a caller on a well-mixing graph need not pay
// for the worst-case count): K1 = 9 where the count allows -- the correction then starts at damping^9 = 2e-3 of the
// iterate and its fp16 rounding, 2^-11 of that = 9.5e-7 of the iterate, stays inside HRAG_PPR_ERR_FLOOR_F16 -- and at
// least one correction sweep + the final sweep.  Below 11 the fp32 state runs (K1 < 8: c too large for its fp16 rounding).
constexpr int kTwoStageMinIters = 11;
static inline void split16(int iters, int *k1, int *k2) {
    *k1 = iters >= 16 ? iters / 2 : std::min(9, iters - 3);
    *k2 = iters - *k1 - 1;
}
// ... which is an argument about damping^K1: a short count qualifies only where damping^K1 <= 2.2e-3 (0.5^9 = 1.95e-3;
// a larger damping factor keeps the fp32 state below 16 sweeps, as before round 6)
static inline bool two_stage_ok(int iters, float damping) {
    if (iters >= 16) return true;
    if (iters < kTwoStageMinIters) return false;
    int k1, k2;
    split16(iters, &k1, &k2);
    return std::pow((double)damping, (double)k1) <= 2.2e-3;
}

// A/B switch for measurements: HRAG_P8_GROUPWISE=0 keeps the fp8-state sweeps of a batch > 256 in one launch (read once)
static bool p8_groupwise_enabled() {
    static const bool v = [] { const char *x = experiment_env("HRAG_P8_GROUPWISE"); return !(x && x[0] == '0'); }();
    return v;
}

static bool accel_plan16(int iters, float damping, int *k1_out, int *k2_out) {
    const double al = (double)damping;
    // damping above ~0.6: the equi-oscillating error of a Chebyshev stage sits on the SMALL passage scores as well, and
    // their relative error ends 15x above the plain plan's (soak: damping 0.7, 20 sweeps for 39, 1.4e-5): plain plan there
    if (!(al >= 0.2) || al > 0.62 || iters < 16) return false;
    // x4: the plain iteration beats its own bound a^iters on well-mixing graphs (16k-vertex test graphs: 2e-7 after 20
    // sweeps, bound 9.5e-7) while a Chebyshev plan sits ON its bound (equi-oscillation); the margin keeps the accelerated
    // result within ~5x of the plain one there (measured 1.1e-6 .. 5e-6 without it)
    // round 6: two 10 - 20 minute runs of tools/soak_random.py found the margin of 4 too thin on sparse graphs with few
    // passages (mean degree 6, 5 % passages): damping 0.6, 28 -> 18 sweeps: 1.07e-5 on one query of 4 253 cases; damping
    // 0.5, 20 -> 14 sweeps at B = 5: 1.33e-5 on one of 8 876.  A fixed-count accelerated call has no measurement to fall
    // back on, so the margin is 16 everywhere (damping 0.5: 15 - 16 sweeps for 20; the two cases replay at < 4e-6)
    const double margin = 16.0;
    const double target = margin * std::pow(al, -(double)iters), cap = 2048.0;
    int best = iters, bk1 = 0, bk2 = 0;
    for (int k1 = 3; k1 <= 14; ++k1)
        for (int k2 = 2; k2 <= 14; ++k2) {
            const double f = std::min(cheb_T(k1, 1.0 / al), cap) * std::min(cheb_T(k2 + 1, 1.0 / al), cap) / (al * al);
            const int total = k1 + 1 + k2 + 2;
            if (f >= target && total < best) { best = total; bk1 = k1; bk2 = k2; }
        }
    if (best >= iters) return false;
    *k1_out = bk1; *k2_out = bk2;
    return true;
}

// h_0 = f16(v); K1 sweeps on h (the first one gathers only the columns where h_0 is non-zero: d_colmask); the
// residual sweep; K2 sweeps on the correction, the last of them over the passage rows only (fsell), writing
// x = h + c / cs in fp32 at the passages (d_xp8, passage order) -- nothing else is read afterwards
// (HippoRAG.py:1745).  Buffers: d_h16[0], [1] ping-pong for h; [2] = r; the free h buffer and [3] ping-pong for c.
// Extension stages the contract may add on the two-stage fp16 states (round 4): stage j + 1 of 1, 2, 3, 3 sweeps (the
// schedule of the fp8 state, p8_ext_sweeps) = that many plain correction sweeps over ALL rows -- the first one redoes,
// on every row, the sweep the previous final sweep did on the passage rows -- followed by a final sweep over the passage
// rows that measures again.  All of them are enqueued; ctl[j] (ppr8_decide_kernel after final sweep j) gates them.
static int ext_stages_for(int sweeps, int max_iters, float tol) {
    int e_max = 0;
    if (tol > 0.f)
        while (e_max < kP8MaxExt && sweeps + p8_ext_sweeps(e_max + 1) <= max_iters) ++e_max;
    return e_max;
}

// *sweeps_out: the sweeps of the base plan (HRAG_OPT_ACCEL: fewer than `iters`, which then names the accuracy);
// tol > 0 with est: up to *e_max_out extension stages follow, decided on the device (no host synchronisation)
hrag_status ppr16_run(hrag_engine *e, int batch, float damping, int iters, hipStream_t s, int32_t *est = nullptr,
                      int *sweeps_out = nullptr, bool allow_accel = false, float tol = 0.f, int max_iters = 0,
                      int *e_max_out = nullptr) {
    const int ns = n_slabs64(batch);
    const int nt = (e->opt_flags & HRAG_OPT_TEMPORAL16) ? 0 : 3;
    int k1, k2, kc = 0;                                // kc: Chebyshev steps among the k2 correction sweeps
    split16(iters, &k1, &k2);
    const bool accel = allow_accel && (e->opt_flags & HRAG_OPT_ACCEL) && accel_plan16(iters, damping, &k1, &kc);
    if (accel) k2 = kc + 2;                            // + one plain correction sweep + the final sweep
    const int sweeps = k1 + 1 + k2;
    if (sweeps_out) *sweeps_out = sweeps;
    uint16_t *h = e->d_h16[0], *hn = e->d_h16[1], *r = e->d_h16[2];
    HRAG_TRY(launch_ppr16_init(ppr16_args(e, nullptr, h, nullptr, damping), ns, s));
    for (int it = 0; it < k1; ++it) {
        Ppr16Args a = ppr16_args(e, h, hn, nullptr, damping);
        if (it == 0) { a.colmask = e->d_colmask; a.colmask_bytes = (uint32_t)(e->colmask_words * 4); }
        if (accel && it > 0) { a.omega = (float)accel_omega(it + 1, damping); a.prev = hn; }   // h_{it+1} over h_{it-1}
        HRAG_TRY(launch_ppr16_sweep(a, kPprModeH, ns, nt, false, s));
        std::swap(h, hn);
    }
    HRAG_TRY(launch_ppr16_sweep(ppr16_args(e, h, r, nullptr, damping), kPprModeR, ns, nt, false, s));
    const uint16_t *c = r;            // c_{K1+1} = r
    uint16_t *cn = hn, *cn2 = e->d_h16[3];
    // a correction sweep over all rows (c -> cn) / the final sweep over the passage rows from c (x = h + c' / cs, measured)
    auto sweep_c = [&](float omega, const uint16_t *prev, const int32_t *gate) {
        Ppr16Args a = ppr16_args(e, c, cn, r, damping);
        a.omega = omega; a.prev = prev; a.gate = gate; a.gate_want = 1;
        const hrag_status st = launch_ppr16_sweep(a, kPprModeC, ns, nt, false, s);
        c = cn;
        std::swap(cn, cn2);
        return st;
    };
    auto sweep_f = [&](const int32_t *gate) {
        Ppr16Args a = ppr16_args(e, c, cn, r, damping);
        const Sell8Store &m = e->fsell;
        a.pairs = m.pairs; a.pairs_bytes = m.pairs_bytes(); a.chunk_meta = m.chunk_meta; a.vrow = m.vrow;
        a.n_chunks = m.n_chunks; a.lrow_row = m.lrow_row; a.lrow_first = m.lrow_first; a.lrow_cnt = m.lrow_cnt;
        a.n_lrow = m.n_lrow; a.n_partial = m.n_partial; a.seg_lrow = m.seg_lrow; a.lcount = m.lcount;
        a.hfin = h; a.out = e->d_xp8; a.p_rows = e->p_rows;
        a.est = est; a.est_ws = e->d_est_ws; a.batch = batch;
        a.gate = gate; a.gate_want = 1;
        return launch_ppr16_sweep(a, kPprModeF, ns, nt, false, s);
    };
    for (int it = 0; it + 1 < k2; ++it) {
        // accelerated: step it + 2 of stage 2 -- c_0 = 0 (no history), c_1 = r, then the ping-pong buffers
        const bool cheb = accel && it < kc;
        HRAG_TRY(sweep_c(cheb ? (float)accel_omega(it + 2, damping) : 1.f, !cheb || it == 0 ? nullptr : it == 1 ? r : cn, nullptr));
    }
    HRAG_TRY(sweep_f(nullptr));
    const int e_max = est ? ext_stages_for(sweeps, max_iters, tol) : 0;
    if (e_max_out) *e_max_out = e_max;
    for (int j = 0; j < e_max; ++j) {
        HRAG_TRY(launch_ppr8_decide(est, e->d_flags, batch, damping / (1.0f - damping), tol, j, e_max, e->d_ctl, s));
        for (int i = p8_ext_sweeps(j); i < p8_ext_sweeps(j + 1); ++i) HRAG_TRY(sweep_c(1.f, nullptr, e->d_ctl + j));
        HRAG_TRY(sweep_f(e->d_ctl + j));
    }
    if (h != e->d_h16[0]) std::swap(e->d_h16[0], e->d_h16[1]);  // keep h in [0] for hrag_ppr_sweeps
    return HRAG_OK;
}

// The fp16 two-stage scheme needs K1 >= 8 sweeps before the residual sweep (error ~ 2^-(11+K1)): split16.
inline bool use_f16(const hrag_engine *e, int batch, int iters, float damping) {
    return e->f16_ready && !(e->opt_flags & HRAG_OPT_NO_F16) && batch > kSvMaxBatch && batch <= e->f16_max_batch &&
           two_stage_ok(iters, damping);
}
inline bool use_sv(const hrag_engine *e, int batch) { return e->sell_ready && batch <= kSvMaxBatch; }
inline int sv_width(int batch) { return batch <= 1 ? 1 : batch <= 2 ? 2 : batch <= 4 ? 4 : 8; }

PprSvArgs ppr_sv_args(const hrag_engine *e, const Sell8Store &m, const void *x, void *y, const int32_t *row_slot,
                      const float *tele, float damping) {
    PprSvArgs a;
    a.pairs = m.pairs; a.pairs_bytes = m.pairs_bytes(); a.chunk_meta = m.chunk_meta;
    a.vrow = m.vrow; a.n_chunks = m.n_chunks;
    a.lrow_row = m.lrow_row; a.lrow_first = m.lrow_first; a.lrow_cnt = m.lrow_cnt;
    a.n_lrow = m.n_lrow; a.partial = e->d_partial_sv; a.num_vertices = e->V;
    a.seg_lrow = m.seg_lrow; a.lcount = m.lcount;
    a.x = x; a.y = y; a.row_slot = row_slot; a.tele = tele;
    a.alpha = damping; a.beta = 1.0f - damping;
    a.nt = (e->opt_flags & HRAG_OPT_NT_CSR) ? 1 : 0;   // measured: nt loads are 25 % slower at B = 1
    return a;
}

// the two-stage fp16 state of the small-batch path needs K1 >= 8 sweeps before the residual sweep
inline bool use_sv_half(const hrag_engine *e, int iters, float damping) {
    return e->d_sv16[0] != nullptr && !(e->opt_flags & HRAG_OPT_NO_F16) && two_stage_ok(iters, damping);
}

// hrag_ppr (all rows wanted): x_0 = v, `iters` plain sweeps; the final state ends in e->d_x ([V][bp] fp32)
hrag_status ppr_sv_run_full(hrag_engine *e, const int32_t *row_slot, const float *tele, int bp, float damping,
                            int iters, hipStream_t s) {
    float *x = e->d_x, *y = e->d_y;
    HRAG_TRY(launch_ppr_sv_init(ppr_sv_args(e, e->sell, nullptr, x, row_slot, tele, damping), bp, s));
    for (int it = 0; it < iters; ++it) {
        HRAG_TRY(launch_ppr_sv_sweep(ppr_sv_args(e, e->sell, x, y, row_slot, tele, damping), bp, false, s));
        std::swap(x, y);
    }
    if (x != e->d_x) std::swap(e->d_x, e->d_y);
    return HRAG_OK;
}

// hrag_retrieve, batch <= 8 (ppr_sv.hip header): x_0 = v; the first sweep gathers only the passage / seed columns
// (d_colmask), the last one runs over the passage rows only and leaves x (fp32 [V][bp], passage rows valid) in
// e->d_x; with >= 16 sweeps the state in between is the two-stage fp16 one (v must carry the per-query scale).
hrag_status ppr_sv_run(hrag_engine *e, const int32_t *row_slot, const float *tele, int bp, float damping,
                       int iters, hipStream_t s, int32_t *est = nullptr, int batch = 0, int *sweeps_out = nullptr,
                       bool allow_accel = false, float tol = 0.f, int max_iters = 0, int *e_max_out = nullptr) {
    if (sweeps_out) *sweeps_out = iters;
    if (e_max_out) *e_max_out = 0;
    if (iters < 1) return ppr_sv_run_full(e, row_slot, tele, bp, damping, iters, s);
    if (!use_sv_half(e, iters, damping)) {
        float *x = e->d_x, *y = e->d_y;
        HRAG_TRY(launch_ppr_sv_init(ppr_sv_args(e, e->sell, nullptr, x, row_slot, tele, damping), bp, s));
        for (int it = 0; it < iters; ++it) {
            PprSvArgs a = ppr_sv_args(e, it + 1 == iters ? e->fsell : e->sell, x, y, row_slot, tele, damping);
            a.colmask = it == 0 ? e->d_colmask : nullptr;
            if (it + 1 == iters) { a.est = est; a.est_ws = e->d_est_ws; a.batch = batch; }
            HRAG_TRY(launch_ppr_sv_sweep(a, bp, false, s));
            std::swap(x, y);
        }
        if (x != e->d_x) std::swap(e->d_x, e->d_y);
        return HRAG_OK;
    }
    int k1, k2, kc = 0;                                // the plan of ppr16_run, Chebyshev steps included (accel_plan16)
    split16(iters, &k1, &k2);
    const bool accel = allow_accel && (e->opt_flags & HRAG_OPT_ACCEL) && accel_plan16(iters, damping, &k1, &kc);
    if (accel) k2 = kc + 2;
    const int sweeps = k1 + 1 + k2;
    if (sweeps_out) *sweeps_out = sweeps;
    uint16_t *h = e->d_sv16[0], *hn = e->d_sv16[1], *r = e->d_sv16[2];
    {
        PprSvArgs a = ppr_sv_args(e, e->sell, nullptr, h, row_slot, tele, damping);
        a.half_state = 1;
        HRAG_TRY(launch_ppr_sv_init(a, bp, s));
    }
    for (int it = 0; it < k1; ++it) {
        PprSvArgs a = ppr_sv_args(e, e->sell, h, hn, row_slot, tele, damping);
        a.half_state = 1; a.mode = 0; a.colmask = it == 0 ? e->d_colmask : nullptr;
        if (accel && it > 0) { a.omega = (float)accel_omega(it + 1, damping); a.prev = hn; }
        HRAG_TRY(launch_ppr_sv_sweep(a, bp, false, s));
        std::swap(h, hn);
    }
    {
        PprSvArgs a = ppr_sv_args(e, e->sell, h, r, row_slot, tele, damping);
        a.half_state = 1; a.mode = 1; a.cscale = kPpr16CScale;
        HRAG_TRY(launch_ppr_sv_sweep(a, bp, false, s));
    }
    const uint16_t *c = r;                 // c_{K1+1} = r
    uint16_t *cn = hn, *cn2 = e->d_sv16[3];
    // a correction sweep over all rows (c -> cn) / the final sweep over the passage rows from c (see ppr16_run)
    auto sweep_c = [&](float omega, const uint16_t *prev, const int32_t *gate) {
        PprSvArgs a = ppr_sv_args(e, e->sell, c, cn, row_slot, tele, damping);
        a.half_state = 1; a.mode = 2; a.aux16 = r; a.cscale = kPpr16CScale; a.h16 = h; a.xout = e->d_x;
        a.omega = omega; a.prev = prev; a.gate = gate; a.gate_want = 1;
        const hrag_status st = launch_ppr_sv_sweep(a, bp, false, s);
        c = cn;
        std::swap(cn, cn2);
        return st;
    };
    auto sweep_f = [&](const int32_t *gate) {
        PprSvArgs a = ppr_sv_args(e, e->fsell, c, cn, row_slot, tele, damping);
        a.half_state = 1; a.mode = 3; a.aux16 = r; a.cscale = kPpr16CScale; a.h16 = h; a.xout = e->d_x;
        a.est = est; a.est_ws = e->d_est_ws; a.batch = batch;
        a.gate = gate; a.gate_want = 1;
        return launch_ppr_sv_sweep(a, bp, false, s);
    };
    for (int j = 0; j + 1 < k2; ++j) {
        const bool cheb = accel && j < kc;
        HRAG_TRY(sweep_c(cheb ? (float)accel_omega(j + 2, damping) : 1.f, !cheb || j == 0 ? nullptr : j == 1 ? r : cn, nullptr));
    }
    HRAG_TRY(sweep_f(nullptr));
    const int e_max = est ? ext_stages_for(sweeps, max_iters, tol) : 0;
    if (e_max_out) *e_max_out = e_max;
    for (int j = 0; j < e_max; ++j) {   // extension stages of the contract, decided on the device (ppr16_run)
        HRAG_TRY(launch_ppr8_decide(est, e->d_flags, batch, damping / (1.0f - damping), tol, j, e_max, e->d_ctl, s));
        for (int i = p8_ext_sweeps(j); i < p8_ext_sweeps(j + 1); ++i) HRAG_TRY(sweep_c(1.f, nullptr, e->d_ctl + j));
        HRAG_TRY(sweep_f(e->d_ctl + j));
    }
    return HRAG_OK;
}

}  // namespace

// The per-call half of an engine handle, sized once for max_batch: scores, PPR states of every width, seeds, flags,
// counters.  hrag_engine_create builds it after the index; hrag_workspace_create builds ONLY this for a handle that
// borrows the index of another engine.  Needs e->want_sell / want_f16 / has_facts / f8_ready and the SELL-8 metadata.
static hrag_status alloc_workspace(hrag_engine *e) {
    const bool unsharded = e->n_rows == e->V;
    tl_alloc_bytes = &e->workspace_bytes;
    struct Unbook { ~Unbook() { tl_alloc_bytes = nullptr; } } unbook;
    // ---- workspace, sized once for max_batch
    const int B = e->max_batch;
    SlabLayout lay = e->layout(B);
    e->state_elems = unsharded ? (int64_t)lay.n_slabs * e->V * lay.bc : 0;   // d_x / d_y: hrag_retrieve, hrag_ppr
    if (e->want_sell) {
        HRAG_TRY(dev_alloc(&e->d_tele_sv, (e->n_passages + (int64_t)kSvMaxBatch * kMaxSeeds) * kSvMaxBatch));
        HRAG_TRY(dev_alloc(&e->d_partial_sv, (int64_t)std::max({e->sell.n_partial, e->fsell.n_partial, 1}) * kSvMaxBatch));
        for (auto &p : e->d_sv16) {   // two-stage fp16 state of the small-batch path: h ping / pong, r, c
            HRAG_TRY(dev_alloc(&p, e->V * (int64_t)kSvMaxBatch));
            HRAG_HIP_TRY(hipMemset(p, 0, (size_t)e->V * kSvMaxBatch * sizeof(uint16_t)));
        }
        e->sell_ready = true;
    }
    if (e->want_sell || e->f8_ready) {
        // per-batch copy of the row -> teleport slot map with the seed rows patched in (local rows)
        HRAG_TRY(dev_alloc(&e->d_row_slot, e->n_rows));
        HRAG_HIP_TRY(hipMemcpy(e->d_row_slot, e->f8_ready ? e->d_row_ptele : e->d_row_to_tele,
                        (size_t)e->n_rows * sizeof(int32_t), hipMemcpyDeviceToDevice));
    }
    if (e->want_sell && !(e->want_f16 || e->f8_ready)) {   // small-batch-only engine: per-query scale of the fp16 state
        HRAG_TRY(dev_alloc(&e->d_qscale, B));
        HRAG_TRY(dev_alloc(&e->d_ssum, B));
    }
    if (e->want_f16 || e->f8_ready) {
        // teleport rows of the fp16 and fp8 paths: the owned passages, then the seed rows (fp32, 64-query slabs)
        const int ns = n_slabs64(B);
        e->tele16_rows = e->p_rows + (int64_t)B * kMaxSeeds;
        HRAG_TRY(dev_alloc(&e->d_tele16, (int64_t)ns * e->tele16_rows * 64));
        HRAG_HIP_TRY(hipMemset(e->d_tele16, 0, (size_t)ns * e->tele16_rows * 64 * sizeof(float)));
        HRAG_TRY(dev_alloc(&e->d_qscale, B));
        HRAG_TRY(dev_alloc(&e->d_ssum, B));
    }
    if (e->want_f16) {
        const int ns = n_slabs64(B);
        // the fp16 STATE only has to hold the batches the fp8 path does not take (<= 64 queries)
        e->f16_max_batch = e->f8_ready ? std::min(B, 64) : B;
        e->state16_elems = (int64_t)n_slabs64(e->f16_max_batch) * e->V * 64;
        e->state_elems = std::max(e->state_elems, e->state16_elems);  // d_x also receives h + c
        for (auto &p : e->d_h16) HRAG_TRY(dev_alloc(&p, e->state16_elems));
        HRAG_TRY(dev_alloc(&e->d_partial16, (int64_t)ns * std::max(e->sell.n_partial, 1) * 64));
        for (auto &p : e->d_h16) HRAG_HIP_TRY(hipMemset(p, 0, (size_t)e->state16_elems * sizeof(uint16_t)));
        if (!e->f8_ready) HRAG_TRY(dev_alloc(&e->d_xp8, (int64_t)ns * std::max<int64_t>(e->p_rows, 1) * 64));
        e->f16_ready = true;
    }
    if (e->f8_ready) {
        const int ns = n_slabs128(B);
        HRAG_TRY(dev_alloc(&e->d_R8, (int64_t)ns * std::max<int64_t>(e->n_rows, 1) * 128));
        HRAG_TRY(dev_alloc(&e->d_rho8, (int64_t)ns * std::max<int64_t>(e->n_rows, 1) * 128));
        HRAG_TRY(dev_alloc(&e->d_partial8, (int64_t)ns * std::max({e->sell.n_partial, e->fsell.n_partial, 1}) * 128));
        HRAG_TRY(dev_alloc(&e->d_stagep, (int64_t)kP8MaxStages * ns * std::max<int64_t>(e->p_rows, 1) * 128));
        HRAG_TRY(dev_alloc(&e->d_xp8, (int64_t)n_slabs64(B) * std::max<int64_t>(e->p_rows, 1) * 64));
        HRAG_TRY(dev_alloc(&e->d_zmax_bits, B));
        HRAG_TRY(dev_alloc(&e->d_zmax, B));
        HRAG_TRY(dev_alloc(&e->d_mass, 2 * (int64_t)B));
        HRAG_TRY(dev_alloc(&e->d_prior_part, (int64_t)kP8PriorSplit * B * 2));
        if (unsharded) {
            // hrag_retrieve's own state buffers: groups of two slabs, [group][V + 1][2][128] (ppr8_layout)
            e->state8_bytes = (int64_t)round_up(ns, 2) * (e->V + 1) * 128;
            for (auto &p : e->d_pool8) {
                HRAG_TRY(dev_alloc(&p, e->state8_bytes));
                HRAG_HIP_TRY(hipMemset(p, 0, (size_t)e->state8_bytes));
            }
        }
    }
    HRAG_TRY(dev_alloc(&e->d_partial, (int64_t)(round_up(B, 4) + 64) * std::max(e->n_partial, 1)));
    HRAG_TRY(dev_alloc(&e->d_x, e->state_elems));
    HRAG_TRY(dev_alloc(&e->d_y, e->state_elems));
    if (unsharded) HRAG_TRY(dev_alloc(&e->d_tele, (int64_t)lay.n_slabs * std::max<int64_t>(e->n_passages, 1) * lay.bc));
    e->ld_p = round_up(std::max<int64_t>(e->p_rows, 1), 4);   // score rows cover the OWNED passages
    e->ld_f = round_up(std::max<int64_t>(e->f_rows, 1), 4);
    HRAG_TRY(dev_alloc(&e->d_spass, (int64_t)B * e->ld_p));
    HRAG_TRY(dev_alloc(&e->d_doc, (int64_t)B * e->ld_p));
    if (e->has_facts) HRAG_TRY(dev_alloc(&e->d_sfact, (int64_t)B * e->ld_f));
    if (e->has_facts && B > 16) {
        HRAG_TRY(dev_alloc(&e->d_fused_ws, 2 * sim_fused_tiles(std::max<int64_t>(e->f_rows, 1)) * B));
        HRAG_TRY(dev_alloc(&e->d_fused_sel, sim_fused_sel_ints(B)));
        HRAG_HIP_TRY(hipMemset(e->d_fused_sel, 0, (size_t)sim_fused_sel_ints(B) * sizeof(int32_t)));
        HRAG_TRY(dev_alloc(&e->d_mn_f, B));
        HRAG_TRY(dev_alloc(&e->d_mx_f, B));
    }
    HRAG_TRY(dev_alloc(&e->d_mn_p, B));
    HRAG_TRY(dev_alloc(&e->d_mx_p, B));
    HRAG_TRY(dev_alloc(&e->d_seed_vtx, (int64_t)B * kMaxSeeds));
    HRAG_TRY(dev_alloc(&e->d_seed_w, (int64_t)B * kMaxSeeds));
    HRAG_TRY(dev_alloc(&e->d_seed_cnt, B));
    HRAG_TRY(dev_alloc(&e->d_flags, B));
    // colsum partials: worst case is the narrowest slab (most slabs * bc stays ~B, padded)
    HRAG_TRY(dev_alloc(&e->d_colsum_partial, (int64_t)kColsumBlocks * (round_up(B, 4) + 64)));
    HRAG_TRY(dev_alloc(&e->d_sums, B));
    HRAG_TRY(dev_alloc(&e->d_est_f, B));
    HRAG_TRY(dev_alloc(&e->d_ctl, kP8MaxExt + 1));
    HRAG_TRY(dev_alloc(&e->d_iters_used, B));
    HRAG_TRY(dev_alloc(&e->d_resid, B));
    {
        // every wavefront of a sweep that measures est writes one row of (queries per slab row) floats; the widest
        // user is the final sweep of the fp8 state (the chunks that hold passage rows, 128 queries per row)
        const int64_t c_p = e->fsell.n_pchunks, c_f = e->fsell.n_chunks;
        int64_t n = c_f * 8;                                                   // small batches (last sweep: passage rows)
        if (e->f8_ready) n = std::max(n, (int64_t)n_slabs128(B) * c_p * 128);   // the chunks that hold passage rows
        if (e->f16_ready) n = std::max(n, (int64_t)n_slabs64(e->f16_max_batch) * c_f * 64);
        HRAG_TRY(dev_alloc(&e->d_est_ws, std::max<int64_t>(n, 1)));
    }
    HRAG_TRY(dev_alloc(&e->d_mass_tab, (int64_t)(kP8MaxExt + 1) * B));
    if (e->f8_ready) {   // HRAG_OPT_ACCEL: measured stage scales
        e->mmax_slots = (int64_t)(e->sell.n_chunks + 8) * n_slabs128(B);
        HRAG_TRY(dev_alloc(&e->d_dyn, 2 * kP8DynInv));
        HRAG_TRY(dev_alloc(&e->d_mmax_ws, e->mmax_slots));
        HRAG_TRY(dev_alloc(&e->d_mmax_word, 2));   // [0] running maximum (float bits), [1] arrival counter of the reduction
        HRAG_HIP_TRY(hipMemset(e->d_dyn, 0, 2 * kP8DynInv * sizeof(float)));
        HRAG_HIP_TRY(hipMemset(e->d_mmax_ws, 0, (size_t)e->mmax_slots * sizeof(float)));
        HRAG_HIP_TRY(hipMemset(e->d_mmax_word, 0, 2 * sizeof(int32_t)));
    }
    {
        char *ws = nullptr;
        HRAG_TRY(dev_alloc(&ws, (int64_t)kTopkWsBytes));
        e->d_topk_ws = ws;
    }
    if (e->state_elems) {
        HRAG_HIP_TRY(hipMemset(e->d_x, 0, (size_t)e->state_elems * sizeof(float)));
        HRAG_HIP_TRY(hipMemset(e->d_y, 0, (size_t)e->state_elems * sizeof(float)));
    }
    HRAG_HIP_TRY(hipMemset(e->d_seed_cnt, 0, (size_t)B * sizeof(int32_t)));
    // arrival counters of the long rows (one set per handle: two handles may sweep the same matrix at once)
    for (Sell8Store *m : {&e->sell, &e->fsell}) {
        const int64_t n_cnt = (int64_t)n_slabs64(e->max_batch) * (int64_t)m->n_lrow;
        HRAG_TRY(dev_alloc(&m->lcount, n_cnt));
        if (n_cnt > 0) HRAG_HIP_TRY(hipMemset(m->lcount, 0, (size_t)n_cnt * sizeof(int32_t)));
    }
    if (e->colmask_words > 0) HRAG_TRY(dev_alloc(&e->d_colmask, e->colmask_words));   // per-batch copy of the column bitmap
    for (auto &ev : e->ev) HRAG_HIP_TRY(hipEventCreate(&ev));
    HRAG_HIP_TRY(hipEventCreateWithFlags(&e->ev_last, hipEventDisableTiming));
    return HRAG_OK;
}

extern "C" {

hrag_status hrag_engine_create(const hrag_graph_desc *g, const hrag_embed_desc *facts,
                               const hrag_embed_desc *passages, const hrag_fact_desc *fd,
                               const hrag_opts *opts, hrag_engine **out) {
    HRAG_REQUIRE(g && passages && opts && out, "graph, passages, opts and out must be non-NULL");
    *out = nullptr;
    HRAG_REQUIRE(g->num_vertices > 0 && g->num_vertices < (int64_t)0x7fffffff, "bad num_vertices");
    HRAG_REQUIRE(g->n_rows >= 0 && g->row_offset >= 0 && g->row_offset + g->n_rows <= g->num_vertices,
                 "owned row range [%lld, +%lld) outside the graph", (long long)g->row_offset,
                 (long long)g->n_rows);
    HRAG_REQUIRE(g->nnz >= 0 && g->nnz < (int64_t)0x7fffffff, "nnz must fit int32");
    HRAG_REQUIRE(g->row_ptr && (g->nnz == 0 || (g->col_idx && g->val)), "CSR arrays missing");
    HRAG_REQUIRE(g->n_passages >= 0 && (g->n_passages == 0 || g->passage_vertex), "passage_vertex missing");
    auto split_dt = [](int dt) { return dt == HRAG_F32_SPLIT || dt == HRAG_F32_SPLIT_ROWS; };
    HRAG_REQUIRE(passages->dtype == HRAG_BF16 || passages->dtype == HRAG_FP16 || split_dt(passages->dtype),
                 "embeddings must be bf16, fp16 or fp32 (HRAG_F32_SPLIT / HRAG_F32_SPLIT_ROWS)");
    HRAG_REQUIRE(!facts || facts->dtype == passages->dtype || (split_dt(facts->dtype) && split_dt(passages->dtype)),
                 "fact / passage embedding dtypes differ");
    HRAG_REQUIRE(passages->dim > 0 && passages->dim % 8 == 0, "embedding dim must be a multiple of 8");
    HRAG_REQUIRE(!facts || facts->dim == passages->dim, "fact / passage dims differ");
    HRAG_REQUIRE((facts == nullptr) == (fd == nullptr), "facts and fact_desc go together");
    HRAG_REQUIRE(passages->row_offset >= 0 && passages->row_offset + passages->rows <= g->n_passages,
                 "passage shard outside [0, n_passages)");
    HRAG_REQUIRE(!facts || (fd->n_facts >= 0 && facts->row_offset >= 0 &&
                            facts->row_offset + facts->rows <= fd->n_facts && fd->subj_vertex &&
                            fd->obj_vertex && fd->num_chunks),
                 "fact shard outside [0, n_facts) or fact_desc arrays missing");
    HRAG_REQUIRE(opts->max_batch >= 1, "max_batch must be >= 1");
    HRAG_REQUIRE(opts->max_topk >= 1 && opts->max_topk <= kTopkMax, "max_topk outside [1, %d]", kTopkMax);
    const int sw = opts->slab_width;
    HRAG_REQUIRE(sw == 0 || sw == 4 || sw == 8 || sw == 16 || sw == 32 || sw == 64,
                 "slab_width must be 0 or one of 4, 8, 16, 32, 64");

    if (opts->device >= 0) HRAG_HIP_TRY(hipSetDevice(opts->device));
    hrag_engine *e = new (std::nothrow) hrag_engine();
    if (!e) { set_error("out of host memory"); return HRAG_ENOMEM; }
    HRAG_HIP_TRY(hipGetDevice(&e->device));
    hrag_status st = HRAG_OK;
    tl_alloc_bytes = &e->index_bytes;      // everything allocated until alloc_workspace is the (shareable) index
    struct Unbook { ~Unbook() { tl_alloc_bytes = nullptr; } } unbook;
#define E_TRY(expr) do { st = (expr); if (st != HRAG_OK) { free_engine(e); return st; } } while (0)
#define E_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s -> %s", #expr, hipGetErrorString(_e)); free_engine(e); return HRAG_EHIP; } } while (0)

    e->V = g->num_vertices; e->row_offset = g->row_offset; e->n_rows = g->n_rows; e->nnz = g->nnz;
    e->n_passages = g->n_passages;
    e->max_batch = opts->max_batch; e->max_topk = opts->max_topk;
    e->slab_cap = sw ? sw : 32;
    // a short row is walked by G lanes in deg/G dependent gather rounds: cap it at 8 rounds
    e->short_thresh = opts->long_row_nnz > 0 ? opts->long_row_nnz : 8 * (e->slab_cap / 4);
    e->seg_len = opts->segment_nnz > 0 ? (int)round_up(opts->segment_nnz, 64) : 512;
    e->sell_seg_len = opts->sell_seg_len;
    e->sell_sigma = opts->sell_sigma;
    e->opt_flags = opts->flags;

    // ---- CSR to the device; row lists on the host (copy row_ptr back if it came from the device)
    std::vector<int32_t> h_row_ptr((size_t)e->n_rows + 1);
    E_HIP(hipMemcpy(h_row_ptr.data(), g->row_ptr, h_row_ptr.size() * sizeof(int32_t), hipMemcpyDefault));
    if (h_row_ptr[0] != 0 || h_row_ptr[(size_t)e->n_rows] != (int32_t)e->nnz) {
        set_error("row_ptr[0]=%d / row_ptr[n_rows]=%d inconsistent with nnz=%lld", h_row_ptr[0],
                  h_row_ptr[(size_t)e->n_rows], (long long)e->nnz);
        free_engine(e);
        return HRAG_EINVAL;
    }
    E_TRY(dev_upload(&e->d_row_ptr, h_row_ptr.data(), (int64_t)h_row_ptr.size()));
    E_TRY(dev_upload(&e->d_col, g->col_idx, e->nnz));
    E_TRY(dev_upload(&e->d_val, g->val, e->nnz));
    {
        // short rows in degree-descending order (stable => deterministic); longer rows are cut
        // into segments of <= seg_len entries (one wavefront each)
        std::vector<int32_t> order, seg_row, seg_begin, seg_end, seg_slot, mrow_row, mrow_first, mrow_cnt;
        order.reserve((size_t)e->n_rows);
        int32_t n_partial = 0;
        for (int64_t r = 0; r < e->n_rows; ++r) {
            const int32_t b0 = h_row_ptr[(size_t)r], e0 = h_row_ptr[(size_t)r + 1];
            const int32_t deg = e0 - b0;
            if (deg <= e->short_thresh) {
                if (deg > 0) order.push_back((int32_t)r);
                continue;
            }
            ++e->n_long_rows;
            const int32_t nseg = (deg + e->seg_len - 1) / e->seg_len;
            if (nseg > 1) {
                mrow_row.push_back((int32_t)r);
                mrow_first.push_back(n_partial);
                mrow_cnt.push_back(nseg);
            }
            for (int32_t sidx = 0; sidx < nseg; ++sidx) {
                seg_row.push_back((int32_t)r);
                seg_begin.push_back(b0 + sidx * e->seg_len);
                seg_end.push_back(std::min(e0, b0 + (sidx + 1) * e->seg_len));
                seg_slot.push_back(nseg > 1 ? n_partial++ : -1);
            }
        }
        if (!(opts->flags & HRAG_OPT_NATURAL_ROW_ORDER))
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                return h_row_ptr[(size_t)a + 1] - h_row_ptr[(size_t)a] > h_row_ptr[(size_t)b + 1] - h_row_ptr[(size_t)b];
            });
        // rows without entries still need y = (1-alpha) v written: append them at the end
        for (int64_t r = 0; r < e->n_rows; ++r)
            if (h_row_ptr[(size_t)r + 1] == h_row_ptr[(size_t)r]) order.push_back((int32_t)r);
        e->n_short = (int32_t)order.size();
        e->n_seg = (int32_t)seg_row.size();
        e->n_mrow = (int32_t)mrow_row.size();
        e->n_partial = n_partial;
        E_TRY(dev_upload(&e->d_row_order, order.data(), (int64_t)order.size()));
        E_TRY(dev_upload(&e->d_seg_row, seg_row.data(), (int64_t)seg_row.size()));
        E_TRY(dev_upload(&e->d_seg_begin, seg_begin.data(), (int64_t)seg_begin.size()));
        E_TRY(dev_upload(&e->d_seg_end, seg_end.data(), (int64_t)seg_end.size()));
        E_TRY(dev_upload(&e->d_seg_slot, seg_slot.data(), (int64_t)seg_slot.size()));
        E_TRY(dev_upload(&e->d_mrow_row, mrow_row.data(), (int64_t)mrow_row.size()));
        E_TRY(dev_upload(&e->d_mrow_first, mrow_first.data(), (int64_t)mrow_first.size()));
        E_TRY(dev_upload(&e->d_mrow_cnt, mrow_cnt.data(), (int64_t)mrow_cnt.size()));
    }
    // ---- passages: vertex map and its inverse on the owned rows
    std::vector<int32_t> h_pv((size_t)e->n_passages);
    std::vector<int32_t> h_r2t((size_t)e->n_rows, -1);
    e->p_rows = passages->rows; e->p_offset = passages->row_offset;
    {
        if (e->n_passages)
            E_HIP(hipMemcpy(h_pv.data(), g->passage_vertex, h_pv.size() * sizeof(int32_t), hipMemcpyDefault));
        int64_t owned_in_range = 0, owned_total = 0;
        for (int64_t p = 0; p < e->n_passages; ++p) {
            const int64_t v = h_pv[(size_t)p];
            if (v < 0 || v >= e->V) {
                set_error("passage_vertex[%lld]=%lld outside [0, V)", (long long)p, (long long)v);
                free_engine(e);
                return HRAG_EINVAL;
            }
            const int64_t lr = v - e->row_offset;
            if (lr >= 0 && lr < e->n_rows) {
                h_r2t[(size_t)lr] = (int32_t)p;
                ++owned_total;
                if (p >= e->p_offset && p < e->p_offset + e->p_rows) ++owned_in_range;
            }
        }
        // the passage prior stays on its GPU when the passage embedding shard == the passages of the owned rows
        e->shard_aligned = owned_total == e->p_rows && owned_in_range == e->p_rows;
        E_TRY(dev_upload(&e->d_passage_vertex, h_pv.data(), e->n_passages));
        E_TRY(dev_upload(&e->d_row_to_tele, h_r2t.data(), e->n_rows));
    }
    // ---- SELL-8 matrices: P-valued for the fp16 / small-batch kernels (unsharded engines), At-valued for the
    //      staged fp8 iteration (any row shard, needs the weighted degrees the caller normalised P with)
    const bool unsharded = e->n_rows == e->V;
    const bool want_sell = !(opts->flags & HRAG_OPT_F32_STATE) && unsharded && e->V * 128 < ((int64_t)1 << 32);
    const bool want_f16 = want_sell && opts->max_batch > kSvMaxBatch;
    const bool want_f8 = !(opts->flags & HRAG_OPT_F32_STATE) && !(opts->flags & HRAG_OPT_NO_FP8) && g->col_sum &&
                         e->V + 1 <= ((int64_t)1 << 24) && e->shard_aligned &&
                         (unsharded ? (want_f16 && opts->max_batch > 64) : true);
    e->fp8_unavailable = want_f8 ? 0
        : ((opts->flags & (HRAG_OPT_F32_STATE | HRAG_OPT_NO_FP8)) ? HRAG_FP8_UNAVAILABLE_DISABLED : 0) |
          (!g->col_sum ? HRAG_FP8_UNAVAILABLE_NO_COL_SUM : 0) |
          (e->V + 1 > ((int64_t)1 << 24) ? HRAG_FP8_UNAVAILABLE_TOO_MANY_VERTICES : 0) |
          (!e->shard_aligned ? HRAG_FP8_UNAVAILABLE_SHARD_NOT_ALIGNED : 0) |
          (unsharded && !(want_f16 && opts->max_batch > 64) ? HRAG_FP8_UNAVAILABLE_SMALL_MAX_BATCH : 0);
    if (want_sell || want_f8) {
        std::vector<int32_t> h_col((size_t)e->nnz);
        std::vector<float> h_val((size_t)e->nnz);
        if (e->nnz) {
            E_HIP(hipMemcpy(h_col.data(), g->col_idx, h_col.size() * sizeof(int32_t), hipMemcpyDefault));
            E_HIP(hipMemcpy(h_val.data(), g->val, h_val.size() * sizeof(float), hipMemcpyDefault));
        }
        std::vector<double> h_deg;
        std::vector<uint8_t> h_iso;
        if (want_f8) {
            h_deg.resize((size_t)e->V);
            h_iso.assign((size_t)e->V, 0);
            E_HIP(hipMemcpy(h_deg.data(), g->col_sum, h_deg.size() * sizeof(double), hipMemcpyDefault));
            for (int64_t i = 0; i < e->V; ++i) {
                double &d = h_deg[(size_t)i];
                if (!(d >= 0.0) || !(d < 1e300)) {
                    set_error("col_sum must be finite and >= 0");
                    free_engine(e);
                    return HRAG_EINVAL;
                }
                if (d == 0.0) { d = 1.0; h_iso[(size_t)i] = 1; }   // isolated vertex: no entries, z = x
            }
            // P d = d must hold (P = A D^-1 with A symmetric): it makes At = D^-1 P D row-stochastic,
            // which is what keeps the static fp8 scales of ppr8.hip valid (checked on the owned rows)
            double worst = 0.0;
            for (int64_t r = 0; r < e->n_rows; ++r) {
                double acc = 0.0;
                for (int32_t k = h_row_ptr[(size_t)r]; k < h_row_ptr[(size_t)r + 1]; ++k)
                    acc += (double)h_val[(size_t)k] * h_deg[(size_t)h_col[(size_t)k]];
                if (h_row_ptr[(size_t)r + 1] > h_row_ptr[(size_t)r])
                    worst = std::max(worst, std::fabs(acc / h_deg[(size_t)(e->row_offset + r)] - 1.0));
                else if (!h_iso[(size_t)(e->row_offset + r)])
                    worst = 1.0;   // a vertex with weight but no row entries: not a symmetric adjacency
            }
            if (worst > 1e-3) {
                set_error("col_sum is not the weighted degree of a symmetric adjacency: max |sum_j P_ij d_j / d_i - 1| = %.3g",
                          worst);
                free_engine(e);
                return HRAG_EINVAL;
            }
        }
        E_TRY(build_sell8(e, h_row_ptr, h_col.data(), h_val.data(), want_f8 ? h_deg.data() : nullptr, nullptr,
                          want_sell, &e->sell, h_r2t.data()));
        // the last sweep only needs the passage rows (HippoRAG.py:1745 reads nothing else): a second matrix
        std::vector<int32_t> prow_list;
        std::vector<int32_t> ptele((size_t)e->n_rows, -1);
        for (int64_t r = 0; r < e->n_rows; ++r)
            if (h_r2t[(size_t)r] >= 0) {
                prow_list.push_back((int32_t)r);
                ptele[(size_t)r] = h_r2t[(size_t)r] - (int32_t)e->p_offset;
            }
        E_TRY(build_sell8(e, h_row_ptr, h_col.data(), h_val.data(), want_f8 ? h_deg.data() : nullptr, &prow_list,
                          want_sell, &e->fsell, h_r2t.data()));
        if (!want_f8) {
            // isolated vertices (the closed-form mass of the small-batch path): no row entries <=> no edges (symmetric)
            h_iso.assign((size_t)e->V, 0);
            for (int64_t r = 0; r < e->n_rows; ++r)
                if (h_row_ptr[(size_t)r + 1] == h_row_ptr[(size_t)r]) h_iso[(size_t)(e->row_offset + r)] = 1;
        }
        {
            E_TRY(dev_upload(&e->d_iso, h_iso.data(), e->V));
            std::vector<uint8_t> piso((size_t)e->p_rows);
            for (int64_t q = 0; q < e->p_rows; ++q) piso[(size_t)q] = h_iso[(size_t)h_pv[(size_t)(e->p_offset + q)]];
            E_TRY(dev_upload(&e->d_piso, piso.data(), e->p_rows));
            // column bitmap of the passage vertices (ALL passages: the columns are global)
            e->colmask_words = ceil_div(e->V + 1, 32);
            std::vector<uint32_t> mask((size_t)e->colmask_words, 0u);
            for (int64_t q = 0; q < e->n_passages; ++q) mask[(size_t)(h_pv[(size_t)q] >> 5)] |= 1u << (h_pv[(size_t)q] & 31);
            E_TRY(dev_upload(&e->d_colmask_static, mask.data(), e->colmask_words));
        }
        if (want_f8) {
            E_TRY(dev_upload(&e->d_row_ptele, ptele.data(), e->n_rows));
            std::vector<float> f_deg((size_t)e->V);
            for (int64_t i = 0; i < e->V; ++i) f_deg[(size_t)i] = (float)h_deg[(size_t)i];
            E_TRY(dev_upload(&e->d_deg, f_deg.data(), e->V));
            std::vector<float> pinv((size_t)e->p_rows);
            for (int64_t q = 0; q < e->p_rows; ++q)
                pinv[(size_t)q] = 1.0f / f_deg[(size_t)h_pv[(size_t)(e->p_offset + q)]];
            E_TRY(dev_upload(&e->d_pinvdeg, pinv.data(), e->p_rows));
            e->f8_ready = true;   // buffers follow with the workspace
        }
    }
    // ---- embeddings + fact lookup arrays
    e->dim = passages->dim;
    e->split = split_dt(passages->dtype);
    e->emb_dtype = e->split ? (int32_t)HRAG_FP16 : (int32_t)passages->dtype;   // the halves of a split vector are fp16
    e->kdim = e->split ? 3 * e->dim : e->dim;
    // fp32 rows are split on the device into [hi | lo | hi] (knn.hip); the fp32 copy is transient
    auto upload_emb = [&](uint16_t **dst, const void *data, int64_t rows, int dt) -> hrag_status {
        if (!e->split) return dev_upload(dst, static_cast<const uint16_t *>(data), rows * e->dim);
        if (dt == HRAG_F32_SPLIT_ROWS) return dev_upload(dst, static_cast<const uint16_t *>(data), rows * e->kdim);
        HRAG_TRY(dev_alloc(dst, rows * e->kdim));
        float *tmp = nullptr;
        HRAG_TRY(dev_upload(&tmp, static_cast<const float *>(data), rows * e->dim));
        hrag_status st = launch_split3(tmp, rows, e->dim, 0, *dst, nullptr);
        hipError_t err = hipDeviceSynchronize();
        if (tmp) (void)hipFree(tmp);
        if (st == HRAG_OK && err != hipSuccess) { set_error("split of the fp32 embeddings failed: %s", hipGetErrorString(err)); st = HRAG_EHIP; }
        return st;
    };
    // passages->data == NULL: an engine WITHOUT passage embeddings -- the raw passage scores come from the caller
    // (hrag_retrieve_scored: the PPR side of the hybrid multi-GPU mode, whose embeddings live row-sharded elsewhere)
    if (passages->data) E_TRY(upload_emb(&e->d_pemb, passages->data, e->p_rows, passages->dtype));
    if (e->split) E_TRY(dev_alloc(&e->d_qsplit, (int64_t)opts->max_batch * e->kdim));
    if (facts) {
        e->f_rows = facts->rows; e->f_offset = facts->row_offset; e->n_facts = fd->n_facts;
        E_TRY(upload_emb(&e->d_femb, facts->data, e->f_rows, facts->dtype));
        E_TRY(dev_upload(&e->d_subj, fd->subj_vertex, e->n_facts));
        E_TRY(dev_upload(&e->d_obj, fd->obj_vertex, e->n_facts));
        E_TRY(dev_upload(&e->d_num_chunks, fd->num_chunks, e->V));
    }
    // ---- workspace, sized once for max_batch
    e->want_sell = want_sell; e->want_f16 = want_f16; e->has_facts = facts != nullptr;
    tl_alloc_bytes = nullptr;
    E_TRY(alloc_workspace(e));
    E_HIP(hipDeviceSynchronize());
#undef E_TRY
#undef E_HIP
    *out = e;
    return HRAG_OK;
}

hrag_status hrag_engine_destroy(hrag_engine *e) {
    if (e) {
        if (!e->borrowed && e->n_workspaces.v.load() > 0) {
            set_error("engine still has %d live workspace(s) (hrag_workspace_create): destroy them first -- they borrow "
                      "this engine's index buffers", e->n_workspaces.v.load());
            return HRAG_EINVAL;
        }
        (void)hipSetDevice(e->device);
        (void)hipDeviceSynchronize();
        free_engine(e);
    }
    return HRAG_OK;
}

// "engine immutable after create -> concurrent read-only calls allowed on distinct streams with distinct
// workspaces".  The new handle shares every index buffer of `e` (graph, SELL-8 matrices, embeddings, static tables: no
// second copy of the 1.5 GB of embeddings) and owns a full set of per-call buffers, its own entry flag, events, option
// flags and timings -- so a call on it never meets HRAG_EBUSY because of a call on `e` or on another workspace.
hrag_status hrag_workspace_create(hrag_engine *e, hrag_engine **out) {
    HRAG_REQUIRE(e != nullptr && out != nullptr, "engine and out must be non-NULL");
    *out = nullptr;
    hrag_engine *root = e->borrowed ? e->parent : e;
    HRAG_REQUIRE(root != nullptr, "workspace without a parent engine");
    HRAG_HIP_TRY(hipSetDevice(root->device));
    hrag_engine *w = new (std::nothrow) hrag_engine(*root);      // index pointers, sizes, options: copied
    if (!w) { set_error("out of host memory"); return HRAG_ENOMEM; }
    w->borrowed = true;
    w->parent = root;
    std::vector<void **> ws;
    workspace_ptrs(w, ws);
    for (void **p : ws) *p = nullptr;                            // ... the per-call half: its own
    for (auto &ev : w->ev) ev = nullptr;
    w->ev_last = nullptr; w->last_stream = nullptr; w->have_last = false;
    w->profiling = false; w->have_retrieve_ev = false; w->have_fact_ev = false;
    w->last = hrag_timings{};
    w->p8 = Ppr8Session();
    w->sell_ready = w->f16_ready = false;                        // set again by alloc_workspace
    w->state_elems = w->state16_elems = w->state8_bytes = 0;
    w->index_bytes = 0; w->workspace_bytes = 0;
    w->last_ppr_state = 0;
    root->n_workspaces.v.fetch_add(1);
    const hrag_status st = alloc_workspace(w);
    if (st != HRAG_OK) { free_engine(w); return st; }
    HRAG_HIP_TRY(hipDeviceSynchronize());
    *out = w;
    return HRAG_OK;
}

hrag_status hrag_engine_stats(hrag_engine *e, hrag_stats *out) {
    HRAG_REQUIRE(e != nullptr && out != nullptr, "engine and out must be non-NULL");
    hrag_stats s = {};
    s.is_workspace = e->borrowed ? 1 : 0;
    s.live_workspaces = e->borrowed ? 0 : e->n_workspaces.v.load();
    s.index_bytes = e->borrowed && e->parent ? e->parent->index_bytes : e->index_bytes;
    s.workspace_bytes = e->workspace_bytes;
    s.ppr_states = HRAG_PPR_STATE_F32 | (e->f16_ready ? HRAG_PPR_STATE_F16 : 0) | (e->sell_ready ? HRAG_PPR_STATE_SMALL : 0) |
                   (e->f8_ready ? HRAG_PPR_STATE_FP8 : 0);
    s.fp8_unavailable = e->fp8_unavailable;
    s.last_ppr_state = e->last_ppr_state;
    s.calls_score_facts = e->counters.score_facts.load();
    s.calls_retrieve = e->counters.retrieve.load();
    s.calls_dense_retrieve = e->counters.dense.load();
    s.calls_ppr = e->counters.ppr.load();
    s.calls_shard = e->counters.shard.load();
    s.queries = e->counters.queries.load();
    *out = s;
    return HRAG_OK;
}

hrag_status hrag_set_profiling(hrag_engine *e, int32_t enabled) {
    HRAG_REQUIRE(e != nullptr, "engine is NULL");
    e->profiling = enabled != 0;
    return HRAG_OK;
}

hrag_status hrag_get_timings(hrag_engine *e, hrag_timings *out) {
    HRAG_REQUIRE(e && out, "NULL argument");
    hrag_timings t = e->last;
    if (e->have_retrieve_ev) {
        HRAG_HIP_TRY(hipEventSynchronize(e->ev[EV_RANK]));
        HRAG_HIP_TRY(hipEventElapsedTime(&t.pass_sim_ms, e->ev[EV_START], e->ev[EV_SIM]));
        HRAG_HIP_TRY(hipEventElapsedTime(&t.seed_ms, e->ev[EV_SIM], e->ev[EV_SEED]));
        HRAG_HIP_TRY(hipEventElapsedTime(&t.ppr_ms, e->ev[EV_SEED], e->ev[EV_PPR]));
        HRAG_HIP_TRY(hipEventElapsedTime(&t.rank_ms, e->ev[EV_PPR], e->ev[EV_RANK]));
        HRAG_HIP_TRY(hipEventElapsedTime(&t.total_ms, e->ev[EV_START], e->ev[EV_RANK]));
    }
    if (e->have_fact_ev) {
        HRAG_HIP_TRY(hipEventSynchronize(e->ev[EV_FACT1]));
        HRAG_HIP_TRY(hipEventElapsedTime(&t.fact_sim_ms, e->ev[EV_FACT0], e->ev[EV_FACT1]));
    }
    t.n_long_rows = e->n_long_rows;
    *out = t;
    return HRAG_OK;
}

hrag_status hrag_ppr_layout(hrag_engine *e, int32_t batch, int32_t *bc, int32_t *n_slabs) {
    HRAG_TRY(check_batch(e, batch));
    SlabLayout l = e->layout(batch);
    if (bc) *bc = l.bc;
    if (n_slabs) *n_slabs = l.n_slabs;
    return HRAG_OK;
}

hrag_status hrag_sim_scores(hrag_engine *e, int32_t which, const uint16_t *q, int32_t batch,
                            float *out, hrag_stream stream) {
    HRAG_REQUIRE(e && q && out, "NULL argument");
    HRAG_REQUIRE(batch >= 1, "batch must be >= 1");
    HRAG_REQUIRE(which == 0 || which == 1, "which must be 0 (facts) or 1 (passages)");
    HRAG_ENGINE_CALL(e, stream);
    HRAG_TRY(prep_query(e, q, batch, (hipStream_t)stream, &q));
    if (which == 0) {
        HRAG_REQUIRE(e->d_femb != nullptr || e->f_rows == 0, "engine has no fact embeddings");
        return launch_sim_gemm(e->d_femb, e->f_rows, e->kdim, q, batch, out, e->f_rows, (hipStream_t)stream, 0, e->emb_dtype);
    }
    HRAG_REQUIRE(e->d_pemb != nullptr || e->p_rows == 0, "engine has no passage embeddings (created for hrag_retrieve_scored)");
    return launch_sim_gemm(e->d_pemb, e->p_rows, e->kdim, q, batch, out, e->p_rows, (hipStream_t)stream, 0, e->emb_dtype);
}

hrag_status hrag_row_minmax(const float *scores, int32_t batch, int64_t n, int64_t ld, float *mn,
                            float *mx, hrag_stream stream) {
    HRAG_REQUIRE(scores && mn && mx && batch >= 1 && n >= 1 && ld >= n, "bad argument");
    return launch_row_minmax(scores, batch, n, ld, mn, mx, (hipStream_t)stream);
}

hrag_status hrag_topk_rows(const float *scores, int32_t batch, int64_t n, int64_t ld, int32_t k,
                           int32_t idx_offset, int32_t normalize, int32_t *idx_out, float *val_out,
                           float *mn, float *mx, hrag_stream stream) {
    HRAG_REQUIRE(scores && idx_out && val_out && batch >= 1 && n >= 0 && ld >= n, "bad argument");
    return launch_row_topk(scores, batch, n, ld, k, idx_offset, normalize ? kNormMinMax : kNormNone,
                           idx_out, val_out, mn, mx, (hipStream_t)stream);
}

hrag_status hrag_score_facts(hrag_engine *e, const uint16_t *q, int32_t batch, int32_t k,
                             int32_t *idx_out, float *score_out, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(q && idx_out && score_out, "NULL argument");
    HRAG_REQUIRE(e->d_sfact != nullptr || e->f_rows == 0, "engine has no fact embeddings");
    HRAG_REQUIRE(e->f_rows == e->n_facts, "hrag_score_facts needs the whole fact matrix; a sharded "
                 "engine goes through hrag_sim_scores + hrag_topk_rows + an all-gather");
    hipStream_t s = (hipStream_t)stream;
    HRAG_ENGINE_CALL(e, stream);
    e->counters.score_facts.fetch_add(1);
    if (e->profiling) HRAG_HIP_TRY(hipEventRecord(e->ev[EV_FACT0], s));
    HRAG_TRY(prep_query(e, q, batch, s, &q));
    if (batch > 16 && k <= 16 && e->f_rows > 0 && e->d_fused_ws) {
        // no [B, F] score matrix: tile maxima -> k tiles per query -> exact top-k of k * 128 recomputed
        // scores (bit-identical to the two-step path below; sim_gemm.hip)
        HRAG_TRY(launch_sim_topk_fused(e->d_femb, e->f_rows, e->kdim, q, batch, k, 0, 1, e->d_fused_ws,
                                       e->d_fused_sel, e->d_mn_f, e->d_mx_f, idx_out, score_out, s, e->emb_dtype));
    } else {
        HRAG_TRY(launch_sim_gemm(e->d_femb, e->f_rows, e->kdim, q, batch, e->d_sfact, e->ld_f, s, 0, e->emb_dtype));
        // get_fact_scores' min_max_normalize + rerank_facts' argsort prefix in one kernel
        HRAG_TRY(launch_row_topk(e->d_sfact, batch, e->f_rows, e->ld_f, k, 0, kNormMinMax, idx_out, score_out,
                                 nullptr, nullptr, s, e->d_topk_ws, kTopkWsBytes));
    }
    if (e->profiling) {
        HRAG_HIP_TRY(hipEventRecord(e->ev[EV_FACT1], s));
        e->have_fact_ev = true;
    }
    return HRAG_OK;
}

hrag_status hrag_stage_seeds(hrag_engine *e, const int32_t *kept_idx, const float *kept_score,
                             const int32_t *kept_count, int32_t kf, int32_t link_top_k, int32_t batch,
                             int32_t *seed_vtx, float *seed_w, int32_t *seed_cnt, int32_t *flags,
                             hrag_stream stream) {
    HRAG_REQUIRE(e && kept_idx && kept_score && kept_count && seed_vtx && seed_w && seed_cnt && flags,
                 "NULL argument");
    HRAG_REQUIRE(e->d_subj != nullptr, "engine was created without fact_desc");
    return launch_build_seeds(kept_idx, kept_score, kept_count, kf, link_top_k, batch, e->d_subj, e->d_obj,
                              e->n_facts, e->d_num_chunks, e->V, seed_vtx, seed_w, seed_cnt, flags,
                              (hipStream_t)stream);
}

hrag_status hrag_stage_teleport(hrag_engine *e, const float *scores, int64_t ld, const float *mn,
                                const float *mx, float weight, const int32_t *flags, int32_t batch,
                                float *tele_out, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(scores && mn && mx && tele_out && ld >= e->n_passages, "bad argument");
    return launch_rows_to_slab(scores, ld, e->n_passages, batch, kMinMaxScale, mn, mx, weight, flags,
                               tele_out, e->layout(batch), (hipStream_t)stream);
}

hrag_status hrag_stage_ppr_init(hrag_engine *e, const float *tele, const int32_t *sv, const float *sw,
                                const int32_t *sc, int32_t batch, float *x, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(tele && x, "NULL argument");
    return ppr_init(e, tele, e->n_passages, e->d_row_to_tele, sv, sw, sc, batch, x, e->layout(batch),
                    (hipStream_t)stream);
}

hrag_status hrag_stage_ppr_step(hrag_engine *e, const float *tele, const int32_t *sv, const float *sw,
                                const int32_t *sc, int32_t batch, float damping, const float *x,
                                float *y, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(tele && x && y && x != y, "bad argument");
    return ppr_step(e, tele, e->n_passages, e->d_row_to_tele, sv, sw, sc, batch, damping, x, y,
                    e->layout(batch), false, (hipStream_t)stream);
}

int64_t hrag_colsum_workspace_bytes(hrag_engine *e, int32_t batch) {
    if (!e || batch < 1) return 0;
    SlabLayout l = e->layout(batch);
    return (int64_t)l.n_slabs * kColsumBlocks * l.bc * (int64_t)sizeof(double);
}

hrag_status hrag_stage_colsum(hrag_engine *e, const float *x, int32_t batch, void *ws, double *sums,
                              hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(x && ws && sums, "NULL argument");
    return launch_colsum(x, e->V, e->row_offset, e->n_rows, batch, e->layout(batch),
                         static_cast<double *>(ws), sums, (hipStream_t)stream);
}

hrag_status hrag_stage_doc_scores(hrag_engine *e, const float *x, const double *sums, int32_t batch,
                                  const float *scores, int64_t ld, const float *mn, const float *mx,
                                  int32_t *flags, float *out, int64_t out_ld, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(x && sums && out && out_ld >= e->n_passages, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    SlabLayout lay = e->layout(batch);
    HRAG_TRY(launch_slab_to_rows(x, e->V, e->d_passage_vertex, e->n_passages, batch, sums, out, out_ld,
                                 scores, ld, mn, mx, flags, lay, s));
    if (flags) HRAG_TRY(launch_flag_zero_mass(sums, batch, flags, 2, s));
    return HRAG_OK;
}

// hrag_retrieve / hrag_retrieve_scored: the passage scores come from this engine's embeddings (q_pass) or from the
// caller (pass_scores fp32 [batch, pass_ld] raw cosine scores in passage order)
static hrag_status retrieve_impl(hrag_engine *e, const uint16_t *q_pass, const float *pass_scores, int64_t pass_ld,
                                 int32_t batch, const int32_t *kept_idx, const float *kept_score,
                                 const int32_t *kept_count, int32_t kf, int32_t link_top_k, float damping,
                                 float passage_node_weight, int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol,
                                 int32_t k, int32_t *doc_idx_out, float *doc_score_out, int32_t *flags_out,
                                 float *residual_out, int32_t *iters_out, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(ppr_tol >= 0.f && ppr_tol == ppr_tol, "ppr_tol must be >= 0");
    HRAG_REQUIRE(ppr_tol == 0.f || ppr_tol >= HRAG_PPR_TOL_MIN,
                 "ppr_tol=%g is below HRAG_PPR_TOL_MIN=%g: fp32 arithmetic leaves an error floor of %g however small the "
                 "measured residual reads (include/hrag.h: error <= max(%g * residual, floor))", (double)ppr_tol,
                 (double)HRAG_PPR_TOL_MIN, (double)HRAG_PPR_ERR_FLOOR_F32, (double)HRAG_PPR_ERR_K);
    HRAG_REQUIRE(ppr_tol == 0.f || ppr_max_iters >= ppr_iters, "ppr_max_iters=%d < ppr_iters=%d", ppr_max_iters, ppr_iters);
    HRAG_REQUIRE((q_pass || pass_scores) && kept_idx && kept_score && kept_count && doc_idx_out && doc_score_out,
                 "NULL argument");
    HRAG_REQUIRE(e->n_rows == e->V && e->p_rows == e->n_passages,
                 "hrag_retrieve needs an unsharded engine; sharded engines use the hrag_stage_* operators");
    HRAG_REQUIRE(pass_scores || e->d_pemb, "the engine was created without passage embeddings: use hrag_retrieve_scored");
    HRAG_REQUIRE(!pass_scores || pass_ld >= e->n_passages, "pass_ld=%lld < n_passages", (long long)pass_ld);
    HRAG_REQUIRE(k >= 1 && k <= e->max_topk, "k=%d outside [1, max_topk=%d]", k, e->max_topk);
    HRAG_REQUIRE(ppr_iters >= 0, "ppr_iters must be >= 0");
    HRAG_REQUIRE(damping >= 0.f && damping < 1.f, "damping %g outside [0, 1)", (double)damping);
    HRAG_REQUIRE(e->n_passages >= 1, "engine has no passages");
    hipStream_t s = (hipStream_t)stream;
    HRAG_ENGINE_CALL(e, stream);
    const bool sv = use_sv(e, batch);
    const bool f8 = !sv && batch > 64 && e->d_pool8[0] && ppr8_usable(e, batch, ppr_iters, damping);
    const int f8_iters = ppr_iters;
    const bool f16 = !sv && !f8 && use_f16(e, batch, ppr_iters, damping);
    const int bp = sv_width(batch);
    SlabLayout lay = e->layout(batch);
    if (f16 || f8) { lay.bc = 64; lay.n_slabs = n_slabs64(batch); }
    if (sv) { lay.bc = bp; lay.n_slabs = 1; }
    const bool prof = e->profiling;

    // convergence contract on the fp16 / small-batch / fp32 states: the fixed count runs, the last sweep measures the
    // relative update of the passage scores (residual_out, flags bit 4); the fp8 and the fp16 states extend on the
    // device (gate words, ext_max below), the fp32 state reports and leaves the repeat to the caller
    int sweeps_run = ppr_iters;   // fp16 states under HRAG_OPT_ACCEL: fewer (ppr_iters then names the accuracy)
    int ext_max = 0;              // fp16 states: extension stages enqueued (the device decides which of them run)
    const bool want_est = residual_out != nullptr || ppr_tol > 0.f;
    int32_t *est = (want_est && !f8 && ppr_iters >= 1) ? e->d_est_f : nullptr;
    {
        BlitList z;   // the call's small fills in one launch (common.h)
        z.zero(e->d_flags, (int64_t)batch * sizeof(int32_t));
        z.zero(est, (int64_t)batch * sizeof(int32_t));
        if (!f8) z.zero(e->d_ctl, (int64_t)(kP8MaxExt + 1) * sizeof(int32_t));   // read by the finalize step below
        HRAG_TRY(launch_blits(z, s));
    }
    if (prof) HRAG_HIP_TRY(hipEventRecord(e->ev[EV_START], s));
    // dense_passage_retrieval: raw scores + min / max (HippoRAG.py:1496-1498)
    if (pass_scores) {
        HRAG_HIP_TRY(hipMemcpy2DAsync(e->d_spass, (size_t)e->ld_p * sizeof(float), pass_scores,
                                      (size_t)pass_ld * sizeof(float), (size_t)e->n_passages * sizeof(float),
                                      (size_t)batch, hipMemcpyDeviceToDevice, s));
    } else {
        HRAG_TRY(prep_query(e, q_pass, batch, s, &q_pass));
        HRAG_TRY(launch_sim_gemm(e->d_pemb, e->p_rows, e->kdim, q_pass, batch, e->d_spass, e->ld_p, s, 0, e->emb_dtype));
    }
    const bool sv_half = sv && use_sv_half(e, ppr_iters, damping);
    HRAG_TRY(launch_row_minmax(e->d_spass, batch, e->n_passages, e->ld_p, e->d_mn_p, e->d_mx_p, s,
                               (f16 || sv_half) ? e->d_ssum : nullptr));
    if (prof) HRAG_HIP_TRY(hipEventRecord(e->ev[EV_SIM], s));
    // reset vector: entity seeds + passage prior (HippoRAG.py:1574-1638)
    if (e->d_subj) {
        HRAG_TRY(hrag_stage_seeds(e, kept_idx, kept_score, kept_count, kf, link_top_k, batch, e->d_seed_vtx,
                                  e->d_seed_w, e->d_seed_cnt, e->d_flags, stream));
    } else {
        // an index without facts (no triples extracted): every query takes the DPR ranking, as the
        // reference does when rerank_facts returns nothing (HippoRAG.py:467-469, :1453-1455)
        HRAG_TRY(launch_fill_i32(e->d_flags, 1, batch, s));
        HRAG_HIP_TRY(hipMemsetAsync(e->d_seed_cnt, 0, (size_t)batch * sizeof(int32_t), s));
    }
    if (f8) {
        // staged fp8 state (ppr8.hip / shard.hip): the single-GPU engine is the row shard that owns everything
        hrag_shard_layout sl;
        HRAG_TRY(ppr8_layout(e, batch, 0, &sl));   // narrowest groups: two adjacent slabs per vertex
        HRAG_TRY(ppr8_prior(e, e->d_mn_p, e->d_mx_p, passage_node_weight, e->d_flags, batch, e->d_zmax, e->d_mass, s));
        HRAG_TRY(ppr8_begin(e, e->d_mn_p, e->d_mx_p, e->d_zmax, e->d_mass, passage_node_weight, e->d_seed_vtx,
                            e->d_seed_w, e->d_seed_cnt, e->d_flags, batch, damping, f8_iters, sl, e->d_pool8, s,
                            ppr_max_iters, ppr_tol, residual_out != nullptr, true));
    } else if (f16) {
        // v is scaled per query by a power of two so that every iterate fits fp16 (ppr16.hip); the seeds
        // become extra teleport rows, i.e. v is one array that every sweep reads identically
        HRAG_TRY(launch_ppr16_scale(e->d_mn_p, e->d_mx_p, e->d_ssum, e->n_passages, passage_node_weight,
                                    e->d_seed_w, e->d_seed_cnt, e->d_flags, batch, e->d_qscale, s));
        HRAG_TRY(launch_rows_to_slab(e->d_spass, e->ld_p, e->n_passages, batch, kMinMaxScale, e->d_mn_p,
                                     e->d_mx_p, passage_node_weight, e->d_flags, e->d_tele16, lay, s,
                                     e->tele16_rows, e->d_qscale));
        {
            // seed rows of the teleport matrix start from zero, row -> teleport-row map and the column bitmap (the
            // columns where h_0 = f16(v) can be non-zero: passage vertices + this batch's seeds) from their static parts
            BlitList z;
            z.zero(e->d_tele16 + (size_t)e->n_passages * 64, (int64_t)batch * kMaxSeeds * 64 * sizeof(float), lay.n_slabs,
                   (int64_t)e->tele16_rows * 64 * sizeof(float));
            z.copy(e->d_row_slot, e->d_row_to_tele, (int64_t)e->V * sizeof(int32_t));
            z.copy(e->d_colmask, e->d_colmask_static, (int64_t)e->colmask_words * sizeof(uint32_t));
            HRAG_TRY(launch_blits(z, s));
        }
        HRAG_TRY(launch_ppr16_seed_rows(e->d_seed_vtx, e->d_seed_w, e->d_seed_cnt, e->d_qscale, batch,
                                        e->n_passages, e->V, e->d_row_slot, e->d_tele16, e->tele16_rows, 64, s));
        HRAG_TRY(launch_ppr8_mask_seeds(e->d_seed_vtx, e->d_seed_cnt, batch, e->V, e->d_colmask, s));
    } else if (sv) {
        // small batch (ppr_sv.hip): v = [Np + seed rows][bp] fp32, same "seeds are teleport rows" form; with the
        // fp16 state v carries the per-query power-of-two scale of ppr16.hip (every iterate fits fp16)
        const float *qs = nullptr;
        if (sv_half) {
            HRAG_TRY(launch_ppr16_scale(e->d_mn_p, e->d_mx_p, e->d_ssum, e->n_passages, passage_node_weight,
                                        e->d_seed_w, e->d_seed_cnt, e->d_flags, batch, e->d_qscale, s));
            qs = e->d_qscale;
        }
        HRAG_TRY(launch_ppr_sv_tele(e->d_spass, e->ld_p, e->n_passages, batch, e->d_mn_p, e->d_mx_p,
                                    passage_node_weight, e->d_flags, e->d_tele_sv, bp, s, qs));
        {
            BlitList z;   // as on the fp16 path: zeroed seed rows, row map and column bitmap from their static parts
            z.zero(e->d_tele_sv + (size_t)e->n_passages * bp, (int64_t)batch * kMaxSeeds * bp * sizeof(float));
            z.copy(e->d_row_slot, e->d_row_to_tele, (int64_t)e->V * sizeof(int32_t));
            z.copy(e->d_colmask, e->d_colmask_static, (int64_t)e->colmask_words * sizeof(uint32_t));
            HRAG_TRY(launch_blits(z, s));
        }
        HRAG_TRY(launch_ppr16_seed_rows(e->d_seed_vtx, e->d_seed_w, e->d_seed_cnt, qs, batch,
                                        e->n_passages, e->V, e->d_row_slot, e->d_tele_sv, 0, bp, s));
        HRAG_TRY(launch_ppr8_mask_seeds(e->d_seed_vtx, e->d_seed_cnt, batch, e->V, e->d_colmask, s));
    } else {
        HRAG_TRY(hrag_stage_teleport(e, e->d_spass, e->ld_p, e->d_mn_p, e->d_mx_p, passage_node_weight,
                                     e->d_flags, batch, e->d_tele, stream));
    }
    if (prof) HRAG_HIP_TRY(hipEventRecord(e->ev[EV_SEED], s));
    // PPR (HippoRAG.py:1736-1743): fixed-count leaky power iteration
    if (f8) {
        // `f8_iters` sweeps + the conditional steps of the convergence contract (their gate words decide on the device)
        // more than 256 queries: one launch per group of two slabs instead of one launch over all of them -- a launch then
        // gathers from ONE group's 256 MB of state (configs[2]'s working set, which the Infinity Cache still helps with)
        // instead of B / 256 times that: batch 1024 on configs[2]'s graph 69.8 -> 67.0 ms per call, 14.67 k -> 15.28 k queries/s
        // (profiles/r06w_bench_cfg4_groupwise_ab.json: the rate of four batches of 256); results
        // are bit-identical (the row shards have always issued their sweeps per group: tests/test_gpu_full_size.py)
        const bool per_group = e->p8.n_groups > 1 && p8_groupwise_enabled();
        for (int it = 0; it < e->p8.n_steps; ++it) {
            if (per_group)
                for (int g = 0; g < e->p8.n_groups; ++g) HRAG_TRY(ppr8_sweep(e, it, g, nullptr, s));
            else
                HRAG_TRY(ppr8_sweep(e, it, -1, nullptr, s));
            HRAG_TRY(ppr8_decide(e, it, s));
        }
    } else if (f16) {
        HRAG_TRY(ppr16_run(e, batch, damping, ppr_iters, s, est, &sweeps_run, ppr_tol == 0.f, ppr_tol, ppr_max_iters, &ext_max));
    } else if (sv) {
        HRAG_TRY(ppr_sv_run(e, e->d_row_slot, e->d_tele_sv, bp, damping, ppr_iters, s, est, batch, &sweeps_run, ppr_tol == 0.f,
                            ppr_tol, ppr_max_iters, &ext_max));
    } else {
        float *x = e->d_x, *y = e->d_y;
        HRAG_TRY(ppr_init(e, e->d_tele, e->n_passages, e->d_row_to_tele, e->d_seed_vtx, e->d_seed_w,
                          e->d_seed_cnt, batch, x, lay, s));
        for (int it = 0; it < ppr_iters; ++it) {
            HRAG_TRY(ppr_step(e, e->d_tele, e->n_passages, e->d_row_to_tele, e->d_seed_vtx, e->d_seed_w,
                              e->d_seed_cnt, batch, damping, x, y, lay, false, s));
            std::swap(x, y);
        }
        if (x != e->d_x) std::swap(e->d_x, e->d_y);  // keep the final state in d_x (d_y: the sweep before)
        if (est) HRAG_TRY(launch_passage_delta(e->d_x, e->d_y, e->V, e->d_passage_vertex, e->n_passages, batch, lay, est, s));
    }
    if (prof) HRAG_HIP_TRY(hipEventRecord(e->ev[EV_PPR], s));
    // doc scores + ranking (HippoRAG.py:1745-1747, :503)
    if (sv) {
        // normalisation: the closed-form mass of the K-sweep iterate (the last sweep only produced the passage rows)
        HRAG_TRY(launch_ppr_sv_mass(e->d_tele_sv, bp, 0, e->n_passages, e->n_passages + (int64_t)batch * kMaxSeeds,
                                    e->d_piso, e->d_iso, e->d_row_to_tele, e->d_seed_vtx, e->d_seed_w, e->d_seed_cnt,
                                    sv_half ? e->d_qscale : nullptr, e->V, batch, damping, ppr_iters,
                                    e->d_colsum_partial, e->d_sums, s));
        HRAG_TRY(launch_ppr_sv_rows(e->d_x, e->d_passage_vertex, e->n_passages, batch, e->d_sums, e->d_doc,
                                    e->ld_p, e->d_spass, e->ld_p, e->d_mn_p, e->d_mx_p, e->d_flags, bp, s));
    } else if (f8) {
        HRAG_TRY(ppr8_doc_scores(e, e->d_mn_p, e->d_mx_p, e->d_flags, batch, s, true));   // d_sums: the analytic mass
    } else if (f16) {
        // the same tail as the fp8 path: closed-form mass, x at the passages already in passage order
        HRAG_TRY(launch_ppr_sv_mass(e->d_tele16, 64, e->tele16_rows, e->n_passages,
                                    e->n_passages + (int64_t)batch * kMaxSeeds, e->d_piso, e->d_iso, e->d_row_to_tele,
                                    e->d_seed_vtx, e->d_seed_w, e->d_seed_cnt, e->d_qscale, e->V, batch, damping,
                                    ppr_iters, e->d_colsum_partial, e->d_sums, s));
        HRAG_TRY(ppr8_doc_scores(e, e->d_mn_p, e->d_mx_p, e->d_flags, batch, s, false));
    } else {
        HRAG_TRY(launch_colsum(e->d_x, e->V, 0, e->V, batch, lay, e->d_colsum_partial, e->d_sums, s));
        HRAG_TRY(launch_slab_to_rows(e->d_x, e->V, e->d_passage_vertex, e->n_passages, batch, e->d_sums,
                                     e->d_doc, e->ld_p, e->d_spass, e->ld_p, e->d_mn_p, e->d_mx_p, e->d_flags,
                                     lay, s));
    }
    HRAG_TRY(launch_flag_zero_mass(e->d_sums, batch, e->d_flags, 2, s));
    HRAG_TRY(launch_row_topk(e->d_doc, batch, e->n_passages, e->ld_p, k, 0, kNormNone, doc_idx_out,
                             doc_score_out, nullptr, nullptr, s, e->d_topk_ws, kTopkWsBytes));
    const bool measured = f8 || est != nullptr;
    if (!f8 && est)
        HRAG_TRY(launch_ppr8_finalize(est, e->d_flags, batch, damping / (1.0f - damping), ppr_tol, sweeps_run, e->d_ctl,
                                      ext_max, nullptr, 0, e->d_sums, e->d_resid, e->d_iters_used, s));
    {
        BlitList out;   // the per-query results in one launch
        out.copy(flags_out, e->d_flags, (int64_t)batch * sizeof(int32_t));
        if (measured) {
            out.copy(residual_out, e->d_resid, (int64_t)batch * sizeof(float));
            out.copy(iters_out, e->d_iters_used, (int64_t)batch * sizeof(int32_t));
        }
        HRAG_TRY(launch_blits(out, s));
    }
    if (!measured) {   // no sweep measured anything (ppr_iters == 0, or nothing asked for): -1 / the count as given
        if (residual_out) HRAG_TRY(launch_fill_i32(reinterpret_cast<int32_t *>(residual_out), (int32_t)0xbf800000u, batch, s));
        if (iters_out) HRAG_TRY(launch_fill_i32(iters_out, sweeps_run, batch, s));
    }
    if (prof) {
        HRAG_HIP_TRY(hipEventRecord(e->ev[EV_RANK], s));
        e->have_retrieve_ev = true;
    }
    e->last.ppr_iters = ppr_iters;
    e->last.n_slabs = lay.n_slabs;
    e->last.slab_width = f8 ? 128 : lay.bc;
    e->last_ppr_state = f8 ? HRAG_PPR_STATE_FP8 : f16 ? HRAG_PPR_STATE_F16 : sv ? HRAG_PPR_STATE_SMALL : HRAG_PPR_STATE_F32;
    e->counters.retrieve.fetch_add(1);
    e->counters.queries.fetch_add(batch);
    return HRAG_OK;
}

hrag_status hrag_retrieve(hrag_engine *e, const uint16_t *q_pass, int32_t batch,
                          const int32_t *kept_idx, const float *kept_score, const int32_t *kept_count,
                          int32_t kf, int32_t link_top_k, float damping, float passage_node_weight,
                          int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol, int32_t k, int32_t *doc_idx_out,
                          float *doc_score_out, int32_t *flags_out, float *residual_out, int32_t *iters_out,
                          hrag_stream stream) {
    HRAG_REQUIRE(q_pass != nullptr, "NULL argument");
    return retrieve_impl(e, q_pass, nullptr, 0, batch, kept_idx, kept_score, kept_count, kf, link_top_k, damping,
                         passage_node_weight, ppr_iters, ppr_max_iters, ppr_tol, k, doc_idx_out, doc_score_out,
                         flags_out, residual_out, iters_out, stream);
}

hrag_status hrag_retrieve_scored(hrag_engine *e, const float *pass_scores, int64_t pass_ld, int32_t batch,
                                 const int32_t *kept_idx, const float *kept_score, const int32_t *kept_count,
                                 int32_t kf, int32_t link_top_k, float damping, float passage_node_weight,
                                 int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol, int32_t k,
                                 int32_t *doc_idx_out, float *doc_score_out, int32_t *flags_out, float *residual_out,
                                 int32_t *iters_out, hrag_stream stream) {
    HRAG_REQUIRE(pass_scores != nullptr, "NULL argument");
    return retrieve_impl(e, nullptr, pass_scores, pass_ld, batch, kept_idx, kept_score, kept_count, kf, link_top_k,
                         damping, passage_node_weight, ppr_iters, ppr_max_iters, ppr_tol, k, doc_idx_out, doc_score_out,
                         flags_out, residual_out, iters_out, stream);
}

hrag_status hrag_last_doc_scores(hrag_engine *e, int32_t batch, float *out, int64_t ld, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(out && ld >= e->p_rows, "bad argument");
    HRAG_REQUIRE(e->d_doc != nullptr, "engine has no score rows");
    HRAG_ENGINE_CALL(e, stream);
    HRAG_HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld * sizeof(float), e->d_doc, (size_t)e->ld_p * sizeof(float),
                                  (size_t)e->p_rows * sizeof(float), (size_t)batch, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return HRAG_OK;
}

hrag_status hrag_dense_retrieve(hrag_engine *e, const uint16_t *q_pass, int32_t batch, int32_t k,
                                int32_t *doc_idx_out, float *doc_score_out, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(q_pass && doc_idx_out && doc_score_out, "NULL argument");
    HRAG_REQUIRE(e->p_rows == e->n_passages && e->d_pemb, "hrag_dense_retrieve needs the whole passage matrix");
    HRAG_REQUIRE(k >= 1 && k <= e->max_topk, "k=%d outside [1, max_topk=%d]", k, e->max_topk);
    hipStream_t s = (hipStream_t)stream;
    HRAG_ENGINE_CALL(e, stream);
    e->counters.dense.fetch_add(1);
    e->counters.queries.fetch_add(batch);
    HRAG_TRY(prep_query(e, q_pass, batch, s, &q_pass));
    HRAG_TRY(launch_sim_gemm(e->d_pemb, e->p_rows, e->kdim, q_pass, batch, e->d_spass, e->ld_p, s, 0, e->emb_dtype));
    return launch_row_topk(e->d_spass, batch, e->n_passages, e->ld_p, k, 0, kNormMinMax, doc_idx_out,
                           doc_score_out, nullptr, nullptr, s, e->d_topk_ws, kTopkWsBytes);
}

hrag_status hrag_ppr(hrag_engine *e, const float *reset, int32_t batch, float damping, int32_t iters,
                     float *x_out, int32_t *flags_out, hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(reset && x_out && iters >= 0, "bad argument");
    HRAG_REQUIRE(damping >= 0.f && damping < 1.f, "damping %g outside [0, 1)", (double)damping);
    HRAG_REQUIRE(e->n_rows == e->V, "hrag_ppr needs an unsharded engine");
    hipStream_t s = (hipStream_t)stream;
    HRAG_ENGINE_CALL(e, stream);
    e->counters.ppr.fetch_add(1);
    if (!e->d_tele_dense) HRAG_TRY(dev_alloc(&e->d_tele_dense, e->state_elems));  // first use only
    if (use_sv(e, batch)) {   // run_ppr seam at B = 1 (ppr_sv.hip), v dense over all vertices
        const int bp = sv_width(batch);
        HRAG_TRY(launch_ppr_sv_reset(reset, e->V, batch, e->d_tele_dense, bp, s));
        HRAG_TRY(ppr_sv_run_full(e, nullptr, e->d_tele_dense, bp, damping, iters, s));
        HRAG_TRY(launch_ppr_sv_colsum(e->d_x, e->V, bp, e->d_colsum_partial, e->d_sums, s));
        HRAG_TRY(launch_ppr_sv_rows(e->d_x, nullptr, e->V, batch, e->d_sums, x_out, e->V, nullptr, 0, nullptr,
                                    nullptr, nullptr, bp, s));
        if (flags_out) {
            HRAG_HIP_TRY(hipMemsetAsync(flags_out, 0, (size_t)batch * sizeof(int32_t), s));
            HRAG_TRY(launch_flag_zero_mass(e->d_sums, batch, flags_out, 2, s));
        }
        return HRAG_OK;
    }
    const SlabLayout lay = e->layout(batch);
    HRAG_TRY(launch_rows_to_slab(reset, e->V, e->V, batch, kSanitize, nullptr, nullptr, 1.f, nullptr,
                                 e->d_tele_dense, lay, s));
    float *x = e->d_x, *y = e->d_y;
    HRAG_TRY(ppr_init(e, e->d_tele_dense, e->V, nullptr, nullptr, nullptr, nullptr, batch, x, lay, s));
    for (int it = 0; it < iters; ++it) {
        HRAG_TRY(ppr_step(e, e->d_tele_dense, e->V, nullptr, nullptr, nullptr, nullptr, batch, damping, x,
                          y, lay, false, s));
        std::swap(x, y);
    }
    if (x != e->d_x) std::swap(e->d_x, e->d_y);
    HRAG_TRY(launch_colsum(e->d_x, e->V, 0, e->V, batch, lay, e->d_colsum_partial, e->d_sums, s));
    HRAG_TRY(launch_slab_to_rows(e->d_x, e->V, nullptr, e->V, batch, e->d_sums, x_out, e->V, nullptr, 0,
                                 nullptr, nullptr, nullptr, lay, s));
    if (flags_out) {
        HRAG_HIP_TRY(hipMemsetAsync(flags_out, 0, (size_t)batch * sizeof(int32_t), s));
        HRAG_TRY(launch_flag_zero_mass(e->d_sums, batch, flags_out, 2, s));
    }
    return HRAG_OK;
}

hrag_status hrag_ppr_sweeps(hrag_engine *e, int32_t batch, int32_t n, float damping, int32_t flags,
                            hrag_stream stream) {
    HRAG_TRY(check_batch(e, batch));
    HRAG_REQUIRE(n >= 0, "n must be >= 0");
    if (flags & 4) {
        HRAG_REQUIRE(use_sv(e, batch), "small-batch kernels need an unsharded engine and batch <= 8");
        const int bp = sv_width(batch);
        float *x = e->d_x, *y = e->d_y;
        for (int it = 0; it < n; ++it) {
            HRAG_TRY(launch_ppr_sv_sweep(ppr_sv_args(e, e->sell, x, y, e->d_row_slot, e->d_tele_sv, damping), bp,
                                         (flags & 1) != 0, (hipStream_t)stream));
            std::swap(x, y);
        }
        if (x != e->d_x) std::swap(e->d_x, e->d_y);
        return HRAG_OK;
    }
    if (flags & 8) {   // fp8 sweeps over the state the last hrag_retrieve left; flags bits 4..5 pick the mode
        HRAG_REQUIRE(e->f8_ready && e->d_pool8[0] && e->p8.active && e->p8.batch == batch,
                     "no fp8 PPR state for batch %d (needs col_sum, max_batch > 64 and a preceding hrag_retrieve)", batch);
        const int mode = (flags >> 4) & 3;   // 0 = C, 1 = B, 2 = F, 3 = B0 (Ppr8Mode)
        const int rio = (flags >> 6) & 3;    // residual form of B / F (Ppr8Args.rio)
        for (int it = 0; it < n; ++it) {
            if (flags & 256) HRAG_TRY(ppr8_bench_gather_replay(e, it, (hipStream_t)stream));   // the gathers alone
            else HRAG_TRY(ppr8_bench_sweep(e, mode, rio, it, (flags & 1) != 0, (hipStream_t)stream));
        }
        return HRAG_OK;
    }
    if (flags & 2) {
        HRAG_REQUIRE(e->f16_ready && batch <= e->f16_max_batch, "engine has no fp16 PPR state for batch %d", batch);
        const int nt = (e->opt_flags & HRAG_OPT_TEMPORAL16) ? 0 : 3;
        uint16_t *h = e->d_h16[0], *hn = e->d_h16[1];
        for (int it = 0; it < n; ++it) {
            HRAG_TRY(launch_ppr16_sweep(ppr16_args(e, h, hn, nullptr, damping), kPprModeH, n_slabs64(batch), nt,
                                        (flags & 1) != 0, (hipStream_t)stream));
            std::swap(h, hn);
        }
        if (h != e->d_h16[0]) std::swap(e->d_h16[0], e->d_h16[1]);
        return HRAG_OK;
    }
    HRAG_REQUIRE(e->d_x && e->d_tele, "the fp32-state sweep hook needs an unsharded engine");
    const SlabLayout lay = e->layout(batch);
    float *x = e->d_x, *y = e->d_y;
    for (int it = 0; it < n; ++it) {
        HRAG_TRY(ppr_step(e, e->d_tele, e->n_passages, e->d_row_to_tele, e->d_seed_vtx, e->d_seed_w,
                          e->d_seed_cnt, batch, damping, x, y, lay, (flags & 1) != 0, (hipStream_t)stream));
        std::swap(x, y);
    }
    if (x != e->d_x) std::swap(e->d_x, e->d_y);
    return HRAG_OK;
}

}  // extern "C"
