// Internal: the engine object behind the opaque hrag_engine handle of include/hrag.h, shared by
// engine.hip (creation, single-GPU entry points) and shard.hip (staged fp8 PPR driver + row-shard ABI).
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "common.h"

namespace hrag {

constexpr int kSvMaxBatch = 8;         // batches up to this take the small-batch kernels (ppr_sv.hip)
constexpr float kPpr16CScale = 64.f;  // correction / residual are stored as f16(c * 64); see ppr16.hip

enum EvId { EV_START = 0, EV_SIM, EV_SEED, EV_PPR, EV_RANK, EV_FACT0, EV_FACT1, EV_COUNT };

// where dev_alloc books the bytes it hands out (hrag_engine_stats: index vs workspace), per creating thread
inline thread_local int64_t *tl_alloc_bytes = nullptr;

template <typename T>
inline hrag_status dev_alloc(T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) return HRAG_OK;
    hipError_t err = hipMalloc(reinterpret_cast<void **>(p), (size_t)count * sizeof(T));
    if (err != hipSuccess) {
        set_error("hipMalloc of %lld bytes failed: %s", (long long)(count * (int64_t)sizeof(T)),
                  hipGetErrorString(err));
        *p = nullptr;
        return HRAG_ENOMEM;
    }
    if (tl_alloc_bytes) *tl_alloc_bytes += count * (int64_t)sizeof(T);
    return HRAG_OK;
}

template <typename T>
inline hrag_status dev_upload(T **p, const T *src, int64_t count) {
    HRAG_TRY(dev_alloc(p, count));
    if (count > 0) HRAG_HIP_TRY(hipMemcpy(*p, src, (size_t)count * sizeof(T), hipMemcpyDefault));
    return HRAG_OK;
}

inline int auto_slab_width(int batch, int cap) {
    int bc = 4;
    while (bc < batch && bc < cap) bc <<= 1;
    return bc;
}
inline int n_slabs64(int batch) { return (int)ceil_div(batch, 64); }
inline int n_slabs128(int batch) { return (int)ceil_div(batch, 128); }

// SELL-8 matrix on the device (ppr16.hip header): the structure arrays + up to two value variants
struct Sell8Store {
    int2 *pairs = nullptr;      // (col, P value)                      -- ppr16 / ppr_sv
    int2 *pairs_at = nullptr;   // (col, At value = p_ij d_j / d_i)    -- ppr8
    int2 *chunk_meta = nullptr;
    int32_t *vrow = nullptr, *lrow_row = nullptr, *lrow_first = nullptr, *lrow_cnt = nullptr;
    int32_t *seg_lrow = nullptr;   // [n_partial] long row of a partial slot
    int32_t *lcount = nullptr;     // [n_slabs64(max_batch)][n_lrow] arrival counters, zero between launches
    int32_t *pslot = nullptr;      // [n_chunks] dense number of a chunk that holds a passage row, else -1
    int32_t n_chunks = 0, n_lrow = 0, n_partial = 0, n_pchunks = 0;
    int64_t steps = 0;
    uint32_t pairs_bytes() const { return (uint32_t)((steps + 4) * 512); }
    Sell8Dev dev_at() const {
        Sell8Dev d;
        d.pairs = pairs_at; d.pairs_bytes = pairs_bytes(); d.chunk_meta = chunk_meta; d.vrow = vrow;
        d.n_chunks = n_chunks; d.lrow_row = lrow_row; d.lrow_first = lrow_first; d.lrow_cnt = lrow_cnt;
        d.n_lrow = n_lrow; d.n_partial = n_partial; d.seg_lrow = seg_lrow; d.lcount = lcount;
        d.pslot = pslot; d.n_pchunks = n_pchunks;
        return d;
    }
};

// One step of the staged fp8 iteration (csrc/ppr8.hip): which kernel mode, on which of the three state
// buffers, with which static scales.
struct Ppr8Step {
    int32_t mode;        // Ppr8Mode
    int32_t stage;       // stage the step belongs to (B / B0 / F: the stage being closed)
    int32_t x, y, rt;    // state buffer indices (y: written = to be exchanged, -1 for mode F; rt: the stage's rhs)
    float inv_cs, cs_next;
    int32_t rio;         // residual form of a boundary / final step (Ppr8Args.rio)
    float c_mul = 0.f, r_mul = 1.f;   // mode C: multipliers of the gathered sum and of the right-hand side
    // convergence contract: gate = index of the control word the launch is conditional on (-1: always runs);
    // decide = j >= 0: this final sweep (variant j) is followed by decision number j (ppr8_decide_kernel: its measured
    // update against the tolerance), -1: none
    int32_t gate = -1, gate_want = 1;
    int32_t decide = -1;
    // HRAG_OPT_ACCEL, boundary steps: kappa_growth = (max-norm contraction of the NEXT stage) x (growth of the iterate of
    // the stage after it): what ppr8_next_scale_kernel multiplies the measured maximum with (0: last boundary)
    float kappa_growth = 0.f;
};
struct Ppr8Session {
    bool active = false;
    int32_t batch = 0, iters = 0, n_steps = 0, n_stage = 0;   // iters: the sweeps of the base plan (accelerated: fewer than asked)
    bool accel = false;
    bool dyn = false;                    // stage scales measured on the device (accelerated plans; plain plans at small damping)
    int32_t n_slabs = 0, n_groups = 0, spg = 0;
    int64_t group_bytes = 0;
    float damping = 0.f;
    uint8_t *buf[3] = {nullptr, nullptr, nullptr};
    const int32_t *flags = nullptr;      // the batch's flag words (bit 3: fp8 saturation)
    Ppr8Step steps[64];
    float stage_inv[kP8MaxStages];
    // convergence contract (ppr8_begin): sweeps always run / extension stages allowed / tolerance on est
    int32_t e_max = 0;
    bool want_est = false;
    float tol = 0.f;
    // per-group issue of a session with measured scales (dyn): the boundary of a step finalises its maximum when the LAST
    // group arrives, so the groups of a step must come in ascending order and a step must be complete before the next
    // begins -- ppr8_sweep enforces it (a stale scale would otherwise only show as a saturation flag)
    int32_t last_step = -1, last_group = -1;
};

}  // namespace hrag

using namespace hrag;

// host-side entry flag of an engine handle; copying a handle (hrag_workspace_create) starts from "not in a call"
struct CallFlag {
    std::atomic<int> v{0};
    CallFlag() = default;
    CallFlag(const CallFlag &) : v(0) {}
    CallFlag &operator=(const CallFlag &) { v.store(0); return *this; }
};
// cumulative counters of one engine handle (hrag_engine_stats)
struct CallCounters {
    std::atomic<long long> score_facts{0}, retrieve{0}, dense{0}, ppr{0}, shard{0}, queries{0};
    CallCounters() = default;
    CallCounters(const CallCounters &) {}
    CallCounters &operator=(const CallCounters &) { return *this; }
};

struct hrag_engine {
    int device = 0;
    // graph (owned rows)
    int64_t V = 0, row_offset = 0, n_rows = 0, nnz = 0, n_passages = 0;
    int32_t *d_row_ptr = nullptr, *d_col = nullptr;
    float *d_val = nullptr;
    int32_t *d_row_order = nullptr;
    int32_t n_short = 0;
    int32_t *d_seg_row = nullptr, *d_seg_begin = nullptr, *d_seg_end = nullptr, *d_seg_slot = nullptr;
    int32_t *d_mrow_row = nullptr, *d_mrow_first = nullptr, *d_mrow_cnt = nullptr;
    int32_t n_seg = 0, n_mrow = 0, n_partial = 0, n_long_rows = 0;
    float *d_partial = nullptr;
    int32_t *d_passage_vertex = nullptr;  // [Np] global vertex ids
    int32_t *d_row_to_tele = nullptr;     // [n_rows] global passage index of an owned row, or -1
    // embeddings (owned rows)
    int32_t dim = 0, emb_dtype = HRAG_BF16;   // emb_dtype: what the similarity kernels see (fp16 on a split engine)
    int32_t kdim = 0;              // elements per stored row / per query as the kernels see them: dim, or 3 * dim (split)
    bool split = false;            // HRAG_F32_SPLIT: rows [hi | lo | hi], queries arrive as fp32 and become [hi | hi | lo]
    uint16_t *d_qsplit = nullptr;  // [max_batch][3 * dim] the current call's queries in that layout
    int64_t p_rows = 0, p_offset = 0, f_rows = 0, f_offset = 0, n_facts = 0;
    uint16_t *d_pemb = nullptr, *d_femb = nullptr;
    int32_t *d_subj = nullptr, *d_obj = nullptr, *d_num_chunks = nullptr;
    // options
    int32_t max_batch = 0, max_topk = 0, slab_cap = 32, short_thresh = 0, seg_len = 0, opt_flags = 0, sell_seg_len = 0, sell_sigma = 0;
    // workspace
    int64_t state_elems = 0;  // floats in each of d_x / d_y
    float *d_x = nullptr, *d_y = nullptr, *d_tele = nullptr, *d_tele_dense = nullptr;
    int64_t ld_p = 0, ld_f = 0;
    float *d_spass = nullptr, *d_sfact = nullptr, *d_doc = nullptr;
    float *d_mn_p = nullptr, *d_mx_p = nullptr;
    int32_t *d_seed_vtx = nullptr, *d_seed_cnt = nullptr, *d_flags = nullptr;
    float *d_seed_w = nullptr;
    double *d_colsum_partial = nullptr, *d_sums = nullptr;
    void *d_topk_ws = nullptr;   // kTopkWsBytes: lets small batches split a row over several workgroups
    // fused fact top-k (sim_gemm.hip): tile max / min, selected tiles, global min / max per query
    float *d_fused_ws = nullptr, *d_mn_f = nullptr, *d_mx_f = nullptr;
    int32_t *d_fused_sel = nullptr;
    // two-stage fp16 PPR (ppr16.hip): SELL-8 matrix + fp16 state, unsharded engines with max_batch > 8
    bool f16_ready = false;   // fp16 state buffers present (max_batch > 8)
    int32_t f16_max_batch = 0;  // ... sized for this many queries (64 when the fp8 path serves the larger batches)
    bool sell_ready = false;  // SELL-8 matrix + small-batch buffers present (every unsharded engine)
    float *d_tele_sv = nullptr, *d_partial_sv = nullptr;   // small-batch path (ppr_sv.hip), BP <= 8
    uint16_t *d_sv16[4] = {nullptr, nullptr, nullptr, nullptr};   // ... its two-stage fp16 state [V][8]: h ping / pong, r, c
    float *d_partial16 = nullptr;
    uint16_t *d_h16[4] = {nullptr, nullptr, nullptr, nullptr};  // hA, hB, r, cA
    int64_t state16_elems = 0;
    float *d_tele16 = nullptr;      // fp32 [n_slabs64][tele16_rows][64]: passages, then seed rows
    int64_t tele16_rows = 0;
    int32_t *d_row_slot = nullptr;  // [V] per-batch copy of d_row_to_tele with the seed rows patched in
    float *d_qscale = nullptr, *d_ssum = nullptr;
    // staged fp8 PPR (ppr8.hip): SELL-8 over the OWNED rows with the row-normalised values, fp32 residual on
    // the owned rows, three e4m3 state buffers; needs hrag_graph_desc.col_sum.  Serves hrag_retrieve for
    // batches > 64 on an unsharded engine and the hrag_shard_* entry points on a row shard.
    bool f8_ready = false;
    bool shard_aligned = false;   // owned passages == passages whose vertex is an owned row
    Sell8Store sell;              // owned rows (unsharded engines: also the P-valued pairs of ppr16 / ppr_sv)
    Sell8Store fsell;             // the owned PASSAGE rows only: the last sweep (mode F)
    int32_t *d_row_ptele = nullptr;   // [n_rows] LOCAL passage number of an owned row (-1: not a passage)
    float *d_deg = nullptr, *d_pinvdeg = nullptr, *d_R8 = nullptr, *d_partial8 = nullptr;
    uint16_t *d_rho8 = nullptr;    // fp16 remainder of the residual in its 3-byte form, same shape as d_R8
    uint8_t *d_iso = nullptr, *d_piso = nullptr;   // [V] / [p_rows]: vertex (of the passage) has no edges
    uint32_t *d_colmask_static = nullptr, *d_colmask = nullptr;   // [ceil(V / 32)] passage columns (+ seeds)
    int64_t colmask_words = 0;
    uint8_t *d_stagep = nullptr;   // [kP8MaxStages][n_slabs][p_rows][128]: c of every stage at the owned passages
    float *d_xp8 = nullptr;        // mode F output: x at the owned passages [n_slabs64][p_rows][64]
    uint8_t *d_pool8[3] = {nullptr, nullptr, nullptr};   // engine-owned state buffers (max_batch > 64, unsharded)
    int64_t state8_bytes = 0;
    int32_t *d_zmax_bits = nullptr;
    float *d_zmax = nullptr;
    double *d_mass = nullptr, *d_prior_part = nullptr;
    Ppr8Session p8;
    // convergence contract of the PPR solve (every state type): est_f = max over the passages of the relative size of
    // the update the (latest) final sweep applied, as float bits; control words ctl[j] = 1: extension stage j + 1 runs
    // (the launches of that stage are gated on it); the results
    int32_t *d_est_f = nullptr, *d_ctl = nullptr, *d_iters_used = nullptr;
    float *d_resid = nullptr;
    float *d_est_ws = nullptr;      // per-wavefront maxima of a sweep that measures est: [slabs][chunks][queries per slab row]
    double *d_mass_tab = nullptr;   // [kP8MaxExt + 1][max_batch]: mass of the (iters + 3 j)-sweep iterate
    // HRAG_OPT_ACCEL: stage scales measured on the device (ppr8.hip): table [2 * kP8DynInv], one slot per (chunk, slab)
    // for a boundary's max |R cs'|, the atomic word of the long rows
    float *d_dyn = nullptr, *d_mmax_ws = nullptr;
    int32_t *d_mmax_word = nullptr;
    int64_t mmax_slots = 0;
    // one call in flight per engine (include/hrag.h): host-side entry flag + the end of the last call on its stream
    CallFlag in_call;
    CallCounters counters;
    // hrag_workspace_create: a WORKSPACE handle borrows the immutable index buffers of `parent` (graph, SELL-8 matrices,
    // embeddings, static tables) and owns only the per-call buffers; free_engine frees what the handle owns
    bool borrowed = false;
    hrag_engine *parent = nullptr;
    CallFlag n_workspaces;          // live workspaces of this (root) engine: hrag_engine_destroy refuses while > 0
    // what hrag_engine_create decided (alloc_workspace reads them again for a workspace handle)
    bool want_sell = false, want_f16 = false, has_facts = false;
    int32_t fp8_unavailable = 0;    // HRAG_FP8_UNAVAILABLE_* bits: why the engine has no e4m3 PPR state (0: it has one)
    int32_t last_ppr_state = 0;     // HRAG_PPR_STATE_* of the last retrieve on this handle
    int64_t index_bytes = 0, workspace_bytes = 0;   // device memory of the two halves (hrag_engine_stats)
    hipEvent_t ev_last = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    // timing
    hipEvent_t ev[EV_COUNT] = {};
    bool profiling = false, have_retrieve_ev = false, have_fact_ev = false;
    hrag_timings last = {};

    SlabLayout layout(int batch) const {
        SlabLayout l;
        l.bc = auto_slab_width(batch, slab_cap);
        l.n_slabs = (int)ceil_div(batch, l.bc);
        return l;
    }
    SpmmArgs spmm_args(const float *x, float *y, const float *tele, int64_t tele_rows,
                       const int32_t *row_to_tele, float damping) const {
        SpmmArgs a;
        a.row_ptr = d_row_ptr; a.col_idx = d_col; a.val = d_val;
        a.row_order = d_row_order; a.n_short = n_short;
        a.seg_row = d_seg_row; a.seg_begin = d_seg_begin; a.seg_end = d_seg_end; a.seg_slot = d_seg_slot;
        a.n_seg = n_seg; a.mrow_row = d_mrow_row; a.mrow_first = d_mrow_first; a.mrow_cnt = d_mrow_cnt;
        a.n_mrow = n_mrow; a.partial = d_partial; a.n_partial = n_partial;
        a.n_rows = n_rows; a.row_offset = row_offset; a.num_vertices = V;
        a.x = x; a.y = y; a.row_to_tele = row_to_tele; a.tele = tele; a.tele_rows = tele_rows;
        a.alpha = damping; a.beta = 1.0f - damping; a.flags = opt_flags;
        return a;
    }
};

namespace hrag {
// Guard of the single-call entry points that use the engine's workspace.  A second THREAD inside a call is rejected
// (HRAG_EBUSY: the host-side entry flag).  A call on ANOTHER STREAM than the previous one is ordered behind it on the
// device -- hipStreamWaitEvent on the event that marks the end of the previous call -- so the workspace is never used
// by two calls at once and a caller who already ordered the two streams (or did not) gets correct results without a
// host synchronisation.  The end of this call is recorded on its stream when it leaves.  During stream capture the
// event logic is skipped (a captured event cannot be waited on from outside the capture); KNOWN HOLE: replays of a
// captured graph do not pass through here, so a direct call on another stream racing a replay is not ordered -- replay
// on the stream the direct calls use, or synchronise between them (engine.CapturedPipeline documents the same).
struct EngineCall {
    hrag_engine *e;
    hipStream_t s;
    hrag_status st = HRAG_OK;
    bool entered = false, capturing = false;
    EngineCall(hrag_engine *eng, hipStream_t stream) : e(eng), s(stream) {
        if (!e) return;
        int expected = 0;
        if (!e->in_call.v.compare_exchange_strong(expected, 1)) {
            set_error("engine busy: another thread is inside a call on this engine (one call in flight per engine)");
            st = HRAG_EBUSY;
            return;
        }
        entered = true;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) capturing = true;
        if (!capturing && e->have_last && e->last_stream != s && e->ev_last) {
            const hipError_t err = hipStreamWaitEvent(s, e->ev_last, 0);
            if (err != hipSuccess) {
                set_error("hipStreamWaitEvent on the previous call's end event -> %s", hipGetErrorString(err));
                st = HRAG_EHIP;
            }
        }
    }
    ~EngineCall() {
        if (!entered) return;
        if (st == HRAG_OK && !capturing && e->ev_last) {
            if (hipEventRecord(e->ev_last, s) == hipSuccess) { e->last_stream = s; e->have_last = true; }
        }
        e->in_call.v.store(0);
    }
};
#define HRAG_ENGINE_CALL(e, stream)                 \
    EngineCall _engine_call((e), (hipStream_t)(stream)); \
    if (_engine_call.st != HRAG_OK) return _engine_call.st

// the query matrix as the similarity kernels take it: the caller's pointer, or (HRAG_F32_SPLIT) its fp32 rows split
// into [hi | hi | lo] in the engine's buffer (one call in flight per engine: include/hrag.h)
inline hrag_status prep_query(hrag_engine *e, const uint16_t *q, int32_t batch, hipStream_t s, const uint16_t **out) {
    *out = q;
    if (!e->split) return HRAG_OK;
    if (batch < 1 || batch > e->max_batch) {
        set_error("batch %d outside [1, max_batch=%d]", batch, e->max_batch);
        return HRAG_ECAPACITY;
    }
    HRAG_TRY(launch_split3(reinterpret_cast<const float *>(q), batch, e->dim, 1, e->d_qsplit, s));
    *out = e->d_qsplit;
    return HRAG_OK;
}
// ---- shared between engine.hip and shard.hip
// The fp8 path serves a batch when the truncation error of `iters` sweeps is below the parity bar
// (damping^iters <= 2^-18) and the stage plan fits.
bool ppr8_usable(const hrag_engine *e, int batch, int iters, float damping);
int ppr8_plan(int iters, float damping, bool measured, int *plan);
// HRAG_OPT_ACCEL: stage lengths 1, 3, 3, ... (+ a closing plain stage of 1 sweep when `measured`: the convergence measure
// then reads a plain sweep's update) standing for the accuracy of `iters` plain sweeps; kind[i] = 1 marks a stage whose
// sweeps are Chebyshev steps.  Returns the number of stages, 0 when the variant saves no sweep
int ppr8_plan_accel(int iters, float damping, bool measured, int *plan, int *kind);
double cheb_T(int m, double x);   // Chebyshev polynomial T_m(x), x >= 1
// layout of the state buffers for `batch` queries in `want_groups` exchange groups (0 = the narrowest groups: slab
// pairs; the group width is kept even, see hrag.h)
hrag_status ppr8_layout(const hrag_engine *e, int32_t batch, int32_t want_groups, hrag_shard_layout *out);
// local statistics of the passage prior over the owned passages (d_spass must hold the local scores)
hrag_status ppr8_prior(hrag_engine *e, const float *mn, const float *mx, float passage_weight,
                       const int32_t *flags, int32_t batch, float *zmax_out, double *mass_out, hipStream_t s);
// scale, teleport rows, seed rows, column mask, stage plan, c_0 on the owned rows of buf[0]
// max_iters / tol / want_est: the convergence contract (include/hrag.h, hrag_retrieve): tol > 0 lets the DEVICE add
// up to kP8MaxExt stages of 1, 2, 3, 3 sweeps; the session then has more steps than `iters` (p8.n_steps)
hrag_status ppr8_begin(hrag_engine *e, const float *mn, const float *mx, const float *zmax, const double *mass,
                       float passage_weight, const int32_t *seed_vtx, const float *seed_w,
                       const int32_t *seed_cnt, int32_t *flags, int32_t batch, float damping, int32_t iters,
                       const hrag_shard_layout &lay, uint8_t *const bufs[3], hipStream_t s,
                       int32_t max_iters = 0, float tol = 0.f, bool want_est = false, bool allow_accel = false);
// step `i` (0-based, < p8.n_steps) on exchange group `group` (-1: every group); *exchange = buffer written (-1: none)
hrag_status ppr8_sweep(hrag_engine *e, int32_t i, int32_t group, int32_t *exchange, hipStream_t s);
// after the checkpoint step `i` ran on every group: decide whether the next stage is the last (one tiny launch)
hrag_status ppr8_decide(hrag_engine *e, int32_t i, hipStream_t s);
// residual / sweeps used / NOT_CONVERGED flag / the mass of the iterate actually computed -> d_resid, d_iters_used,
// flags, d_sums (every session, after its last step)
hrag_status ppr8_finalize(hrag_engine *e, int32_t *flags, hipStream_t s);
// d_doc[q][p_local] = x / mass, or the normalised DPR score on the fallback; flags bit 1 on zero mass
// fp8_session: the scores come from the active fp8 session (ppr8_finalize runs first: mass, residual, flag)
hrag_status ppr8_doc_scores(hrag_engine *e, const float *mn, const float *mx, int32_t *flags, int32_t batch,
                            hipStream_t s, bool fp8_session);
// measurement hook: one launch of kernel mode `mode` over the buffers of the active session (it: parity of
// the ping-pong); results are garbage, the memory traffic is that of a real sweep of that mode
hrag_status ppr8_bench_sweep(hrag_engine *e, int mode, int rio, int it, bool main_only, hipStream_t s);
hrag_status ppr8_bench_gather_replay(hrag_engine *e, int it, hipStream_t s);
}  // namespace hrag
