// Host driver of the staged fp8 PPR (csrc/ppr8.hip) and the row-shard entry points of include/hrag.h.
//
// The reference runs one igraph/PRPACK solve per query on the host (src/hipporag/HippoRAG.py:459 loop,
// :1736-1743); there is no distributed code upstream.  Here a row shard (one GPU of a node, or the
// single GPU that owns every row) iterates its rows of  z <- a At z + b v/d  on an e4m3 state that is
// replicated across the shards; between two sweeps the host exchanges the owners' row blocks
// (hipporag_amd/dist.py: one RCCL all-gather per exchange group, 1 byte per vertex and query).
// hrag_retrieve drives the same code with every exchange being a no-op.
#include "engine_impl.h"

namespace hrag {

// Stage lengths.  In exact arithmetic a stage of m sweeps leaves R_new = (aAt)^m R + (R - rt/cs): the residual contracts
// by a^m and the e4m3 rounding of the right-hand side (~2^-4.5 |R|) comes on top un-attenuated.  Every stage costs a
// boundary sweep (the fp32 / 3-byte residual in and out: 0.97 - 1.08 ms against 0.77 ms for a stage sweep at BASELINE
// configs[2]), so fewer, longer stages are cheaper -- as long as the rounding does not start to dominate a^m.
//   * round 1 - 4: 1 (the quantised start), 2, then 3-sweep stages, the remainder (1 or 2) last: 20 = 1+2+3+3+3+3+3+2,
//     six boundaries.  Kept for ppr_iters < 19 and for damping < 0.46 (a^4 is then below the rounding's share: longer
//     stages only waste sweeps there; at 0.3 the scales are measured, see ppr8_begin).
//   * round 5, damping >= 0.46 and ppr_iters >= 19, fixed count: 1, 2, 3, then 3-sweep stages, then as many 4-sweep stages as the
//     count allows, a 2-sweep stage last -- 20 = 1+2+3+4+4+4+2, FIVE boundaries.  The 4-sweep stages sit where the
//     residual already travels in its 3-byte form (their boundaries are the cheap ones); the last stage stays as short
//     as it was: its right-hand side is quantised at sweep K - 2, and that rounding is what the final sweep's measure
//     (the contract's residual) reads -- a first version ending on a 3-sweep stage (1+2+3+3+4+4+3) had the same TRUE
//     error but reported 2.7x the residual (cfg 3: 6.8e-6 against 2.5e-6; measured, profiles/r05a_*), which costs
//     extension sweeps under a tolerance.  CPU emulation of the device arithmetic over 45 candidate plans x {benchmark,
//     power-law, star forest, barbell} graphs (tools/exp_fp8_final.py, docs/experiments/README.md round 5): true error AND
//     measured residual within -10 % .. +40 % of the old plan at every count 19 .. 30 and damping 0.5 .. 0.6 (20 sweeps,
//     benchmark graph: 3.4e-7 / 4.8e-7 against 3.9e-7 / 4.9e-7); plans with TWO boundaries fewer (1,3,4,4,5,3) cost a
//     factor 2 in accuracy and 6x in the reported residual and are not taken.
//     On the device the fixed-count plan keeps the true error (cfg 3: the 12 oracle queries unchanged at 5.5e-7) but its
//     final sweep REPORTS 2.2x the round-1 plan's residual (5.5e-6 against 2.5e-6): under a tolerance that is an extension
//     stage for every batch.  So the rule looks at what the call asked for: a fixed count (ppr_tol = 0: only the launches
//     count) takes the plan above; `measured` (ppr_tol > 0: the final measure drives decisions) ends on 2 + 1 instead
//     (20 = 1+2+3+3+4+4+2+1: the round-1 plan's stage count, a lower final measure -- docs/experiments/README.md round 5).
// HRAG_P8_PLAN="1,2,4,4,4,4,1" (experiments only) overrides the rule when it sums to ppr_iters.
int ppr8_plan(int iters, float damping, bool measured, int *plan) {
    if (const char *env = experiment_env("HRAG_P8_PLAN")) {
        int n = 0, sum = 0;
        for (const char *c = env; *c && n < kP8MaxStages;) {
            const int m = atoi(c);
            if (m < 1) { n = 0; break; }
            plan[n++] = m; sum += m;
            while (*c && *c != ',') ++c;
            if (*c == ',') ++c;
        }
        if (n >= 2 && sum == iters && plan[0] == 1) return n;
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "libhrag: HRAG_P8_PLAN=%s ignored for ppr_iters=%d (needs >= 2 stages, at most %d, starting with 1, "
                            "summing to ppr_iters): the default plan runs\n", env, iters, kP8MaxStages);
        }
    }
    int n = 0;
    plan[n++] = 1;
    plan[n++] = 2;
    if (iters < 19 || !(damping >= 0.46f)) {
        int left = iters - 3;
        while (left >= 3) { plan[n++] = 3; left -= 3; }
        if (left > 0) plan[n++] = left;
        return n;
    }
    // fixed count:       iters - 8 = 4 a + 3 b with the largest a: [1, 2, 3] + b x [3] + a x [4] + [2]
    // under a tolerance: iters - 9 = 4 a + 3 b:                    [1, 2, 3] + b x [3] + a x [4] + [2, 1] -- as many stages
    //   as the round-1 plan had (20 = 1+2+3+3+4+4+2+1), but the last right-hand side is quantised one sweep before the
    //   end: what the final sweep reports is lower (cfg 3: fewer batches need the 21st sweep, 14.65 k -> 15.1 k queries/s
    //   under the default tolerance; same emulated true error: 3.6e-7 against 3.9e-7)
    const int t = iters - (measured ? 9 : 8);
    int a = t / 4;
    while (a > 0 && (t - 4 * a) % 3 != 0) --a;
    const int b = (t - 4 * a) / 3;
    plan[n++] = 3;
    for (int i = 0; i < b; ++i) plan[n++] = 3;
    for (int i = 0; i < a; ++i) plan[n++] = 4;
    plan[n++] = 2;
    if (measured) plan[n++] = 1;
    return n;
}

// Accelerated stages (HRAG_OPT_ACCEL).  A stage solves (I - G) c = R', G = a At, from c_0 = 0, c_1 = R'.  On an
// undirected graph (HippoRAG's: is_directed_graph = False) At = D^-1 A is similar to a symmetric matrix: the spectrum of
// G is real, inside [-a, a], and the Chebyshev semi-iteration
//     c_{k+1} = w_{k+1} (G c_k + R' - c_{k-1}) + c_{k-1},   w_1 = 1, w_2 = 1 / (1 - a^2 / 2), w_{k+1} = 1 / (1 - a^2 w_k / 4)
// leaves 1 / T_m(1 / a) of the residual after m sweeps instead of a^m (a = 0.5: 1/26 instead of 1/8 after three).  With
// c_0 = 0 and c_1 = R' the first two steps need no history term:
//     c_2 = w_2 (G c_1 + R'),   c_3 = w_3 G c_2 + R'
// -- two scalars per stage sweep (Ppr8Args.c_mul / r_mul); every boundary still forms the TRUE residual with plain a, so
// the refinement stays exact.  The e4m3 rounding of the iterates puts a floor of ~1/14 under the contraction of a stage
// (measured, tools/exp_fp8_chebyshev.py), so three sweeps are the sweet spot: 1, 3, 3, ... with
// n3 = ceil((iters - 1) ln(1/a) / ln(1 / max(1/T_3(1/a), 1/14))) stages stands for `iters` plain sweeps (a = 0.5,
// iters = 20: 16 sweeps, 4 full boundaries instead of 6).
//   * Mass.  T_3 is odd, so the error polynomial of every stage vanishes at 0 like the plain one: the mass of the
//     result is M - a S from the first sweep on, whatever the polynomial (ppr8_scale_kernel's closed form holds).
//   * Scales.  A Chebyshev residual polynomial is NOT a contraction by 1/T_3 in the max norm: p_3(G) = (4 G^3 / a^3 -
//     3 G / a) / T_3(1/a) has max-norm <= 7 / T_3(1/a) (0.27 at a = 0.5, against the spectral 0.038).  A STATIC chain of
//     scales cannot serve both: built on the spectral factor it leaves the e4m3 range on ring / star / barbell graphs
//     (round 3), built on the max-norm bound the values sink ~2.5x per stage below where e4m3 resolves them (measured at
//     cfg 3: 1.8e-6 instead of 4.3e-7, 26 sweeps under the contract).  The accelerated plan therefore MEASURES: every
//     boundary reports the batch's max |R| (one float per wavefront + a 1-block reduction, ~5 us), and the scale of the
//     stage after next is the power of two that maps (max-norm contraction of the next stage) x (that maximum) x (growth
//     of the iterate) to half the range -- rigorous like the static chain, but re-anchored at every stage
//     (ppr8.hip ppr8_next_scale_kernel).  Measured: cfg 3 4.3e-7 = the plain plan's error, 16k-vertex graphs 2.2e-7 ..
//     2.4e-7 (plain: 2.2e-7), the adversarial suite without a single saturated value.
//   * Convergence measure.  The contract's measure is the size of the update a PLAIN sweep applies: after Chebyshev
//     stages the true residual is spread over the whole spectrum (equi-oscillation) and its mid-spectrum part, which the
//     next plain sweep annihilates, makes the measure read 60x the true error (cfg 3, round 3).  With `measured` the plan
//     therefore ends on a PLAIN stage (round 4: one sweep; round 5: two, see below -- the measure then reads what is
//     left after a plain sweep that follows a plain sweep, the quantity the a-posteriori bound a / (1 - a) |update| is
//     about), and the extension stages are plain.
double cheb_T(int m, double x) {                 // Chebyshev polynomial T_m(x), x >= 1
    double t0 = 1.0, t1 = x;
    if (m == 0) return 1.0;
    for (int k = 1; k < m; ++k) { const double t2 = 2.0 * x * t1 - t0; t0 = t1; t1 = t2; }
    return t1;
}
constexpr double kAccelFloor = 1.0 / 14.0;       // contraction floor of an e4m3 stage
constexpr double kP8StageNoise = 0.05;           // what a stage's e4m3 rounding puts back into the residual, per unit of iterate growth
constexpr double kP8StaticFloor = 0.09;          // per-stage contraction a STATIC chain may assume at small damping

int ppr8_plan_accel(int iters, float damping, bool measured, int *plan, int *kind) {
    const double al = (double)damping;
    if (!(al >= 0.2) || al >= 1.0 || iters < 8) return 0;   // small damping: the plain stages sit on the floor already
    const double k3 = std::max(1.0 / cheb_T(3, 1.0 / al), kAccelFloor);
    const int n3 = (int)std::ceil((double)(iters - 1) * std::log(1.0 / al) / std::log(1.0 / k3) - 1e-9);
    const int total = 1 + 3 * n3 + (measured ? 1 : 0);
    if (n3 < 2 || total >= iters || n3 + 1 + (measured ? 1 : 0) > kP8MaxStages) return 0;
    int n = 0;
    plan[n] = 1; kind[n++] = 0;
    // Under a tolerance (round 5): the same 3 n3 + 2 sweeps and the same number of boundaries, arranged as 1, 2 (plain),
    // 3-sweep Chebyshev stages x (n3 - 1), 2 (plain) instead of 1, 3 x n3, 1 (plain).  What the final sweep REPORTS depends
    // on the last stages (ppr8_plan): a plain 2-sweep stage right after the quantised start takes the large first residual
    // down before the Chebyshev stages see it, and two plain sweeps at the end leave less mid-spectrum residue for the
    // measure to read than one.  CPU emulation (benchmark / power-law / barbell graphs): the measure reads 2 - 5x lower at
    // the same true error (8e-8 against 4e-7, 3e-7 against 6e-7, 5e-7 against 1.4e-6); on the device at BASELINE configs[2]
    // the residual after 17 sweeps sat just above the default tolerance for most batches (18 sweeps), see profiles/r05*.
    // HRAG_P8_ACCEL_CLOSE=1 keeps the round-4 arrangement (A/B measurements).
    static const bool old_close = [] { const char *e = experiment_env("HRAG_P8_ACCEL_CLOSE"); return e && e[0] == '1'; }();
    if (measured && !old_close && n3 >= 3) {
        plan[n] = 2; kind[n++] = 0;
        for (int i = 0; i < n3 - 1; ++i) { plan[n] = 3; kind[n++] = 1; }
        plan[n] = 2; kind[n++] = 0;
        return n;
    }
    for (int i = 0; i < n3; ++i) { plan[n] = 3; kind[n++] = 1; }
    if (measured) { plan[n] = 1; kind[n++] = 0; }
    return n;
}

// The truncation error of K sweeps is ~ damping^K of the mass whatever the state type; the staged scheme
// multiplies it by up to ~6 on graphs whose spectrum makes the bound tight (bipartite hubs), so it takes a
// batch when damping^K <= 2^-20 (0.5: K >= 20; 0.3: the minimum of 16; 0.6: K >= 28; 0.7 would need 39 > 30
// sweeps: the fp16 / fp32 state serves).
bool ppr8_usable(const hrag_engine *e, int batch, int iters, float damping) {
    if (!e->f8_ready || (e->opt_flags & HRAG_OPT_NO_FP8) || batch < 1) return false;
    if (iters < 16 || iters > 30 || !(damping >= 0.f)) return false;
    return std::pow((double)damping, (double)iters) <= 1.0 / 1048576.0;
}

hrag_status ppr8_layout(const hrag_engine *e, int32_t batch, int32_t want_groups, hrag_shard_layout *out) {
    HRAG_REQUIRE(e->f8_ready, "engine has no fp8 PPR state (needs col_sum, V < 2^24, aligned passage shard)");
    const int ns = n_slabs128(batch);
    const int64_t row_bytes = (e->V + 1) * 128;
    const int spg_max = (int)std::max<int64_t>(1, std::min<int64_t>(ns, ((int64_t)1 << 32) / row_bytes));
    int spg = want_groups <= 0 ? 1 : (int)ceil_div(ns, std::min(want_groups, ns));
    spg = std::max(1, std::min(spg, spg_max));
    // the sweep kernels work on PAIRS of adjacent slabs (ppr8_pair_kernel): keep the group width even
    if (ns >= 2 && (spg & 1)) spg = spg + 1 <= spg_max ? spg + 1 : std::max(1, spg - 1);
    hrag_shard_layout l = {};
    l.n_slabs = ns;
    l.slabs_per_group = spg;
    l.n_groups = (int)ceil_div(ns, spg);
    l.group_bytes = row_bytes * spg;
    l.state_bytes = l.group_bytes * l.n_groups;
    l.own_offset = e->row_offset * (int64_t)spg * 128;
    l.own_bytes = e->n_rows * (int64_t)spg * 128;
    *out = l;
    return HRAG_OK;
}

namespace {

Ppr8Args base_args(const hrag_engine *e) {
    const Ppr8Session &p = e->p8;
    Ppr8Args a = {};
    a.m = e->sell.dev_at();
    a.partial = e->d_partial8;
    a.row_offset = e->row_offset; a.n_rows = e->n_rows;
    a.spg = p.spg; a.row_stride = (uint32_t)p.spg * 128u; a.group_bytes = p.group_bytes;
    a.R = e->d_R8; a.rho = e->d_rho8; a.rio = 0;
    a.alpha = p.damping; a.beta = 1.0f - p.damping;
    a.c_mul = p.damping; a.r_mul = 1.0f;
    a.tele = e->d_tele16; a.tele_rows = e->tele16_rows; a.n_slabs64 = n_slabs64(p.batch);
    a.row_slot = e->d_row_slot; a.deg = e->d_deg; a.p_rows = e->p_rows;
    a.colmask = e->d_colmask; a.colmask_bytes = (uint32_t)(e->colmask_words * 4); a.zero_row = (uint32_t)e->V;
    a.flags = const_cast<int32_t *>(p.flags); a.batch = p.batch;
    a.slab0 = 0; a.n_slabs = p.n_slabs;
    a.wps = (e->opt_flags & HRAG_OPT_SLABS_PER_WG_1) ? 1 : 4;
    a.cg_per_xcd = (e->opt_flags & HRAG_OPT_XCD_BLOCKED) ? 1 : 0;   // the launcher fills in the count
    return a;
}

uint8_t *stage_copy(const hrag_engine *e, int stage) {
    return e->d_stagep + (size_t)stage * (size_t)e->p8.n_slabs * (size_t)std::max<int64_t>(e->p_rows, 1) * 128;
}

}  // namespace

hrag_status ppr8_prior(hrag_engine *e, const float *mn, const float *mx, float passage_weight,
                       const int32_t *flags, int32_t batch, float *zmax_out, double *mass_out, hipStream_t s) {
    HRAG_REQUIRE(e->f8_ready, "engine has no fp8 PPR state");
    return launch_ppr8_prior(e->d_spass, e->ld_p, e->p_rows, mn, mx, passage_weight, e->d_pinvdeg, e->d_piso,
                             flags, batch, e->d_zmax_bits, e->d_prior_part, zmax_out, mass_out, s);
}

hrag_status ppr8_begin(hrag_engine *e, const float *mn, const float *mx, const float *zmax, const double *mass,
                       float passage_weight, const int32_t *seed_vtx, const float *seed_w,
                       const int32_t *seed_cnt, int32_t *flags, int32_t batch, float damping, int32_t iters,
                       const hrag_shard_layout &lay, uint8_t *const bufs[3], hipStream_t s, int32_t max_iters,
                       float tol, bool want_est, bool allow_accel) {
    HRAG_REQUIRE(ppr8_usable(e, batch, iters, damping),
                 "the fp8-state PPR does not serve ppr_iters=%d at damping %g (needs 16..30 sweeps and "
                 "damping^ppr_iters <= 2^-20) or the engine has no fp8 state", iters, (double)damping);
    HRAG_REQUIRE(bufs[0] && bufs[1] && bufs[2] && bufs[0] != bufs[1] && bufs[1] != bufs[2] && bufs[0] != bufs[2],
                 "three distinct state buffers are needed");
    HRAG_REQUIRE(tol >= 0.f, "ppr_tol must be >= 0");
    Ppr8Session &p = e->p8;
    p.active = false;
    p.batch = batch; p.iters = iters; p.damping = damping; p.flags = flags;
    p.n_slabs = lay.n_slabs; p.n_groups = lay.n_groups; p.spg = lay.slabs_per_group; p.group_bytes = lay.group_bytes;
    for (int i = 0; i < 3; ++i) p.buf[i] = bufs[i];

    // ---- stage plan with static power-of-two scales (damping >= 0.46; below: measured, see `dyn` further down): the max-norm of the true residual contracts by
    // `damping` per sweep (At is row-stochastic), |R_0| <= max(a, 1 - a) max(v/d) + the rounding of c_0; a stage
    // of m sweeps grows its iterate by at most (1 - a^m) / (1 - a).  cs = the power of two that maps that bound
    // to <= 224 (half the e4m3 range: the bound ignores rounding noise; a clamped value raises flags bit 3).
    int plan[kP8MaxStages + 4], kind[kP8MaxStages + 4] = {};
    // accelerated stages need the measured scale chain: the single-GPU engine only (a row shard would have to all-reduce
    // one float per boundary; the shard entry points run the plain plan)
    const bool may_accel = allow_accel && (e->opt_flags & HRAG_OPT_ACCEL) && e->d_dyn && e->n_rows == e->V;
    int n_stage = may_accel ? ppr8_plan_accel(iters, damping, tol > 0.f, plan, kind) : 0;
    const bool accel = n_stage > 0;
    if (!accel) n_stage = ppr8_plan(iters, damping, tol > 0.f, plan);
    if (accel) {                                   // the base plan's own sweep count is what the session runs and reports
        iters = 0;
        for (int i = 0; i < n_stage; ++i) iters += plan[i];
        p.iters = iters;
    }
    p.accel = accel;
    // Small damping: the static chain assumes that the max-norm of the residual contracts by a^m per stage, which ignores
    // what the e4m3 rounding of a stage puts back (~0.05 .. 0.1 of the residual per stage, whatever a).  At a >= 0.46
    // (a^3 >= 0.1) the half-range target of the scales absorbs that (worst emulated occupancy 0.63 of the range on a
    // sparse power-law graph at a = 0.6); below it the real residual outgrows the assumed one stage after stage and
    // valid inputs saturated (found by tools/soak_random.py at a = 0.3: ring, star forest, sparse power-law graphs;
    // emulation 3x .. 30x over the range).  There the plain plan MEASURES its scales like the accelerated one (one float per
    // wavefront per boundary + ppr8_next_scale_kernel): re-anchored at every stage, with the rounding inside the
    // contraction it multiplies with.  A row shard has no such maximum (it would be an all-reduce per boundary): its static
    // chain takes max(a^m, 0.09) per stage instead -- no saturation in the emulation, at the price of resolution on
    // slowly mixing graphs, which the convergence contract then extends.
    const bool small_al = damping < 0.46f;
    const bool dyn = accel || (small_al && e->d_dyn && e->n_rows == e->V);
    p.dyn = dyn;
    HRAG_REQUIRE(n_stage >= 2 && n_stage <= kP8MaxStages, "ppr_iters=%d needs %d fp8 stages (2..%d)", iters, n_stage,
                 kP8MaxStages);
    // ---- convergence contract (reference: PRPACK iterates until its residual is below 1e-10, HippoRAG.py:1736-1743;
    // here: `iters` sweeps always run, then the DEVICE may add stages of 1, 2, 3, 3 sweeps while the measured update of the
    // passage scores predicts an error above tol).  Every conditional launch is enqueued; its gate word decides.
    const bool est = want_est || tol > 0.f;
    int e_max = 0;
    if (tol > 0.f && max_iters > iters)
        while (e_max < std::min(kP8MaxExt, kP8MaxStages - n_stage) &&
               iters + p8_ext_sweeps(e_max + 1) <= std::min(max_iters, 30))
            ++e_max;
    for (int j = 0; j < e_max; ++j) plan[n_stage + j] = p8_ext_sweeps(j + 1) - p8_ext_sweeps(j);   // 1, 2, 3, 3
    p.e_max = e_max; p.want_est = est; p.tol = tol;
    const double al = (double)damping;
    const double w2 = 1.0 / (1.0 - al * al / 2.0), w3 = 1.0 / (1.0 - al * al * w2 / 4.0);   // Chebyshev weights
    double bound = std::max(al, 1.0 - al) + 0.07;
    // growth of a stage's iterate over its right-hand side; an accelerated 3-sweep stage: |c_2| <= w2 (a + 1),
    // |c_3| <= w3 a w2 (a + 1) + 1
    auto growth_of = [&](int si) {
        const int m = plan[si];
        double growth = al < 1.0 ? (1.0 - std::pow(al, m)) / (1.0 - al) : (double)m;
        if (kind[si]) growth = std::max(w2 * (al + 1.0), w3 * al * w2 * (al + 1.0) + 1.0);
        return std::max(growth, 1.0);
    };
    auto scale_for = [&](int si) {
        const double growth = growth_of(si);
        int ex = (int)std::floor(std::log2(224.0 / std::max(bound * std::max(growth, 1.0), 1e-18)));   // damping ~ 0: bound -> 0
        ex = std::min(std::max(ex, -60), 60);
        return std::ldexp(1.0f, ex);
    };
    // max-norm contraction of the true residual over stage si (the static scales rest on it) and the modelled one (when
    // the 3-byte residual form is precise enough): plain a^m for both; accelerated 7 / T_3(1/a) against 1 / T_3(1/a)
    auto norm_contraction = [&](int si) {
        if (kind[si]) return std::min(1.0, 7.0 / cheb_T(plan[si], 1.0 / al));
        const double am = std::pow(al, plan[si]);
        if (!small_al) return am;
        return dyn ? std::min(1.0, am + kP8StageNoise * growth_of(si)) : std::max(am, kP8StaticFloor);
    };
    auto model_contraction = [&](int si) {
        return kind[si] ? std::max(1.0 / cheb_T(plan[si], 1.0 / al), 1.0 / 16.0) : std::pow(al, plan[si]);
    };
    int n = 0, c = 0, rt = -1;   // c_0 lives in buffer 0
    double shrunk = 1.0;         // modelled size of the residual relative to the start (plain plan: damping^sweeps)
    bool r16 = false;            // the stored residual is in the 3-byte form (rt + fp16 remainder)
    float cs = kP8C0Scale, cs_next = scale_for(1);
    const int n_total = n_stage + e_max;
    for (int si = 0; si < n_total; ++si) {
        const int m = plan[si];
        const int ext = si - (n_stage - 1);            // >= 1: extension stage number `ext`, runs iff ctl[ext - 1]
        const int stage_gate = ext >= 1 ? ext - 1 : -1;
        if (si > 0) {
            cs = cs_next;
            c = rt;
            for (int j = 1; j < m; ++j) {
                const int dst = c == rt ? (c + 1) % 3 : 3 - c - rt;
                Ppr8Step st{kP8ModeC, si, c, dst, rt, 0.f, 0.f, 0};
                st.c_mul = damping; st.r_mul = 1.0f;
                if (kind[si]) {   // iterate j + 1 of an accelerated stage (m = 3: no history term, see ppr8_plan_accel)
                    st.c_mul = (float)((j == 1 ? w2 : w3) * al);
                    st.r_mul = j == 1 ? (float)w2 : 1.0f;
                }
                st.gate = stage_gate;
                p.steps[n++] = st;
                c = dst;
            }
            bound *= norm_contraction(si);
            cs_next = si + 1 < n_total ? scale_for(si + 1) : 1.0f;
        }
        p.stage_inv[si] = 1.0f / cs;
        shrunk *= si == 0 ? al : model_contraction(si);
        if (si >= n_stage - 1) {
            // the stage may be the last one: final sweep variant j (passage rows only) measures the update it applies;
            // decision j follows it (ppr8_decide_kernel): while that update is above the tolerance, the boundary below
            // closes the stage for real and extension stage j + 1 runs
            const int j = si - (n_stage - 1);
            Ppr8Step st{kP8ModeF, si, c, -1, rt, 1.0f / cs, 1.0f, r16 ? 1 : 0};
            st.gate = stage_gate;
            st.decide = (e_max > 0 && j < e_max) ? j : -1;
            p.steps[n++] = st;
        }
        if (si + 1 < n_total) {
            // the new right-hand side overwrites the old one in place (row by row: a row only ever reads its own
            // rt): R, or its fp16 remainder next to rt, carries everything else
            const int y = (rt >= 0 && c != rt) ? rt : (c + 1) % 3;
            // residual form: fp32 while it is large; (rt + fp16 remainder) once damping^k <= 2^-6, where the
            // 2^-15 relative error of that form is below 5e-7 of the solution (ppr8.hip finish_row)
            const bool out16 = si > 0 && shrunk <= 1.0 / 64.0;
            const int rio = (r16 ? 1 : 0) | (out16 ? 2 : 0);
            Ppr8Step st{si == 0 ? kP8ModeB0 : kP8ModeB, si, c, y, rt, 1.0f / cs, cs_next, rio};
            if (si >= n_stage - 1) st.gate = si - (n_stage - 1);   // closes a stage that could have been the last
            // accelerated plan: this boundary's measured maximum fixes the scale of stage si + 2 (if there is one)
            if (si + 2 < n_total) st.kappa_growth = (float)(norm_contraction(si + 1) * growth_of(si + 2));
            p.steps[n++] = st;
            r16 = out16;
            rt = y;
        }
    }
    HRAG_REQUIRE(n <= (int)(sizeof(p.steps) / sizeof(p.steps[0])), "internal: %d steps", n);
    p.n_steps = n;
    p.n_stage = n_stage;

    // ---- reset vector on the owned rows: per-query scale, passage prior rows, seed rows, column bitmap
    HRAG_TRY(launch_ppr8_scale(zmax, mass, passage_weight, seed_vtx, seed_w, seed_cnt, e->d_deg, e->d_iso, e->V,
                               flags, batch, damping, iters, e->d_qscale, e->d_mass_tab, s, e_max + 1, e->max_batch));
    SlabLayout l64;
    l64.bc = 64; l64.n_slabs = n_slabs64(batch);
    HRAG_TRY(launch_rows_to_slab(e->d_spass, e->ld_p, e->p_rows, batch, kMinMaxScale, mn, mx, passage_weight,
                                 flags, e->d_tele16, l64, s, e->tele16_rows, e->d_qscale));
    {
        // one launch for the session's small fills and copies: the contract's words, the seed rows of the teleport
        // matrix, the row -> teleport-row map and the column bitmap from their static parts
        BlitList z;
        if (est) z.zero(e->d_est_f, (int64_t)batch * sizeof(int32_t));
        if (dyn) z.zero(e->d_mmax_ws, e->mmax_slots * (int64_t)sizeof(float));   // slots a launch geometry never writes stay 0
        z.zero(e->d_ctl, (int64_t)(kP8MaxExt + 1) * sizeof(int32_t));
        z.zero(e->d_tele16 + (size_t)e->p_rows * 64, (int64_t)batch * kMaxSeeds * 64 * sizeof(float), l64.n_slabs,
               (int64_t)e->tele16_rows * 64 * sizeof(float));
        z.copy(e->d_row_slot, e->d_row_ptele, (int64_t)e->n_rows * sizeof(int32_t));
        z.copy(e->d_colmask, e->d_colmask_static, (int64_t)e->colmask_words * sizeof(uint32_t));
        HRAG_TRY(launch_blits(z, s));
    }
    HRAG_TRY(launch_ppr16_seed_rows(seed_vtx, seed_w, seed_cnt, e->d_qscale, batch, e->p_rows, e->V, e->d_row_slot,
                                    e->d_tele16, e->tele16_rows, 64, s, e->row_offset, e->n_rows));
    HRAG_TRY(launch_ppr8_mask_seeds(seed_vtx, seed_cnt, batch, e->V, e->d_colmask, s));
    if (dyn)   // the two scales known before anything is measured: c_0's and the first stage's
        HRAG_TRY(launch_ppr8_next_scale(nullptr, 0, e->d_mmax_word, e->d_dyn, 0, 0.f, 1, kP8C0Scale, p.steps[0].cs_next,
                                        nullptr, 0, s));
    // ---- c_0 = Q(v/d * 2^7) on the owned rows of every group
    Ppr8Args a = base_args(e);
    a.y = p.buf[0];
    HRAG_TRY(launch_ppr8_init(a, kP8C0Scale, s));
    p.active = true;
    p.last_step = p.last_group = -1;
    return HRAG_OK;
}

hrag_status ppr8_sweep(hrag_engine *e, int32_t i, int32_t group, int32_t *exchange, hipStream_t s) {
    const Ppr8Session &p = e->p8;
    HRAG_REQUIRE(p.active, "no fp8 PPR session: call the begin step first");
    HRAG_REQUIRE(i >= 0 && i < p.n_steps, "sweep %d outside [0, %d)", i, p.n_steps);
    HRAG_REQUIRE(group >= -1 && group < p.n_groups, "group %d outside [0, %d)", group, p.n_groups);
    if (p.dyn && group >= 0) {
        // measured stage scales + per-group issue: (step, group) must advance in lexicographic order without gaps
        const bool next_group = i == p.last_step && group == p.last_group + 1;
        const bool next_step = group == 0 && (p.last_step < 0 ? true : (i > p.last_step && p.last_group == p.n_groups - 1));
        HRAG_REQUIRE(next_group || next_step,
                     "hrag_shard_ppr_sweep(step %d, group %d) after (step %d, group %d): with measured stage scales the groups of a "
                     "step must be issued in ascending order and a step completed before the next begins (include/hrag.h)",
                     i, group, p.last_step, p.last_group);
        e->p8.last_step = i;
        e->p8.last_group = group;
    }
    const Ppr8Step &st = p.steps[i];
    Ppr8Args a = base_args(e);
    if (st.gate >= 0) { a.gate = e->d_ctl + st.gate; a.gate_want = st.gate_want; }
    if (group >= 0) {
        a.slab0 = group * p.spg;
        a.n_slabs = std::min(p.spg, p.n_slabs - a.slab0);
    }
    a.x = p.buf[st.x];
    a.inv_cs = st.inv_cs; a.cs_next = st.cs_next;
    const bool boundary = st.mode == kP8ModeB || st.mode == kP8ModeB0;
    if (p.dyn && st.mode != kP8ModeC) {     // measured stage scales (HRAG_OPT_ACCEL; plain plans at small damping)
        a.dyn = e->d_dyn; a.dyn_stage = st.stage;
        if (boundary) {
            a.mmax_ws = e->d_mmax_ws; a.mmax_atomic = e->d_mmax_word;
            a.mmax_units = a.n_slabs; a.mmax_slab0 = a.slab0;
        }
    }
    if (st.mode == kP8ModeC) {
        a.y = p.buf[st.y];
        a.rt = p.buf[st.rt];
        a.c_mul = st.c_mul; a.r_mul = st.r_mul;
    } else if (st.mode == kP8ModeB || st.mode == kP8ModeB0) {
        a.y = p.buf[st.y];
        a.rt = st.rt >= 0 ? p.buf[st.rt] : nullptr;
        a.rio = st.rio;
        a.stage_out = stage_copy(e, st.stage);
    } else {   // kP8ModeF: the passage rows only; stage st.stage is the last one
        a.rt = st.rt >= 0 ? p.buf[st.rt] : nullptr;
        a.rio = st.rio;
        a.m = e->fsell.dev_at();
        for (int k = 0; k < st.stage; ++k) a.stage[k] = stage_copy(e, k);
        for (int k = 0; k <= st.stage; ++k) a.stage_inv[k] = p.stage_inv[k];
        a.n_stage = st.stage + 1;
        a.out = e->d_xp8;
        if (p.want_est) { a.est = e->d_est_f; a.est_ws = e->d_est_ws; }
    }
    if (exchange) *exchange = st.y;
    HRAG_TRY(launch_ppr8_sweep(a, st.mode, false, s));
    // this boundary's maximum -> the scale of the stage after next.  A step that is launched per exchange group
    // (hrag_shard_ppr_sweep on an engine that owns every row) measures the maximum of the WHOLE batch: every group folds
    // its slots into the running maximum, the last group of the step (groups are issued in ascending order) finalizes.
    if (p.dyn && boundary && st.kappa_growth > 0.f)
        HRAG_TRY(launch_ppr8_next_scale(e->d_mmax_ws, (int32_t)std::min<int64_t>(e->mmax_slots, (int64_t)e->sell.n_chunks * a.mmax_units),
                                        e->d_mmax_word, e->d_dyn, st.stage, st.kappa_growth, 0, 0.f, 0.f, a.gate, a.gate_want, s,
                                        group < 0 || group == p.n_groups - 1));
    return HRAG_OK;
}

hrag_status ppr8_decide(hrag_engine *e, int32_t i, hipStream_t s) {
    const Ppr8Session &p = e->p8;
    HRAG_REQUIRE(p.active && i >= 0 && i < p.n_steps, "no such step");
    const Ppr8Step &st = p.steps[i];
    if (st.decide < 0) return HRAG_OK;
    const float g = p.damping / (1.0f - p.damping);
    return launch_ppr8_decide(e->d_est_f, p.flags, p.batch, g, p.tol, st.decide, p.e_max, e->d_ctl, s);
}

hrag_status ppr8_finalize(hrag_engine *e, int32_t *flags, hipStream_t s) {
    const Ppr8Session &p = e->p8;
    HRAG_REQUIRE(p.active, "no fp8 PPR session");
    const float g = p.damping / (1.0f - p.damping);
    return launch_ppr8_finalize(e->d_est_f, flags, p.batch, g, p.tol, p.iters, e->d_ctl, p.e_max, e->d_mass_tab,
                                e->max_batch, e->d_sums, e->d_resid, e->d_iters_used, s);
}

hrag_status ppr8_bench_sweep(hrag_engine *e, int mode, int rio, int it, bool main_only, hipStream_t s) {
    const Ppr8Session &p = e->p8;
    HRAG_REQUIRE(p.active, "no fp8 PPR session");
    Ppr8Args a = base_args(e);
    a.x = p.buf[it & 1];
    a.y = p.buf[(it & 1) ^ 1];
    a.rt = p.buf[2];
    a.inv_cs = 1.0f / 64.f; a.cs_next = 64.f;
    a.rio = (mode == kP8ModeB && (rio == 2 || rio == 3)) || (mode == kP8ModeF && rio == 1) ? rio : 0;
    a.stage_out = stage_copy(e, 0);
    if (mode == kP8ModeF) {
        a.m = e->fsell.dev_at();
        for (int k = 0; k + 1 < p.n_stage; ++k) a.stage[k] = stage_copy(e, k);
        for (int k = 0; k < p.n_stage; ++k) a.stage_inv[k] = p.stage_inv[k];
        a.n_stage = p.n_stage;
        a.out = e->d_xp8;
    }
    return launch_ppr8_sweep(a, mode, main_only, s);
}

// hrag_ppr_sweeps flag 256: the gathers of a stage sweep alone, on the session's matrix and state (measurement only)
hrag_status ppr8_bench_gather_replay(hrag_engine *e, int it, hipStream_t s) {
    const Ppr8Session &p = e->p8;
    HRAG_REQUIRE(p.active, "no fp8 PPR session");
    Ppr8Args a = base_args(e);
    a.x = p.buf[it & 1];
    return launch_ppr8_gather_replay(a, s);
}

hrag_status ppr8_doc_scores(hrag_engine *e, const float *mn, const float *mx, int32_t *flags, int32_t batch,
                            hipStream_t s, bool fp8_session) {
    SlabLayout l64;
    l64.bc = 64; l64.n_slabs = n_slabs64(batch);
    if (fp8_session) HRAG_TRY(ppr8_finalize(e, flags, s));   // d_sums <- the mass of the iterate that was computed
    HRAG_TRY(launch_slab_to_rows(e->d_xp8, e->p_rows, nullptr, e->p_rows, batch, e->d_sums, e->d_doc, e->ld_p,
                                 e->d_spass, e->ld_p, mn, mx, flags, l64, s));
    if (flags) HRAG_TRY(launch_flag_zero_mass(e->d_sums, batch, flags, 2, s));
    return HRAG_OK;
}

}  // namespace hrag

namespace {

hrag_status shard_batch(const hrag_engine *e, int32_t batch) {
    HRAG_REQUIRE(e != nullptr, "engine is NULL");
    if (batch < 1 || batch > e->max_batch) {
        set_error("batch %d outside [1, max_batch=%d]", batch, e->max_batch);
        return HRAG_ECAPACITY;
    }
    return HRAG_OK;
}

// an empty shard contributes the neutral elements of the min / max all-reduce
hrag_status fill_minmax_neutral(float *mn, float *mx, int32_t batch, hipStream_t s) {
    HRAG_TRY(launch_fill_i32(reinterpret_cast<int32_t *>(mn), 0x7f800000, batch, s));            // +inf
    HRAG_TRY(launch_fill_i32(reinterpret_cast<int32_t *>(mx), (int32_t)0xff800000u, batch, s));  // -inf
    return HRAG_OK;
}

}  // namespace

extern "C" {

hrag_status hrag_shard_layout_query(hrag_engine *e, int32_t batch, int32_t want_groups, hrag_shard_layout *out) {
    HRAG_TRY(shard_batch(e, batch));
    HRAG_REQUIRE(out != nullptr, "NULL argument");
    return ppr8_layout(e, batch, want_groups, out);
}

hrag_status hrag_shard_score_facts(hrag_engine *e, const uint16_t *q, int32_t batch, int32_t k, int32_t *idx_out,
                                   float *score_out, float *mn_out, float *mx_out, hrag_stream stream) {
    HRAG_TRY(shard_batch(e, batch));
    HRAG_REQUIRE(q && idx_out && score_out && mn_out && mx_out, "NULL argument");
    HRAG_REQUIRE(k >= 1 && k <= kTopkMax, "k=%d outside [1, %d]", k, kTopkMax);
    HRAG_REQUIRE(e->d_subj != nullptr, "engine was created without facts");
    hipStream_t s = (hipStream_t)stream;
    if (e->f_rows == 0) {
        HRAG_TRY(launch_fill_i32(idx_out, -1, (int64_t)batch * k, s));
        HRAG_HIP_TRY(hipMemsetAsync(score_out, 0, (size_t)batch * k * sizeof(float), s));
        return fill_minmax_neutral(mn_out, mx_out, batch, s);
    }
    HRAG_TRY(prep_query(e, q, batch, s, &q));
    if (batch > 16 && k <= 16 && e->d_fused_ws)
        return launch_sim_topk_fused(e->d_femb, e->f_rows, e->kdim, q, batch, k, (int32_t)e->f_offset, 0, e->d_fused_ws,
                                     e->d_fused_sel, mn_out, mx_out, idx_out, score_out, s, e->emb_dtype);
    HRAG_TRY(launch_sim_gemm(e->d_femb, e->f_rows, e->kdim, q, batch, e->d_sfact, e->ld_f, s, 0, e->emb_dtype));
    return launch_row_topk(e->d_sfact, batch, e->f_rows, e->ld_f, k, (int32_t)e->f_offset, kNormNone, idx_out,
                           score_out, mn_out, mx_out, s, e->d_topk_ws, kTopkWsBytes);
}

hrag_status hrag_shard_passage_scores(hrag_engine *e, const uint16_t *q, int32_t batch, float *mn_out,
                                      float *mx_out, hrag_stream stream) {
    HRAG_TRY(shard_batch(e, batch));
    HRAG_REQUIRE(q && mn_out && mx_out, "NULL argument");
    HRAG_REQUIRE(e->shard_aligned, "the passage embedding shard must hold exactly the passages of the owned rows");
    hipStream_t s = (hipStream_t)stream;
    if (e->p_rows == 0) return fill_minmax_neutral(mn_out, mx_out, batch, s);
    HRAG_REQUIRE(e->d_pemb != nullptr, "engine has no passage embeddings (created for hrag_retrieve_scored)");
    HRAG_TRY(prep_query(e, q, batch, s, &q));
    HRAG_TRY(launch_sim_gemm(e->d_pemb, e->p_rows, e->kdim, q, batch, e->d_spass, e->ld_p, s, 0, e->emb_dtype));
    return launch_row_minmax(e->d_spass, batch, e->p_rows, e->ld_p, mn_out, mx_out, s);
}

hrag_status hrag_shard_prior_stats(hrag_engine *e, const float *mn, const float *mx, float passage_node_weight,
                                   const int32_t *flags, int32_t batch, float *zmax_out, double *mass_out,
                                   hrag_stream stream) {
    HRAG_TRY(shard_batch(e, batch));
    HRAG_REQUIRE(mn && mx && flags && zmax_out && mass_out, "NULL argument");
    return ppr8_prior(e, mn, mx, passage_node_weight, flags, batch, zmax_out, mass_out, (hipStream_t)stream);
}

hrag_status hrag_shard_ppr_begin(hrag_engine *e, const float *mn, const float *mx, const float *zmax,
                                 const double *mass, float passage_node_weight, const int32_t *seed_vtx,
                                 const float *seed_w, const int32_t *seed_cnt, int32_t *flags, int32_t batch,
                                 float damping, int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol,
                                 int32_t n_groups, void *state0, void *state1, void *state2, int32_t *n_steps_out,
                                 hrag_stream stream) {
    if (e) e->counters.shard.fetch_add(1);
    HRAG_TRY(shard_batch(e, batch));
    HRAG_REQUIRE(mn && mx && zmax && mass && seed_vtx && seed_w && seed_cnt && flags, "NULL argument");
    HRAG_REQUIRE(damping >= 0.f && damping < 1.f, "damping %g outside [0, 1)", (double)damping);
    hrag_shard_layout lay;
    HRAG_TRY(ppr8_layout(e, batch, n_groups, &lay));
    HRAG_REQUIRE(n_groups <= 0 || lay.n_groups == n_groups || n_groups > lay.n_slabs,
                 "the engine lays %d queries out in %d exchange groups, not %d: size the state buffers with "
                 "hrag_shard_layout_query and pass its n_groups", batch, lay.n_groups, n_groups);
    uint8_t *bufs[3] = {static_cast<uint8_t *>(state0), static_cast<uint8_t *>(state1), static_cast<uint8_t *>(state2)};
    HRAG_REQUIRE(ppr_tol >= 0.f && (ppr_tol == 0.f || ppr_max_iters >= ppr_iters), "bad ppr_tol / ppr_max_iters");
    HRAG_REQUIRE(ppr_tol == 0.f || ppr_tol >= HRAG_PPR_TOL_MIN, "ppr_tol=%g is below HRAG_PPR_TOL_MIN=%g (include/hrag.h)",
                 (double)ppr_tol, (double)HRAG_PPR_TOL_MIN);
    HRAG_TRY(ppr8_begin(e, mn, mx, zmax, mass, passage_node_weight, seed_vtx, seed_w, seed_cnt, flags, batch, damping,
                        ppr_iters, lay, bufs, (hipStream_t)stream, ppr_max_iters, ppr_tol, true));
    if (n_steps_out) *n_steps_out = e->p8.n_steps;
    return HRAG_OK;
}

hrag_status hrag_shard_ppr_sweep(hrag_engine *e, int32_t sweep, int32_t group, int32_t *exchange_out,
                                 int32_t *checkpoint_out, hrag_stream stream) {
    HRAG_REQUIRE(e != nullptr, "engine is NULL");
    HRAG_REQUIRE(group >= 0, "group must be >= 0");
    HRAG_TRY(ppr8_sweep(e, sweep, group, exchange_out, (hipStream_t)stream));
    if (checkpoint_out) *checkpoint_out = e->p8.steps[sweep].decide != -1 ? 1 : 0;
    return HRAG_OK;
}

// est of the latest final sweep over the OWNED passages: read it (set == 0: float bits -> est_dev fp32 [B]), all-reduce
// MAX over the shards, write it back (set != 0)
hrag_status hrag_shard_ppr_est(hrag_engine *e, float *est_dev, int32_t set, hrag_stream stream) {
    HRAG_REQUIRE(e && est_dev && e->p8.active, "bad argument / no fp8 PPR session");
    int32_t *mine = e->d_est_f;
    const size_t bytes = (size_t)e->p8.batch * sizeof(float);   // non-negative floats: the bit patterns ARE the floats
    HRAG_HIP_TRY(hipMemcpyAsync(set ? (void *)mine : (void *)est_dev, set ? (const void *)est_dev : (const void *)mine, bytes,
                                hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return HRAG_OK;
}

hrag_status hrag_shard_ppr_decide(hrag_engine *e, int32_t sweep, hrag_stream stream) {
    HRAG_REQUIRE(e != nullptr, "engine is NULL");
    return ppr8_decide(e, sweep, (hipStream_t)stream);
}

// Host-side view of a device decision: does step `step` of the session do anything?  (A step without a gate always
// runs.)  Synchronises `stream` -- the one place of the shard interface that does: the row-sharded host loop pays a
// collective per step, so once a decision has closed a gate it wants to stop issuing the steps behind it.
hrag_status hrag_shard_ppr_gate(hrag_engine *e, int32_t step, int32_t *open_out, hrag_stream stream) {
    HRAG_REQUIRE(e && open_out && e->p8.active, "bad argument / no fp8 PPR session");
    const Ppr8Session &p = e->p8;
    if (step < 0 || step >= p.n_steps) { *open_out = 0; return HRAG_OK; }   // past the last step: nothing left to run
    const Ppr8Step &st = p.steps[step];
    if (st.gate < 0) { *open_out = 1; return HRAG_OK; }
    int32_t word = 0;
    HRAG_HIP_TRY(hipMemcpyAsync(&word, e->d_ctl + st.gate, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HRAG_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *open_out = word == st.gate_want ? 1 : 0;
    return HRAG_OK;
}

hrag_status hrag_shard_finish(hrag_engine *e, const float *mn, const float *mx, int32_t *flags, int32_t batch,
                              int32_t k, int32_t *idx_out, float *score_out, float *residual_out, int32_t *iters_out,
                              hrag_stream stream) {
    HRAG_TRY(shard_batch(e, batch));
    HRAG_REQUIRE(mn && mx && flags && idx_out && score_out, "NULL argument");
    HRAG_REQUIRE(k >= 1 && k <= e->max_topk, "k=%d outside [1, max_topk=%d]", k, e->max_topk);
    HRAG_REQUIRE(e->p8.active && e->p8.batch == batch, "no fp8 PPR session for batch %d", batch);
    hipStream_t s = (hipStream_t)stream;
    if (e->p_rows == 0) {
        HRAG_TRY(launch_fill_i32(idx_out, -1, (int64_t)batch * k, s));
        HRAG_HIP_TRY(hipMemsetAsync(score_out, 0, (size_t)batch * k * sizeof(float), s));
        HRAG_TRY(ppr8_finalize(e, flags, s));
        if (residual_out)
            HRAG_HIP_TRY(hipMemcpyAsync(residual_out, e->d_resid, (size_t)batch * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (iters_out)
            HRAG_HIP_TRY(hipMemcpyAsync(iters_out, e->d_iters_used, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        return launch_flag_zero_mass(e->d_sums, batch, flags, 2, s);
    }
    HRAG_TRY(ppr8_doc_scores(e, mn, mx, flags, batch, s, true));
    HRAG_TRY(launch_row_topk(e->d_doc, batch, e->p_rows, e->ld_p, k, (int32_t)e->p_offset, kNormNone, idx_out,
                             score_out, nullptr, nullptr, s, e->d_topk_ws, kTopkWsBytes));
    BlitList out;
    out.copy(residual_out, e->d_resid, (int64_t)batch * sizeof(float));
    out.copy(iters_out, e->d_iters_used, (int64_t)batch * sizeof(int32_t));
    return launch_blits(out, s);
}

hrag_status hrag_engine_gather_embeddings(hrag_engine *e, int32_t which, const int32_t *src_rows, int64_t n,
                                          const void *new_rows, void *out, hrag_stream stream) {
    HRAG_REQUIRE(e && src_rows && out && n >= 0, "bad argument");
    HRAG_REQUIRE(which == 0 || which == 1, "which must be 0 (facts) or 1 (passages)");
    const void *emb = which == 0 ? (const void *)e->d_femb : (const void *)e->d_pemb;
    HRAG_REQUIRE(emb != nullptr || (which == 0 ? e->f_rows : e->p_rows) == 0, "engine holds no such embeddings");
    return launch_gather_rows(emb, new_rows, src_rows, n, e->kdim * 2, out, (hipStream_t)stream);
}

hrag_status hrag_engine_set_flags(hrag_engine *e, int32_t flags, int32_t on) {
    HRAG_REQUIRE(e != nullptr, "engine is NULL");
    const int32_t runtime = HRAG_OPT_NO_FP8 | HRAG_OPT_NT_CSR | HRAG_OPT_NT_STORE | HRAG_OPT_TEMPORAL16 | HRAG_OPT_SLABS_PER_WG_1 |
                            HRAG_OPT_NO_F16 | HRAG_OPT_XCD_BLOCKED | HRAG_OPT_ACCEL;
    HRAG_REQUIRE((flags & ~runtime) == 0, "only HRAG_OPT_NO_FP8 / NO_F16 / ACCEL / NT_CSR / NT_STORE / TEMPORAL16 / SLABS_PER_WG_1 / XCD_BLOCKED can change after creation");
    if (on) e->opt_flags |= flags; else e->opt_flags &= ~flags;
    return HRAG_OK;
}

}  // extern "C"
