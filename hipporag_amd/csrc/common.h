// Internal helpers shared by the libhrag.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hrag.h"

namespace hrag {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *experiment_env(const char *name);   // errors.cpp: a measurement switch of the environment, read once, announced on stderr

#define HRAG_HIP_TRY(expr)                                                                  \
    do {                                                                                    \
        hipError_t _err = (expr);                                                           \
        if (_err != hipSuccess) {                                                           \
            ::hrag::set_error("%s -> %s (%s:%d)", #expr, hipGetErrorString(_err), __FILE__, \
                              __LINE__);                                                    \
            return HRAG_EHIP;                                                               \
        }                                                                                   \
    } while (0)

#define HRAG_TRY(expr)                     \
    do {                                   \
        hrag_status _st = (expr);          \
        if (_st != HRAG_OK) return _st;    \
    } while (0)

#define HRAG_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::hrag::set_error(__VA_ARGS__); \
            return HRAG_EINVAL;            \
        }                                  \
    } while (0)

// Launch check: hipGetLastError after every kernel launch (cheap, no sync).
#define HRAG_LAUNCH_CHECK() HRAG_HIP_TRY(hipGetLastError())

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---- monotone float <-> uint32 key (larger float => larger key); -0.0 is folded into +0.0
__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
    f += 0.0f;  // -0.0 -> +0.0 so that the two zeros tie like they do in numpy
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
// ranking key: score descending, then index descending  <=>  larger 64-bit key first
__device__ __forceinline__ uint64_t rank_key(float score, uint32_t idx) {
    return ((uint64_t)f32_to_ordered(score) << 32) | (uint64_t)idx;
}

// Convergence contract: the running per-query maximum est[q] (float bits) is raised with atomicMax by whoever holds a
// larger value.  The pre-check must see the other workgroups' updates -- a cached plain load keeps reading the
// initial 0 and every lane of every wavefront would issue its atomic (measured: +0.3 ms on a 1 ms sweep) -- so it is
// an agent-scope load; a stale value would only cost a redundant atomic, the maximum itself is order-free.
__device__ __forceinline__ int32_t est_peek(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// seeds.hip limits
constexpr int kMaxKeptFacts = 16; // kf upper bound
constexpr int kMaxSeeds = 32;     // 2 * kMaxKeptFacts

// ---- PPR state layout: [n_slabs][rows][BC] fp32, query q lives in slab q / BC, column q % BC.
struct SlabLayout {
    int32_t bc;       // slab width (4 * lanes-per-row)
    int32_t n_slabs;  // ceil(B / bc)
};

// ------------------------------------------------------------------ kernel launchers
// ppr_spmm.hip
struct SpmmArgs {
    const int32_t *row_ptr;   // [n_rows + 1] local
    const int32_t *col_idx;   // [nnz] global column ids
    const float *val;         // [nnz]
    const int32_t *row_order; // [n_short] local rows handled by the group-per-row kernel
    int32_t n_short;
    // rows above the short-row threshold, cut into segments (one wavefront each)
    const int32_t *seg_row, *seg_begin, *seg_end, *seg_slot;  // [n_seg]; slot -1: single segment
    int32_t n_seg;
    const int32_t *mrow_row, *mrow_first, *mrow_cnt;          // [n_mrow] multi-segment rows
    int32_t n_mrow;
    float *partial;           // [n_slabs][n_partial][BC] per-segment partial sums
    int32_t n_partial;
    int64_t n_rows;           // local rows
    int64_t row_offset;       // global id of local row 0
    int64_t num_vertices;     // V (rows per slab of x / y)
    const float *x;           // [n_slabs][V][BC]
    float *y;                 // [n_slabs][V][BC]
    const int32_t *row_to_tele; // [n_rows] teleport slot of local row (-1 none); nullptr => slot = global row
    const float *tele;        // [n_slabs][tele_rows][BC]
    int64_t tele_rows;
    float alpha;              // damping
    float beta;               // 1 - damping
    int32_t flags;            // HRAG_OPT_NT_CSR | HRAG_OPT_NT_STORE
};
hrag_status launch_ppr_spmm(const SpmmArgs &a, SlabLayout lay, bool main_only, hipStream_t s);
hrag_status launch_ppr_init(const SpmmArgs &a, SlabLayout lay, hipStream_t s); // y = tele (x unused)
hrag_status launch_seed_scatter(float *y, int64_t num_vertices, int64_t row_offset, int64_t n_rows,
                                const int32_t *seed_vtx, const float *seed_w,
                                const int32_t *seed_cnt, int32_t max_seeds, int32_t batch,
                                float scale, SlabLayout lay, hipStream_t s);
// column sums of the owned rows: partial [n_slabs][kColsumBlocks][BC] doubles, then sums[B] doubles
constexpr int kColsumBlocks = 256;
hrag_status launch_colsum(const float *x, int64_t num_vertices, int64_t row_offset, int64_t n_rows,
                          int32_t batch, SlabLayout lay, double *partial, double *sums,
                          hipStream_t s);

// ppr16.hip : two-stage fp16-state PPR (64 queries per 128-byte line), SELL-8 matrix
enum Ppr16Mode { kPprModeH = 0, kPprModeR = 1, kPprModeC = 2, kPprModeF = 3 };   // F: last C sweep, passage rows
constexpr int32_t kVrowNone = (int32_t)0x80000000;  // padding virtual row (no output)
// rows above this (large graphs; engine.hip sell8_seg_len picks 64 .. 256 on small ones) are cut into segments.
// Rows are sorted by length, so the 8 rows of a wavefront have similar trip counts and the longest start first:
// only real hubs need cutting.
constexpr int kSell8SegLen = 512;
constexpr int kSell8MaxSegs = 1024;
struct Ppr16Args {
    const int2 *pairs;         // [total_steps * 64] (col, fp32 bits of val), step-major per chunk
    uint32_t pairs_bytes;      // size of the pairs array incl. the read-ahead padding (< 2^31)
    const int2 *chunk_meta;    // [n_chunks] (first step, number of steps)
    const int32_t *vrow;       // [n_chunks * 8] row id >= 0 | -(partial slot + 1) | kVrowNone
    int32_t n_chunks;
    const int32_t *lrow_row, *lrow_first, *lrow_cnt;  // [n_lrow] long rows and their partial slots
    int32_t n_lrow;
    int32_t n_partial;
    const int32_t *seg_lrow;   // [n_partial] long row of a partial slot
    int32_t *lcount;           // [n_slabs][n_lrow] arrival counters (zero between launches): the LAST segment of a long
                               // row to arrive adds the row's partial sums up and finishes it -- no second kernel
    float *partial;            // [n_slabs][n_partial][64] fp32
    int64_t num_vertices;
    const uint16_t *x;         // gather source, fp16 [n_slabs][V][64]
    uint16_t *y;               // output, same layout
    const uint16_t *aux;       // mode C: r
    const int32_t *row_slot;   // [V] teleport row of a vertex (passage or seed row) or -1
    const float *tele;         // fp32 [n_slabs][tele_rows][64]
    int64_t tele_rows;
    float alpha, beta, cscale;
    // mode H, first sweep: only columns whose bit is set are gathered (h_0 = f16(v) is zero elsewhere); nullptr: all
    const uint32_t *colmask = nullptr;
    uint32_t colmask_bytes = 0;
    // mode F (the matrix holds the passage rows only): x = h + c / cscale, fp32 [n_slabs][p_rows][64], passage order
    const uint16_t *hfin = nullptr;
    float *out = nullptr;
    int64_t p_rows = 0;
    // mode F, convergence contract: est[q] = max over the passage rows of |x_new - x_old| / x_new as float bits
    // (atomicMax; x_old from the own row of the gather source); nullptr: not measured
    int32_t *est = nullptr;
    float *est_ws = nullptr;   // scratch [n_slabs][n_chunks][64]: every wavefront's maximum (reduced by launch_est_reduce)
    int32_t batch = 0;
    // HRAG_OPT_ACCEL, modes H / C: the sweep is a Chebyshev step y = omega (plain result) + (1 - omega) prev, prev = the
    // iterate before x (own row; may alias y: a row is read before it is written, by its one owner; nullptr: zero)
    float omega = 1.f;
    const uint16_t *prev = nullptr;
    // convergence contract: a launch whose gate word differs from gate_want returns at once (the extension stages are
    // enqueued unconditionally and the DEVICE decides which of them run: ppr8.hip ppr8_decide_kernel)
    const int32_t *gate = nullptr;
    int32_t gate_want = 0;
};
// nt: bit0 non-temporal (col, val) loads, bit1 non-temporal state stores
hrag_status launch_ppr16_sweep(const Ppr16Args &a, int mode, int n_slabs, int nt, bool main_only,
                               hipStream_t s);
hrag_status launch_ppr16_init(const Ppr16Args &a, int n_slabs, hipStream_t s);  // y = f16(v)
hrag_status launch_ppr16_scale(const float *mn, const float *mx, const float *ssum, int64_t n_passages,
                               float passage_weight, const float *seed_w, const int32_t *seed_cnt,
                               const int32_t *flags, int32_t batch, float *qscale, hipStream_t s);
// bc: columns per teleport row (64 on the fp16 path, BP on the small-batch path); qscale may be null
// row_offset / n_rows: the owned vertex range (row_slot is indexed by LOCAL row); n_passages = owned passages
hrag_status launch_ppr16_seed_rows(const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt,
                                   const float *qscale, int32_t batch, int64_t n_passages,
                                   int64_t num_vertices, int32_t *row_slot, float *tele,
                                   int64_t tele_rows, int32_t bc, hipStream_t s, int64_t row_offset = 0,
                                   int64_t n_rows = -1);

// ppr8.hip : staged fp8 (e4m3) state + fp32 true residual, 128 queries per 128-byte line, SELL-8 over the
// OWNED rows (row shard; single GPU: all rows) with the row-normalised values At = D^-1 A
enum Ppr8Mode { kP8ModeC = 0, kP8ModeB = 1, kP8ModeF = 2, kP8ModeB0 = 3 };   // B0: first boundary, R_in = b v/d
constexpr int kP8MaxStages = 12;      // stage lengths 1,2,3,..,3,(1|2) => <= 11 stages for ppr_iters <= 30
constexpr float kP8C0Scale = 128.f;   // c_0 = Q(v/d * 2^7),  max(v/d) in (1/2, 1]
constexpr int32_t kFlagFp8Saturated = 8;   // flags bit 3 (HRAG_FLAG_FP8_SATURATED)
constexpr int32_t kFlagNotConverged = 16;  // flags bit 4 (HRAG_FLAG_NOT_CONVERGED)
constexpr int kP8DynInv = kP8MaxStages + 4;   // offset of the 1 / cs half of the dynamic scale table
constexpr int kP8MaxExt = 4;          // extension stages the convergence contract may add: 1, 2, 3, 3 sweeps -- a
                                      // prediction just above the tolerance costs one sweep, a slowly mixing graph gets 9
__host__ __device__ inline int p8_ext_sweeps(int n_ext) {   // sweeps of the first n_ext extension stages
    return n_ext <= 0 ? 0 : n_ext == 1 ? 1 : 3 * (n_ext - 1);
}
struct Sell8Dev {
    const int2 *pairs;         // (col, fp32 bits of the value), step-major per chunk (ppr16.hip layout)
    uint32_t pairs_bytes;      // incl. the read-ahead padding (< 2^31)
    const int2 *chunk_meta;    // [n_chunks] (first step, number of steps)
    const int32_t *vrow;       // [n_chunks * 8] LOCAL row >= 0 | -(partial slot + 1) | kVrowNone
    int32_t n_chunks;
    const int32_t *lrow_row, *lrow_first, *lrow_cnt;  // [n_lrow] long rows and their partial slots
    int32_t n_lrow;
    int32_t n_partial;
    const int32_t *seg_lrow;   // [n_partial] long row of a partial slot
    int32_t *lcount;           // [n_slabs][n_lrow] arrival counters (see Ppr16Args)
    const int32_t *pslot;      // [n_chunks] dense number of a chunk that holds a passage row (else -1): its row of the
    int32_t n_pchunks;         // est scratch [n_slabs][n_pchunks][128] (convergence contract)
};
// State buffers: e4m3 [n_groups][V + 1][spg][128]; slab s lives in group s / spg, column block s % spg;
// row V of every group is all-zero (the target of masked-out gathers in mode B0).  An owner's rows
// [row_offset, row_offset + n_rows) of one group are one contiguous block = the unit of the exchange.
struct Ppr8Args {
    Sell8Dev m;                // main matrix (owned rows) or, in mode F, its passage rows only
    float *partial;            // [n_slabs][n_partial][128] fp32 (lane-interleaved rows)
    int64_t row_offset, n_rows;   // owned rows (global id of local row 0, count)
    int32_t spg;               // slabs per exchange group
    uint32_t row_stride;       // spg * 128: bytes between consecutive vertices of a group
    int64_t group_bytes;       // (V + 1) * row_stride
    const uint8_t *x;          // gather source
    uint8_t *y;                // mode C: the new iterate; mode B: rt of the next stage (owned rows written)
    const uint8_t *rt;         // mode C / rio bit 0: the quantised right-hand side of the stage (being closed)
    float *R;                  // true residual, fp32 [n_slabs][n_rows][128] lane-interleaved (LOCAL rows)
    uint16_t *rho;             // ... or its fp16 remainder next to the fp8 right-hand side (same shape), see rio
    int32_t rio;               // bit 0: R is read as (rt + rho) / cs; bit 1: R is written as rho (ppr8.hip finish_row)
    float alpha, beta, inv_cs, cs_next;
    // mode C: y = Q(c_mul * (At x) + r_mul * rt).  Plain stage sweep: (alpha, 1); accelerated stages (HRAG_OPT_ACCEL,
    // csrc/shard.hip ppr8_plan_accel): second iterate of a stage (w2 alpha, w2), third (w3 alpha, 1)
    float c_mul, r_mul;
    // HRAG_OPT_ACCEL: stage scales measured on the device (ppr8.hip scale_inv / scale_cs): dyn[k] = cs of stage k,
    // dyn[kP8DynInv + k] = 1 / cs; dyn_stage = the stage a boundary / final launch closes; mmax_ws = one slot per
    // (chunk, slab unit of this launch) for the boundary's max |R cs'|, mmax_atomic = the same for long rows
    const float *dyn = nullptr;
    int32_t dyn_stage = 0;
    float *mmax_ws = nullptr;
    int32_t *mmax_atomic = nullptr;
    int32_t mmax_units = 1, mmax_slab0 = 0;
    const float *tele;         // fp32 [n_slabs64][tele_rows][64]: v at the owned passages, then the seed rows
    int64_t tele_rows;
    int32_t n_slabs64;
    const int32_t *row_slot;   // [n_rows] teleport slot of a LOCAL row: slot < p_rows <=> owned passage number
    const float *deg;          // [V] weighted degree (1 for isolated vertices), GLOBAL vertex ids
    int64_t p_rows;            // owned passages
    uint8_t *stage_out;        // mode B / B0: c of this stage at the owned passage rows, [n_slabs][p_rows][128]
    // mode F
    const uint8_t *stage[kP8MaxStages];   // stage_out of the earlier stages (the last stage's c is read from x)
    float stage_inv[kP8MaxStages];        // 1 / cs of every stage, the last one included
    int32_t n_stage;
    float *out;                // x = d z at the owned passages, fp32 [n_slabs64][p_rows][64] (passage order)
    // mode B0: only columns with v != 0 (passages, seeds) are gathered; the others read the zero row
    const uint32_t *colmask;   // [ceil(V / 32)] bit = column may be non-zero in c_0
    uint32_t colmask_bytes;
    uint32_t zero_row;         // V
    int32_t *flags;            // [batch] bit 3 is set when a value had to be clamped to the e4m3 range
    int32_t batch;
    int32_t slab0, n_slabs;    // 128-query slabs covered by this launch
    int32_t wps;               // slabs handled inside one workgroup (1, 2 or 4 wavefronts per chunk); 0 / 1 = one
    int32_t cg_per_xcd;        // > 0 (HRAG_OPT_XCD_BLOCKED): XCD x walks chunk groups [x * cg_per_xcd, (x + 1) * cg_per_xcd)
    // convergence contract (csrc/shard.hip, ppr8_begin): a launch whose gate word differs from gate_want returns at
    // once -- the extension stages and the alternative final sweeps are enqueued unconditionally and the DEVICE
    // decides which of them run (no host synchronisation, graph-capture safe)
    const int32_t *gate = nullptr;
    int32_t gate_want = 0;
    // modes B / F: est[q] = max over the owned passage rows of |R_p| / z_p as float bits (atomicMax), the relative
    // size of the last update of the passage scores; nullptr: not computed.  Mode B then needs stage / stage_inv /
    // n_stage like mode F (z_p = sum of the stage copies + c / cs + R).
    int32_t *est = nullptr;
    float *est_ws = nullptr;   // scratch [n_slabs][m.n_pchunks][128]: the maximum of every wavefront that holds a passage row
};
hrag_status launch_ppr8_sweep(const Ppr8Args &a, int mode, bool main_only, hipStream_t s);
hrag_status launch_ppr8_gather_replay(const Ppr8Args &a, hipStream_t s);   // measurement only (ppr8.hip)
// c_0 = Q(v/d * c0_scale) on the owned rows of slabs [slab0, slab0 + n_slabs)
hrag_status launch_ppr8_init(const Ppr8Args &a, float c0_scale, hipStream_t s);
// per-query statistics of the passage prior over the OWNED passages (scores fp32 [B, ld], local order):
//   zmax[q]  = max_p minmax(score_qp) * pinvdeg[p]          (0 for queries on the DPR fallback)
//   mass[q]  = sum_p float(minmax(score_qp) * weight)       mass[B + q] = the same over isolated passages
// part: kP8PriorSplit * batch * 2 doubles of scratch
constexpr int kP8PriorSplit = 16;
hrag_status launch_ppr8_prior(const float *scores, int64_t ld, int64_t p_rows, const float *mn, const float *mx,
                              float passage_weight, const float *pinvdeg, const uint8_t *piso,
                              const int32_t *flags, int32_t batch, int32_t *zmax_bits, double *part,
                              float *zmax, double *mass, hipStream_t s);
// qscale[q] (power of two, max v/d in (1/2, 1]) from the GLOBAL zmax + the seeds; sums[q] = the mass of the
// `iters`-sweep iterate of x <- a P x + (1 - a) v from x_0 = v, in the units of the scaled v (analytic:
// P is column-stochastic except for isolated vertices, whose mass is known sweep by sweep)
hrag_status launch_ppr8_scale(const float *zmax, const double *mass, float passage_weight,
                              const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt,
                              const float *deg, const uint8_t *iso, int64_t num_vertices, const int32_t *flags,
                              int32_t batch, float damping, int32_t iters, float *qscale, double *sums,
                              hipStream_t s, int32_t n_tab = 1, int64_t tab_stride = 0);
// the convergence contract's two tiny kernels (ppr8.hip): decision j after a checkpoint boundary; the per-query
// results after the last step
// est[slab * w + col] = max(itself, column maxima of ws[slab][chunk][w]) for slabs [slab0, slab0 + n_slabs)
hrag_status launch_est_reduce(const float *ws, int32_t n_chunks, int32_t w, int32_t slab0, int32_t n_slabs, int32_t batch,
                              int32_t *est, const int32_t *gate, int32_t gate_want, hipStream_t s);
hrag_status launch_ppr8_decide(int32_t *est_f, const int32_t *flags, int32_t batch, float g, float tol, int32_t j,
                               int32_t e_max, int32_t *ctl, hipStream_t s);
hrag_status launch_ppr8_finalize(const int32_t *est_f, int32_t *flags, int32_t batch, float g, float tol,
                                 int32_t iters, const int32_t *ctl, int32_t e_max, const double *mass_tab,
                                 int64_t tab_stride, double *sums, float *resid, int32_t *iters_used, hipStream_t s);
// HRAG_OPT_ACCEL: the scale of stage `stage + 2` from the maximum the boundary closing `stage` measured (seed != 0:
// write the first two scales; finalize = false: only fold this launch's maximum into word[0] -- an exchange group of a
// step that is not the step's last one)
hrag_status launch_ppr8_next_scale(const float *ws, int32_t n_slots, int32_t *word, float *dyn, int32_t stage,
                                   float kappa_growth, int32_t seed, float cs0, float cs1, const int32_t *gate,
                                   int32_t gate_want, hipStream_t s, bool finalize = true);
// colmask |= bits of the seed vertices
hrag_status launch_ppr8_mask_seeds(const int32_t *seed_vtx, const int32_t *seed_cnt, int32_t batch,
                                   int64_t num_vertices, uint32_t *colmask, hipStream_t s);

// ppr_sv.hip : small batches (B <= 8), fp32 state [V][BP], same SELL-8 matrix
struct PprSvArgs {
    const int2 *pairs;
    uint32_t pairs_bytes;
    const int2 *chunk_meta;
    const int32_t *vrow;
    int32_t n_chunks;
    const int32_t *lrow_row, *lrow_first, *lrow_cnt;
    int32_t n_lrow;
    const int32_t *seg_lrow;   // [n_partial] long row of a partial slot
    int32_t *lcount;           // [n_lrow] arrival counters (see Ppr16Args)
    float *partial;            // [n_partial][BP]
    int64_t num_vertices;
    const void *x;             // [V][BP] fp32, or fp16 when half_state
    void *y;                   // [V][BP]
    const int32_t *row_slot;   // [V] or nullptr (slot = vertex)
    const float *tele;         // [tele_rows][BP]
    float alpha, beta;
    int32_t nt;                // non-temporal (col, val) loads
    // two-stage fp16 state (ppr_sv.hip header): mode = SvMode (0 plain / H, 1 residual, 2 correction, 3 final)
    int32_t half_state = 0, mode = 0;
    const uint16_t *aux16 = nullptr;   // modes 2, 3: r
    const uint16_t *h16 = nullptr;     // mode 3: h (own row)
    float *xout = nullptr;             // mode 3: x = h + c / cscale, fp32 [V][BP], written at the matrix' rows only
    float cscale = 64.f;
    const uint32_t *colmask = nullptr; // first sweep: bit = column may be non-zero in x_0 (nullptr: gather everything)
    // last sweep (kSvFinal, or the plain fp32 sweep over the passage rows), convergence contract: est[q] = max over
    // the rows of |x_new - x_old| / x_new as float bits (atomicMax); nullptr: not measured
    int32_t *est = nullptr;
    float *est_ws = nullptr;   // scratch [n_chunks][BP]: every wavefront's maximum (reduced by launch_est_reduce)
    int32_t batch = 0;
    // HRAG_OPT_ACCEL, fp16 state, modes plain (H) / correction: Chebyshev step, see Ppr16Args
    float omega = 1.f;
    const uint16_t *prev = nullptr;
    // convergence contract: conditional launch (see Ppr16Args)
    const int32_t *gate = nullptr;
    int32_t gate_want = 0;
};
hrag_status launch_ppr_sv_sweep(const PprSvArgs &a, int bp, bool main_only, hipStream_t s);
hrag_status launch_ppr_sv_init(const PprSvArgs &a, int bp, hipStream_t s);
hrag_status launch_ppr_sv_tele(const float *scores, int64_t ld, int64_t n, int32_t batch, const float *mn,
                               const float *mx, float weight, const int32_t *flags, float *tele, int bp,
                               hipStream_t s, const float *qscale = nullptr);
// sums[b] = mass of the `iters`-sweep iterate, closed form.  tele: [slab][slab_rows][stride], query b in slab
// b / stride, column b % stride (ppr_sv: one slab, stride = bp; ppr16: stride 64); the first tele_rows rows of a slab
// are summed; piso / iso: isolated flags of the passages / of all vertices; passage_of_vertex: [V] passage number or
// -1; part: batch * 64 * 2 doubles of scratch
hrag_status launch_ppr_sv_mass(const float *tele, int stride, int64_t slab_rows, int64_t n_passages, int64_t tele_rows,
                               const uint8_t *piso, const uint8_t *iso, const int32_t *passage_of_vertex,
                               const int32_t *seed_vtx, const float *seed_w, const int32_t *seed_cnt, const float *qscale,
                               int64_t num_vertices, int32_t batch, float damping, int32_t iters, double *part,
                               double *sums, hipStream_t s);
hrag_status launch_ppr_sv_reset(const float *reset, int64_t n, int32_t batch, float *tele, int bp,
                                hipStream_t s);
// partial: 256 * bp doubles
hrag_status launch_ppr_sv_colsum(const float *x, int64_t n, int bp, double *partial, double *sums,
                                 hipStream_t s);
hrag_status launch_ppr_sv_rows(const float *x, const int32_t *gather, int64_t n, int32_t batch,
                               const double *sums, float *out, int64_t ld, const float *alt, int64_t alt_ld,
                               const float *mn, const float *mx, const int32_t *flags, int bp,
                               hipStream_t s);

// layout.hip : [B, n] row-major  <->  slab layout, with the fused element-wise stages
// slab_rows: rows per slab of the destination (>= n; 0 means n); qscale: optional per-query factor
enum ToSlabMode { kSanitize = 0, kMinMaxScale = 1 };
hrag_status launch_rows_to_slab(const float *rows, int64_t ld, int64_t n, int32_t batch,
                                ToSlabMode mode, const float *mn, const float *mx, float scale,
                                const int32_t *skip_flags, float *slab, SlabLayout lay,
                                hipStream_t s, int64_t slab_rows = 0, const float *qscale = nullptr);
// out[q][i] = x[slab(q)][gather ? gather[i] : i][col(q)] / sums[q]
//   alt != nullptr and (flags[q] & 1): out[q][i] = minmax(alt[q][i]) instead (DPR fallback)
hrag_status launch_slab_to_rows(const float *slab, int64_t slab_rows, const int32_t *gather,
                                int64_t n, int32_t batch, const double *sums, float *out,
                                int64_t ld, const float *alt, int64_t alt_ld, const float *mn,
                                const float *mx, const int32_t *flags, SlabLayout lay,
                                hipStream_t s);
hrag_status launch_gather_rows(const void *emb, const void *fresh, const int32_t *src, int64_t n, int32_t row_bytes,
                               void *out, hipStream_t s);
hrag_status launch_fill_i32(int32_t *dst, int32_t value, int64_t n, hipStream_t s);
// Several small fills / device-to-device copies of a call in ONE launch (a hipMemsetAsync / hipMemcpyAsync each is its
// own ~7 us blit kernel; a retrieve had 9 .. 13 of them).  src == nullptr: zero fill of `reps` regions of `bytes` bytes,
// `stride` bytes apart; otherwise one copy (reps = 1).  Sizes and addresses are multiples of 4 bytes.
struct BlitOp {
    void *dst;
    const void *src;
    int64_t bytes, stride;
    int32_t reps;
};
struct BlitList {
    BlitOp op[8];
    int32_t n = 0;
    bool overflow = false;   // a ninth operation was queued: launch_blits fails (HRAG_EINVAL) instead of dropping it
    void zero(void *dst, int64_t bytes, int32_t reps = 1, int64_t stride = 0) {
        if (!(dst && bytes > 0 && reps > 0)) return;
        if (n < 8) op[n++] = BlitOp{dst, nullptr, bytes, stride, reps}; else overflow = true;
    }
    void copy(void *dst, const void *src, int64_t bytes) {
        if (!(dst && src && bytes > 0)) return;
        if (n < 8) op[n++] = BlitOp{dst, src, bytes, 0, 1}; else overflow = true;
    }
};
hrag_status launch_blits(const BlitList &l, hipStream_t s);
hrag_status launch_flag_zero_mass(const double *sums, int32_t batch, int32_t *flags, int32_t bit,
                                  hipStream_t s);
// convergence contract, fp32 slab state: est[q] = max over the passages of |x - x_prev| / x (float bits, atomicMax)
hrag_status launch_passage_delta(const float *x, const float *x_prev, int64_t slab_rows, const int32_t *passage_vertex,
                                 int64_t n_passages, int32_t batch, SlabLayout lay, int32_t *est, hipStream_t s);

// sim_gemm.hip : S[b][m] = sum_k Q[b][k] * E[m][k]   (bf16 in, fp32 out, ld in elements)
// dtype: HRAG_BF16 | HRAG_FP16 element type of emb AND q (16-bit patterns)
hrag_status launch_sim_gemm(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q,
                            int32_t batch, float *out, int64_t ld, hipStream_t s, int32_t accumulate = 0,
                            int32_t dtype = HRAG_BF16);

// sim_gemm256.hip : the same scores (bit-identical) for batch > 64, dim % 64 == 0, rows >= 256: 256-row x
// (128 | 256)-query workgroup tiles, LDS-direct loads, 128 x (64 | 128) wave tiles
bool sim_gemm256_serves(int64_t rows, int32_t dim, int32_t batch);
bool sim_gemm_force_small_tiles();
hrag_status launch_sim_gemm256(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q, int32_t batch,
                               float *out, int64_t ld, float *tmax, float *tmin, hipStream_t s, int32_t dtype,
                               int64_t row_ld = 0);

// fused similarity + top-k for k <= 16 (no [B, rows] score matrix; bit-identical to the two-step path)
//   ws: 2 * sim_fused_tiles(rows) * batch floats; sel: sim_fused_sel_ints(batch) ints, zeroed ONCE at allocation
//   (per-query records: selected tiles, arrival counter, candidate keys); mn / mx: batch floats
int64_t sim_fused_tiles(int64_t rows);
int64_t sim_fused_sel_ints(int32_t batch);
int64_t sim_fused_sel_ints(int32_t batch);   // ints of the `sel` workspace (zero it once, at allocation)
hrag_status launch_sim_topk_fused(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q,
                                  int32_t batch, int32_t k, int32_t idx_offset, int32_t normalize,
                                  float *ws, int32_t *sel, float *mn, float *mx, int32_t *idx_out,
                                  float *val_out, hipStream_t s, int32_t dtype = HRAG_BF16, int32_t approx_dim = 0,
                                  float cut = -INFINITY, int32_t *overflow = nullptr);

// sim_gemv.hip : the same for batch <= 8 (streams E once, queries in registers); false = not handled
bool launch_sim_gemv(const uint16_t *emb, int64_t rows, int32_t dim, const uint16_t *q, int32_t batch,
                     float *out, int64_t ld, hipStream_t s, int32_t dtype = HRAG_BF16);

// topk.hip
constexpr int kTopkMax = 2048;
enum TopkNorm { kNormNone = 0, kNormMinMax = 1 };
// ws / ws_bytes: optional workspace (kTopkWsBytes) that lets small batches split every row over
// several workgroups (two-level selection, same result)
constexpr size_t kTopkWsBytes = (size_t)64 * 4096 * 8 + (size_t)64 * 64 * 8;
hrag_status launch_row_topk(const float *scores, int32_t batch, int64_t n, int64_t ld, int32_t k,
                            int32_t idx_offset, TopkNorm norm, int32_t *idx_out, float *val_out,
                            float *mn_out, float *mx_out, hipStream_t s, void *ws = nullptr,
                            size_t ws_bytes = 0);
hrag_status launch_row_minmax(const float *scores, int32_t batch, int64_t n, int64_t ld,
                              float *mn_out, float *mx_out, hipStream_t s, float *sum_out = nullptr);

// knn.hip : fp32 rows -> the 3 * dim bf16 layout of HRAG_F32_SPLIT engines ([hi | lo | hi]; queries [hi | hi | lo])
hrag_status launch_split3(const float *x, int64_t rows, int32_t dim, int32_t as_query, uint16_t *out, hipStream_t s,
                          int32_t normalize = 0);

// seeds.hip
hrag_status launch_build_seeds(const int32_t *kept_idx, const float *kept_score,
                               const int32_t *kept_count, int32_t kf, int32_t link_top_k,
                               int32_t batch, const int32_t *subj, const int32_t *obj,
                               int64_t n_facts, const int32_t *num_chunks, int64_t num_vertices,
                               int32_t *seed_vtx, float *seed_w, int32_t *seed_cnt,
                               int32_t *flags, hipStream_t s);

}  // namespace hrag
