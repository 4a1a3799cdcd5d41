/*
 * hrag.h -- C ABI of libhrag.so, the MI355X (gfx950) retrieval hot path of HippoRAG.
 *
 * The reference (OSU-NLP-Group/HippoRAG) has no FFI for this path: the seam is a
 * set of Python methods on class HippoRAG (src/hipporag/HippoRAG.py).  Each entry
 * point below names the reference method(s) it replaces; INTEGRATION.md shows
 * the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C types only: pointers + sizes; no torch / C++ types cross the ABI;
 *   - every function returns an hrag_status (0 = OK); hrag_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - pointers named *_dev are DEVICE pointers on the engine's device; pointers
 *     in the *_desc structs used by hrag_engine_create may be host OR device
 *     (copied with hipMemcpyDefault; the caller may free them right after);
 *   - every compute entry point takes the HIP stream to enqueue on (pass
 *     torch.cuda.current_stream().cuda_stream) and returns after enqueueing: no
 *     device synchronisation, no allocation in the call path (graph-capture safe);
 *   - ONE call in flight per engine HANDLE: the workspace (scores, PPR state, seeds) belongs to the handle.  Calls on one
 *     stream queue up as usual; a call on ANOTHER stream is ordered behind the previous call ON THE DEVICE (the
 *     library makes the new stream wait for the event that ends the previous call: no host synchronisation, no
 *     error -- a multi-stream pipeline just serialises on this engine's workspace); a call from a second THREAD
 *     while one is still inside the library is REJECTED with HRAG_EBUSY (never a silent race).  Real concurrency =
 *     one WORKSPACE per stream / thread: hrag_workspace_create gives a second handle on the same index (nothing of
 *     the index is copied) whose calls run concurrently with this one's -- SURVEY.md 8(b).  Replays of a captured HIP
 *     graph bypass the library and are NOT ordered against direct calls on other streams: keep them on one stream;
 *   - bf16 = the upper 16 bits of an IEEE-754 binary32, passed as uint16_t (fp16 engines:
 *     IEEE binary16 bit patterns in the same uint16_t slots);
 *   - all index outputs are int32, all score outputs fp32;
 *   - ranking rule everywhere: score descending, ties -> larger index first
 *     (== numpy.argsort(x, kind="stable")[::-1]; the reference's np.argsort default
 *     leaves ties undefined, HippoRAG.py:1500,1688,1746).
 */
#ifndef HRAG_H_
#define HRAG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version history of the ABI (hrag_version() = major * 1000 + minor):
 *   0.6  error-bound constants, HRAG_PPR_TOL_MIN; stage plan 1+2+3+4+4+4+2 for a FIXED sweep count.  Note for callers
 *        that read residual_out of a fixed-count call (ppr_tol = 0): that plan REPORTS about 2.2x the residual of the
 *        0.5 plan at the same true error (the last right-hand side is quantised later) -- a host-side threshold on it
 *        must be re-derived; a call with ppr_tol > 0 takes the plan ending on 2 + 1, so a fixed-count call and a
 *        tolerance call at the same 20 sweeps are no longer bit-identical;
 *   0.7  hrag_workspace_create, hrag_engine_stats (hrag_stats), hrag_ppr_sweeps flag 256 (gather replay); hrag_comm +
 *        hrag_shard_score_facts_all / hrag_shard_retrieve / hrag_shard_workspace_bytes (one call per phase on a row shard);
 *        hrag_shard_ppr_sweep enforces the ascending (step, group) order of a session with measured stage scales;
 *        later in 0.7 (no signature change): the accelerated two-stage fp16 plan (HRAG_OPT_ACCEL, batch <= 64, ppr_tol = 0)
 *        carries a margin of 16 instead of 4: 16 sweeps stand for 20 at damping 0.5 (14 before);
 *   0.8  hrag_sim_topk_min_score (the thresholded fused top-k of the index-time KNN: prefix first pass + cut). */
#define HRAG_VERSION_MAJOR 0
#define HRAG_VERSION_MINOR 8

typedef enum hrag_status {
    HRAG_OK = 0,
    HRAG_EINVAL = 1,      /* bad shape / null pointer / unsupported option                    */
    HRAG_ENOMEM = 2,      /* hipMalloc failed                                                  */
    HRAG_EHIP = 3,        /* a HIP runtime call failed; text in hrag_last_error()              */
    HRAG_EBUSY = 4,       /* another THREAD is inside a call on this HANDLE (its workspace is in use; a second handle  */
                          /* from hrag_workspace_create runs concurrently; calls on other streams queue on the device) */
    HRAG_ECAPACITY = 5    /* batch / k larger than the engine was created for                  */
} hrag_status;

typedef struct hrag_engine hrag_engine; /* opaque */
typedef void *hrag_stream;              /* hipStream_t */

/* Knowledge graph as the PPR consumes it (replaces the igraph object handed to
 * graph.personalized_pagerank at HippoRAG.py:1736-1743).
 * CSR over OUTPUT vertices: row i lists the in-neighbours j of i with
 *   val = A[i][j] / sum_i A[i][j]           (column-stochastic, already normalised)
 * where A is the symmetric weighted adjacency in which parallel igraph edges are
 * summed (HippoRAG.py:1189-1223, :906-910).  hipporag_amd.graph.build_csr builds it.
 * Row sharding (multi-GPU): the engine owns rows [row_offset, row_offset + n_rows)
 * of a graph with num_vertices columns; single GPU: row_offset = 0, n_rows = V. */
typedef struct hrag_graph_desc {
    int64_t num_vertices;          /* V: columns of P == length of the PPR vector            */
    int64_t row_offset;            /* first global row owned by this engine                  */
    int64_t n_rows;                /* rows owned                                             */
    int64_t nnz;                   /* entries in the owned rows                              */
    const int32_t *row_ptr;        /* [n_rows + 1], row_ptr[0] == 0                          */
    const int32_t *col_idx;        /* [nnz] global column (source vertex) ids                */
    const float *val;              /* [nnz]                                                  */
    int64_t n_passages;            /* Np (global number of passages)                         */
    const int32_t *passage_vertex; /* [Np] vertex id of passage p (passage_node_idxs, :1333) */
    const double *col_sum;         /* [V] optional (may be NULL): sum_i A[i][j], the weighted degree   */
                                   /* each column was normalised with (0 for a vertex without edges). */
                                   /* Enables the fp8-state PPR for batches > 64 (csrc/ppr8.hip), which */
                                   /* iterates in the degree-scaled variable x / col_sum.             */
} hrag_graph_desc;

typedef enum hrag_dtype {
    HRAG_BF16 = 0,
    HRAG_FP16 = 1,     /* IEEE binary16 = BASELINE configs[4] */
    HRAG_F32_SPLIT = 2,/* fp32 in, fp32-faithful scores: what the reference computes (np.dot of fp32 matrices,           */
                       /* HippoRAG.py:1342-1345,1459,1496; embedding_store.py:216-221).  `data` is fp32 [rows, dim]; the  */
                       /* engine stores every vector as hi + lo (two fp16: 22 significant bits) in the layout            */
                       /* [hi | lo | hi] and a query as [qhi | qhi | qlo], so that one fp16 MFMA dot product of          */
                       /* 3 * dim elements IS hi.qhi + lo.qhi + hi.qlo = the fp32 product up to 2^-21 |x||q| -- through  */
                       /* kernel unchanged, at 3x the embedding stream.  Every q_*_dev pointer of such an engine is      */
                       /* fp32 [B, dim].  Ranking with bf16-rounded embeddings instead flips near-tied facts / passages  */
                       /* (tests/test_ref_golden.py counts them on the reference-run fixture ref_synth_f32.npz)             */
    HRAG_F32_SPLIT_ROWS = 3 /* the same engine from rows that are ALREADY in the split layout: `data` is fp16             */
                       /* [rows, 3 * dim] as hrag_split_f32 / hrag_engine_gather_embeddings produce it (`dim` stays the  */
                       /* logical dimension): the index-update path, where the held rows never leave the device          */
} hrag_dtype;

/* Row-major, L2-normalised embedding matrix (self.fact_embeddings /
 * self.passage_embeddings, HippoRAG.py:1343-1345), rounded to bf16 or fp16, or fp32 (HRAG_F32_SPLIT); facts and
 * passages use the same dtype; every q_*_dev query pointer then carries that dtype too (declared uint16_t for the
 * 16-bit types; fp32 [B, dim] behind the same pointer on an HRAG_F32_SPLIT engine).
 * Row sharding: this engine holds rows [row_offset, row_offset + rows). */
typedef struct hrag_embed_desc {
    int64_t rows;
    int64_t row_offset;
    int32_t dim; /* D, multiple of 8 */
    hrag_dtype dtype;
    const void *data;
} hrag_embed_desc;

/* What graph_search_with_fact_entities needs to turn kept facts into seeds
 * (HippoRAG.py:1583-1606): the vertex of each fact's subject / object phrase
 * (-1 when md5("entity-"+phrase) is not a graph node) and, per vertex, the number
 * of chunks the entity occurs in (len(ent_node_to_chunk_ids[key]); 0 = do not divide). */
typedef struct hrag_fact_desc {
    int64_t n_facts;            /* global F                                       */
    const int32_t *subj_vertex; /* [F]                                            */
    const int32_t *obj_vertex;  /* [F]                                            */
    const int32_t *num_chunks;  /* [V]                                            */
} hrag_fact_desc;

#define HRAG_OPT_NATURAL_ROW_ORDER 1 /* keep CSR row order instead of degree-descending          */
#define HRAG_OPT_NT_CSR 2            /* fp32-state kernel: non-temporal loads for the col_idx / val stream */
#define HRAG_OPT_NT_STORE 4          /* fp32-state kernel: non-temporal stores for the new PPR state       */
#define HRAG_OPT_F32_STATE 8         /* plain CSR + fp32 slab state only: none of the staged fp8 state  */
                                     /* (batch > 64), the two-stage fp16 state (batch > 8, ppr_iters    */
                                     /* >= 11) and the small-batch kernels (batch <= 8); same 1e-5 bar  */
#define HRAG_OPT_TEMPORAL16 16       /* fp16-state kernels: plain instead of non-temporal (col, val)   */
                                     /* loads and state stores (non-temporal is 3 % faster at cfg 3)   */

#define HRAG_OPT_NO_FP8 32           /* never take the fp8-state PPR (batch > 64, 16 <= ppr_iters <= 30,                 */
                                     /* damping^ppr_iters <= 2^-20, col_sum given); the fp16 / fp32 state serves instead */

#define HRAG_OPT_ROWS_BY_MINCOL 64    /* experiment: SELL-8 rows of equal length ordered by smallest column id  */
#define HRAG_OPT_ROWS_BFS 128         /* experiment: ... by breadth-first rank (no L2 gain on  */
                                      /* the benchmark graph; kept for graphs with community structure)           */

#define HRAG_OPT_SLABS_PER_WG_1 256   /* fp8 sweep: one slab per workgroup (4 chunks) instead of the wavefronts of a   */
                                      /* workgroup sharing a chunk's (col, val) stream across 2 / 4 slabs             */

/* 512: was HRAG_OPT_FP8_MARGIN (two extra sweeps on the fp8 state); superseded by the convergence contract of   */
/* hrag_retrieve (ppr_tol / ppr_max_iters), which adds sweeps only where the measured residual asks for them    */

#define HRAG_OPT_NO_F16 1024          /* never take the fp16-state PPR either (two-stage fp16 state of batches 9..64 and of  */
                                      /* batches <= 8): the fp32 state serves.  Its values carry a relative precision of     */
                                      /* 2^-24 whatever their size, which passage scores many orders of magnitude below the */
                                      /* largest one need (a query on a slowly mixing graph that is being repeated for      */
                                      /* HRAG_FLAG_NOT_CONVERGED); runtime-switchable like HRAG_OPT_NO_FP8                    */

#define HRAG_OPT_XCD_BLOCKED 2048     /* fp8 sweep: every XCD walks a CONTIGUOUS eighth of the row order (instead of every    */
                                      /* eighth chunk group): with hrag_opts.sell_sigma on a graph with locality the rows     */
                                      /* that share in-neighbours then meet in ONE L2                                          */

#define HRAG_OPT_ACCEL 4096           /* sweep-count acceleration of the fp8-state PPR on an UNDIRECTED graph (HippoRAG's:    */
                                      /* is_directed_graph = False, config_utils.py:176): the stages run three Chebyshev      */
                                      /* steps on the real spectrum [-damping, damping] of the sweep operator instead of      */
                                      /* three Richardson sweeps (csrc/shard.hip ppr8_plan_accel).  `ppr_iters` then names an */
                                      /* ACCURACY -- the truncation error of that many plain sweeps -- and fewer sweeps run   */
                                      /* (damping 0.5, ppr_iters 20: 16; with ppr_tol > 0: 17 -- 1, 2 plain, Chebyshev stages,  */
                                      /* 2 plain at the end, so that the convergence measure reads a plain sweep's update as   */
                                      /* it does without the flag; iters_out reports them).                                   */
                                      /* The reference's PRPACK solve is tolerance-driven (HippoRAG.py:1736-1743): any sweep  */
                                      /* count that meets the tolerance is the same answer.  Guarantees: the e4m3 scale of    */
                                      /* every stage is MEASURED on the device -- each boundary reports the batch's max |R|,  */
                                      /* ppr8_next_scale_kernel maps (that maximum) x (max-norm contraction of the next       */
                                      /* stage, 7 / T_3(1 / damping)) x (growth of the iterate) to half the e4m3 range: a     */
                                      /* rigorous bound re-anchored at every stage, not a static chain.  Failure mode: a      */
                                      /* value outside the range sets HRAG_FLAG_FP8_SATURATED for its query -- clear the      */
                                      /* flag and repeat on the plain plan (the Python wrapper does), then HRAG_OPT_NO_FP8;   */
                                      /* every boundary forms the TRUE residual, so refinement stays exact                    */
                                      /* and the extension stages of the contract (plain) apply unchanged.  With ppr_tol = 0  */
                                      /* the accuracy is that of ppr_iters plain sweeps on well-mixing graphs (cfg 3: 4.4e-7  */
                                      /* against 5.5e-7) and up to 3x their truncation error on small hub-heavy ones: use the */
                                      /* contract where a bound is needed.  Off by default: ppr_iters is then the literal     */
                                      /* sweep count (BASELINE.json's 20).  Runtime-switchable.  The two-stage fp16 states    */
                                      /* (batch <= 64) accelerate too (16 sweeps for 20; 14 up to the first 0.7 builds), ppr_tol = 0 and damping <= 0.62 only.  The library */
                                      /* cannot see whether the CSR it was given came from a symmetric adjacency: setting the */
                                      /* flag on a DIRECTED graph is a caller error (complex spectrum: the steps may converge */
                                      /* more slowly than the plan assumes; the contract would flag it, ppr_tol = 0 would     */
                                      /* not) -- the Python wrapper checks row sums == column sums of A in chunks AND that    */
                                      /* 4096 sampled entries have a mirror entry of the same weight                          */
                                      /* (hipporag_amd.graph.looks_undirected, evaluated when the flag is first asked for),   */
                                      /* and refuses otherwise                                                                 */

typedef struct hrag_opts {
    int32_t max_batch;    /* largest B any call will pass (workspace is sized once)              */
    int32_t max_topk;     /* largest k_p (retrieval_top_k); <= 2048                              */
    int32_t slab_width;   /* PPR state slab width BC in {4,8,16,32,64}; 0 = auto                 */
    int32_t long_row_nnz; /* rows with more entries leave the G-lanes-per-row kernel and are cut  */
                          /* into wavefront-sized segments; 0 = auto (8 gather rounds)            */
    int32_t device;       /* HIP device ordinal, -1 = current                                    */
    int32_t flags;        /* tuning bits, 0 = defaults: HRAG_OPT_*                                  */
    int32_t segment_nnz;  /* entries per long-row segment (multiple of 64); 0 = auto (512)          */
    int32_t sell_seg_len; /* SELL-8 matrices (fp8 / fp16 / small-batch sweeps): rows longer than this are cut into */
                          /* segments whose partial sums are added in a fixed order.  0 = auto, from the work of one */
                          /* sweep of THIS engine (64 .. 2048).  The cut decides the summation order of long rows, so */
                          /* engines whose results must agree bit for bit (the row shards of a graph and the unsharded */
                          /* engine on it) have to be created with the same explicit value                            */
    int32_t sell_sigma;   /* SELL-8 matrices: sort rows by length inside windows of this many consecutive rows         */
                          /* (SELL-C-sigma) instead of globally.  0 = global sort (least padding; right for graphs       */
                          /* without locality such as the BASELINE generator).  With a vertex numbering that has         */
                          /* locality -- rows near each other share in-neighbours -- a window keeps the processing order */
                          /* close to the vertex order, so the gathered state rows are re-used from the XCD's L2; pair   */
                          /* it with HRAG_OPT_XCD_BLOCKED                                                                 */
    int32_t reserved[7];
} hrag_opts;

/* Phase timings of the last hrag_retrieve / hrag_score_facts on an engine, measured
 * with HIP events on the caller's stream (mirrors the reference's ppr_time /
 * rerank_time accumulators, HippoRAG.py:184-186).  Milliseconds. */
typedef struct hrag_timings {
    float fact_sim_ms;  /* fact GEMM + min/max + top-k           */
    float pass_sim_ms;  /* passage GEMM + min/max                */
    float seed_ms;      /* seed + teleport construction          */
    float ppr_ms;       /* all PPR iterations                    */
    float rank_ms;      /* normalise + gather + top-k            */
    float total_ms;
    int32_t ppr_iters;
    int32_t n_slabs;
    int32_t slab_width;
    int32_t n_long_rows;
} hrag_timings;

const char *hrag_last_error(void);
int hrag_version(void); /* major * 1000 + minor */

/* engine_create == prepare_retrieval_objects (HippoRAG.py:1287-1389): stage the graph,
 * the embedding matrices and the lookup arrays on the device once. `facts` / `fact_desc`
 * may be NULL for a DPR-only engine (StandardRAG). */
hrag_status hrag_engine_create(const hrag_graph_desc *graph, const hrag_embed_desc *facts,
                               const hrag_embed_desc *passages, const hrag_fact_desc *fact_desc,
                               const hrag_opts *opts, hrag_engine **out);
hrag_status hrag_engine_destroy(hrag_engine *e);

/* ------------------------------------------------------------------------------------------
 * Workspaces ("engine immutable after create -> concurrent read-only calls allowed on
 * distinct streams with distinct workspaces").  The reference serves one query at a time from one
 * Python thread (HippoRAG.py:459); a server that wants several retrieves in flight on ONE index makes
 * one workspace per stream / thread:
 *
 *   hrag_workspace_create(e, &w)   w is an engine HANDLE like e -- every entry point of this header takes
 *                                  it -- that BORROWS e's index (graph, SELL-8 matrices, embeddings, static
 *                                  tables: nothing is copied, no second 1.5 GB of embeddings) and OWNS a
 *                                  full set of per-call buffers (scores, PPR states, seeds, flags, the
 *                                  long-row arrival counters), its own entry flag, option flags
 *                                  (hrag_engine_set_flags acts on the handle it is given), events and
 *                                  timings.  Sized for e's max_batch / max_topk.
 *
 * Calls on e and on its workspaces may run CONCURRENTLY from different threads on different streams: a
 * handle rejects a second thread (HRAG_EBUSY) only for ITSELF.  Results are bit-identical to the same
 * call on e.  Rules: destroy every workspace (hrag_engine_destroy(w)) before the engine it came from
 * (hrag_engine_destroy(e) returns HRAG_EINVAL while workspaces are alive); the one entry point that
 * WRITES the index -- hrag_engine_gather_embeddings -- must not run while any other handle is in a call.
 * A workspace of a workspace is a workspace of the same root engine.
 * ------------------------------------------------------------------------------------------ */
hrag_status hrag_workspace_create(hrag_engine *e, hrag_engine **out);

/* What an engine handle holds and has done (SURVEY.md 8(b) hrag_stats).  Host-side, no device work. */
#define HRAG_PPR_STATE_F32 1    /* fp32 slabs (csrc/ppr_spmm.hip): every engine                                 */
#define HRAG_PPR_STATE_F16 2    /* two-stage fp16 state, 8 < batch <= 64 (csrc/ppr16.hip)                        */
#define HRAG_PPR_STATE_SMALL 4  /* batch <= 8 kernels (csrc/ppr_sv.hip)                                          */
#define HRAG_PPR_STATE_FP8 8    /* staged e4m3 state, batch > 64 and every row shard (csrc/ppr8.hip)             */
/* why an engine has NO staged e4m3 state -- its batches > 64 then run on the two-stage fp16 state (2 bytes per gathered
 * element instead of 1: about twice the sweep time) or, once V * 128 >= 2^32 (V >= 33.5 M), on the fp32 slabs (4 bytes);
 * same results within the same bar; hrag_shard_ppr_begin refuses such an engine.  hrag_stats.last_ppr_state says which
 * state a call ran on */
#define HRAG_FP8_UNAVAILABLE_NO_COL_SUM 1          /* hrag_graph_desc.col_sum was NULL                            */
#define HRAG_FP8_UNAVAILABLE_TOO_MANY_VERTICES 2   /* V + 1 > 2^24: the gather offsets are formed by a 24-bit     */
                                                   /* multiply ((V + 1) * 256 bytes must also stay below 2^32)   */
#define HRAG_FP8_UNAVAILABLE_SHARD_NOT_ALIGNED 4   /* owned passages != passages whose vertex is an owned row     */
#define HRAG_FP8_UNAVAILABLE_DISABLED 8            /* HRAG_OPT_F32_STATE / HRAG_OPT_NO_FP8 at creation             */
#define HRAG_FP8_UNAVAILABLE_SMALL_MAX_BATCH 16    /* unsharded engine with max_batch <= 64: never needed          */
typedef struct hrag_stats {
    int32_t is_workspace;       /* 1: the handle came from hrag_workspace_create                                  */
    int32_t live_workspaces;    /* workspaces of this engine that have not been destroyed                          */
    int64_t index_bytes;        /* device bytes of the shared half (graph, matrices, embeddings)                  */
    int64_t workspace_bytes;    /* device bytes of this handle's per-call half                                     */
    int32_t ppr_states;         /* HRAG_PPR_STATE_* the handle can run                                             */
    int32_t fp8_unavailable;    /* HRAG_FP8_UNAVAILABLE_* (0: the e4m3 state exists)                               */
    int32_t last_ppr_state;     /* HRAG_PPR_STATE_* the last hrag_retrieve / _scored on this handle ran on (0: none) */
    int32_t reserved;
    int64_t calls_score_facts, calls_retrieve, calls_dense_retrieve, calls_ppr, calls_shard;
    int64_t queries;            /* queries served by hrag_retrieve / _scored / hrag_dense_retrieve                  */
} hrag_stats;
hrag_status hrag_engine_stats(hrag_engine *e, hrag_stats *out);

/* Phase A == get_fact_scores (HippoRAG.py:1427-1465) + the candidate selection of
 * rerank_facts (:1683-1688) for B queries at once.
 *   q_fact_dev  bf16 [B, D]  "query_to_fact" embeddings, unit norm
 *   idx_out_dev int32 [B, k] fact ids (global), best first; -1 beyond min(k, F)
 *   score_out_dev fp32 [B, k] min-max normalised scores of those facts
 * The LLM filter (rerank.py:108-131) runs on the host between phase A and B. */
hrag_status hrag_score_facts(hrag_engine *e, const uint16_t *q_fact_dev, int32_t batch, int32_t k,
                             int32_t *idx_out_dev, float *score_out_dev, hrag_stream stream);

/* Phase B == graph_search_with_fact_entities (:1544-1656) + run_ppr (:1709-1749) +
 * the DPR fallback of retrieve (:467-469), for B queries at once.
 *   q_pass_dev       bf16 [B, D]  "query_to_passage" embeddings
 *   kept_idx_dev     int32 [B, kf] fact ids that survived the filter, in filter order
 *   kept_score_dev   fp32 [B, kf] their normalised scores (from phase A)
 *   kept_count_dev   int32 [B]    number of valid entries per row; 0 => DPR ranking
 *   doc_idx_out_dev  int32 [B, k] passage positions (index into passage_node_keys)
 *   doc_score_out_dev fp32 [B, k] PPR probability (or normalised DPR score on fallback)
 *   flags_out_dev    int32 [B]    bit0: DPR fallback used; bit1: reset vector had no mass
 *                                 bit2: the :1541 assert would fire (kept phrase weight 0)
 *                                 bit3: HRAG_FLAG_FP8_SATURATED -- a value of the fp8-state PPR left the e4m3
 *                                       range (a scale bound was violated; the scales are static powers of two at
 *                                       damping >= 0.46 and measured per stage below that and under
 *                                       HRAG_OPT_ACCEL): treat this query's scores as not trustworthy; rerun the
 *                                       batch after hrag_engine_set_flags(e, HRAG_OPT_NO_FP8, 1)
 *                                 bit4: HRAG_FLAG_NOT_CONVERGED -- ppr_tol > 0 and this query's residual (below)
 *                                       is still above it after ppr_max_iters sweeps: rerun with more sweeps
 *
 * Convergence contract of the PPR solve.  The reference hands the solve to PRPACK, which iterates until its
 * residual is below 1e-10 (HippoRAG.py:1736-1743); a fixed sweep count is only as accurate as the graph mixes.
 *   ppr_iters       sweeps that always run (20 in BASELINE.json)
 *   ppr_tol         0: exactly ppr_iters sweeps.  > 0: the engine measures, per query, the relative size of the
 *                   update its last sweep applied to the passage scores,
 *                       residual = damping / (1 - damping) * max over passages p of |x_p(K) - x_p(K-1)| / x_p(K)
 *                   (the error a contraction with factor `damping` has left after an update of that size), and
 *                   keeps sweeping -- decided ON THE DEVICE, no host synchronisation -- while that MEASURED residual
 *                   is above ppr_tol for some query of the batch and fewer than ppr_max_iters sweeps ran.  The
 *                   fp8-state path (batch > 64) extends in stages of 1, 2, 3, 3 sweeps up to 30 (the final sweep of a
 *                   stage runs over the passage rows only and measures; the launches of the next stage are enqueued
 *                   and skip themselves when a control word says so); the two-stage fp16 states (batch <= 64) extend
 *                   the same way by up to 9 sweeps beyond ppr_iters (stages of 1, 2, 3, 3 plain correction sweeps, each
 *                   closed by a measuring passage-row sweep); the fp32 states (other sweep counts / damping, the repeat
 *                   path) run the fixed count and report.
 *   ppr_max_iters   upper bound on the sweeps (>= ppr_iters; ignored when ppr_tol == 0)
 *   residual_out_dev fp32 [B] (may be NULL): the residual above for the sweeps that ran (0 on the DPR fallback)
 *   iters_out_dev   int32 [B] (may be NULL): sweeps that ran for the query's batch
 * The measure sees the passage rows only and is a heuristic, not a bound: where the error of a passage score is
 * fed from residual that sits on NON-passage rows (a slowly mixing graph with the seeds far from the passages: the
 * ring of tests/test_gpu_fp8_adversarial.py) it under-reads, down to 0.29 of the true error measured -- the mirror's
 * default ppr_tol = 1.5e-6 is the 1e-5 bar divided by that and by a margin of 1.9; where convergence oscillates
 * (bipartite-like graphs, the BASELINE generator) it over-reads by up to (1 + damping) / (1 - damping).  It does NOT
 * go blind on a bipartite graph: the iteration starts at x_0 = v, whose trailing term (damping P)^k v moves the
 * passage rows on every sweep, odd or even (tests/test_gpu_fp8_adversarial.py pins 20 and 21 sweeps).
 *
 * What a met tolerance bounds -- MEASURED over the adversarial suite and the randomised soaks, not proven:
 *       true relative error of every passage score  <=  max(HRAG_PPR_ERR_K * residual, floor)
 *   HRAG_PPR_ERR_K = 3.5: the worst under-reading of the measure (1 / 0.29, the ring graph);
 *   floor: what fp32 arithmetic on the state type leaves however small the measure reads -- the last sweep's update of
 *   a converged iterate rounds towards zero while the rounding of the stages before it stays (measured: a reported
 *   residual of 1.3e-8 next to a true error of 2.2e-7, 16k-vertex power-law graph, 257 queries, accelerated stages).
 *   HRAG_PPR_ERR_FLOOR_FP8 (staged e4m3 state, batch > 64, plain and HRAG_OPT_ACCEL), _F16 (two-stage fp16 states,
 *   batch <= 64), _F32 (fp32 state: other sweep counts / damping, HRAG_OPT_NO_FP8 | HRAG_OPT_NO_F16).  The floor of the
 *   e4m3 state depends on the graph (CPU emulation of the device arithmetic at 30 sweeps, where the measure reads
 *   < 1e-7: benchmark generator 4.4e-7, power-law 8.6e-7, star forest 3.8e-6); the constants are the largest values
 *   seen, rounded up.
 *   tests/test_gpu_accel.py and tests/test_gpu_fp8_adversarial.py assert the inequality query by query.
 * A tolerance below HRAG_PPR_TOL_MIN (the order of the smallest floor / K) promises nothing the arithmetic can deliver: hrag_retrieve,
 * hrag_retrieve_scored and hrag_shard_ppr_begin reject 0 < ppr_tol < HRAG_PPR_TOL_MIN with HRAG_EINVAL instead of
 * letting a caller believe it. */
#define HRAG_FLAG_NOT_CONVERGED 16
#define HRAG_PPR_ERR_K 3.5f
#define HRAG_PPR_ERR_FLOOR_FP8 5e-6f
#define HRAG_PPR_ERR_FLOOR_F16 2e-6f
#define HRAG_PPR_ERR_FLOOR_F32 5e-7f
#define HRAG_PPR_TOL_MIN 1e-7f
hrag_status hrag_retrieve(hrag_engine *e, const uint16_t *q_pass_dev, int32_t batch,
                          const int32_t *kept_idx_dev, const float *kept_score_dev,
                          const int32_t *kept_count_dev, int32_t kf, int32_t link_top_k,
                          float damping, float passage_node_weight, int32_t ppr_iters,
                          int32_t ppr_max_iters, float ppr_tol, int32_t k,
                          int32_t *doc_idx_out_dev, float *doc_score_out_dev,
                          int32_t *flags_out_dev, float *residual_out_dev, int32_t *iters_out_dev,
                          hrag_stream stream);

/* hrag_retrieve with the RAW passage scores supplied by the caller instead of computed from this engine's passage
 * embeddings: pass_scores_dev fp32 [B, pass_ld], np.dot(passage_embeddings, q) in passage order (HippoRAG.py:1496).
 * Everything else -- min-max, prior, seeds, PPR, ranking, the DPR fallback -- is hrag_retrieve's.  The PPR side of
 * the hybrid multi-GPU mode (hipporag_amd/dist.py HybridRetriever: embeddings row-sharded, the scores of a GPU's
 * queries arrive by an all-to-all, the PPR runs query-parallel on a replicated graph without any exchange).  Such an
 * engine can be created WITHOUT passage embeddings: hrag_embed_desc.data == NULL with rows = n_passages (and a fact
 * descriptor with rows == 0 next to a full hrag_fact_desc). */
hrag_status hrag_retrieve_scored(hrag_engine *e, const float *pass_scores_dev, int64_t pass_ld, int32_t batch,
                                 const int32_t *kept_idx_dev, const float *kept_score_dev,
                                 const int32_t *kept_count_dev, int32_t kf, int32_t link_top_k, float damping,
                                 float passage_node_weight, int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol,
                                 int32_t k, int32_t *doc_idx_out_dev, float *doc_score_out_dev, int32_t *flags_out_dev,
                                 float *residual_out_dev, int32_t *iters_out_dev, hrag_stream stream);

/* The scores of ALL passages of the engine's last hrag_retrieve / hrag_retrieve_scored (the array its top-k was taken
 * from: PPR probability, or the normalised DPR score on the fallback), fp32 [B, ld] in passage order.  For callers that
 * want more than max_topk = 2048 documents (HippoRAG.py:501-507 slices any prefix of the full ranking): sort these rows
 * with the library's ranking rule, np.argsort(x, kind="stable")[::-1].  Same stream discipline as every call. */
hrag_status hrag_last_doc_scores(hrag_engine *e, int32_t batch, float *out_dev, int64_t ld, hrag_stream stream);

/* == dense_passage_retrieval (HippoRAG.py:1467-1502, StandardRAG.py:393-429), top-k only. */
hrag_status hrag_dense_retrieve(hrag_engine *e, const uint16_t *q_pass_dev, int32_t batch,
                                int32_t k, int32_t *doc_idx_out_dev, float *doc_score_out_dev,
                                hrag_stream stream);

/* ---- lower-level seams (used by the per-method adapters and by the parity tests) ---- */

/* np.dot(embeddings, q.T) (HippoRAG.py:1459 / :1496): raw cosine scores.
 * which: 0 = facts, 1 = passages.  out_dev fp32 [B, rows] (row stride = rows). */
hrag_status hrag_sim_scores(hrag_engine *e, int32_t which, const uint16_t *q_dev, int32_t batch,
                            float *out_dev, hrag_stream stream);

/* run_ppr's numerical core (HippoRAG.py:1735-1743) for B reset vectors:
 *   reset_dev fp32 [B, V] (NaN / negative entries are zeroed like :1735)
 *   x_out_dev fp32 [B, V] PPR probabilities (each row sums to 1)
 *   flags_out_dev int32 [B] bit1 set when a row had no positive mass (may be NULL). */
hrag_status hrag_ppr(hrag_engine *e, const float *reset_dev, int32_t batch, float damping,
                     int32_t iters, float *x_out_dev, int32_t *flags_out_dev, hrag_stream stream);

/* Generic row-wise top-k with the library's ranking rule (np.argsort(...)[::-1][:k]) fused with
 * the row min / max.  scores_dev fp32 [B, ld] (n valid entries per row), 1 <= k <= 2048.
 *   idx_out_dev int32 [B, k] = position + idx_offset (-1 beyond min(k, n))
 *   val_out_dev fp32 [B, k]  raw score, or (s - min) / (max - min) when normalize != 0
 *   min_out_dev / max_out_dev fp32 [B] (may be NULL).  No engine needed. */
hrag_status hrag_topk_rows(const float *scores_dev, int32_t batch, int64_t n, int64_t ld, int32_t k,
                           int32_t idx_offset, int32_t normalize, int32_t *idx_out_dev,
                           float *val_out_dev, float *min_out_dev, float *max_out_dev,
                           hrag_stream stream);

/* ---- index-time entity KNN (retrieve_knn, utils/embed_utils.py:6-94; engine-less) ----
 * hrag_normalize_split_bf16: F.normalize (embed_utils.py:25,28) when normalize != 0, then the split
 *   x = hi + lo into two bf16 matrices (lo_dev may be NULL: plain bf16 rounding).
 * hrag_sim_gemm: out[b][m] (+)= sum_k q[b][k] * emb[m][k]  (torch.mm, :53); bf16 / fp16 in (dtype), fp32 out,
 *   row stride ld; accumulate != 0 adds to out.  retrieve_knn (hipporag_amd/knn.py) runs it ONCE per query block on
 *   the split layout of hrag_split_f32 (normalize != 0: F.normalize first): [hi | lo | hi] x [qhi | qhi | qlo],
 *   fp16 halves = the fp32 product to 2^-21, on the wide-batch 256-row kernel.
 * The top-k of each score row is hrag_topk_rows (k <= 2048 covers synonymy_edge_topk = 2047). */
hrag_status hrag_normalize_split_bf16(const float *x_dev, int64_t rows, int32_t dim, int32_t normalize,
                                      uint16_t *hi_dev, uint16_t *lo_dev, hrag_stream stream);
hrag_status hrag_sim_gemm(const uint16_t *emb_dev, int64_t rows, int32_t dim, const uint16_t *q_dev,
                          int32_t batch, float *out_dev, int64_t ld, int32_t accumulate, int32_t dtype,
                          hrag_stream stream);
/* The k <= 16 best rows per query WITHOUT the score matrix (the engine-less form of hrag_score_facts' fused path: the
 * GEMM epilogue keeps every 128-row tile's maximum, the k tiles with the largest maxima are rescored with the same MFMA
 * chain -- exact, bit-identical to hrag_sim_gemm + hrag_topk_rows): idx int32 [B, k], raw scores fp32 [B, k].
 * When a query's 16th best score is below the synonymy threshold, its neighbours above the threshold are all among these
 * 16 and no [B, rows] block is ever written (retrieve_knn(min_score=...) ran on this call until ABI 0.8 and runs on the
 * thresholded form below since) (embed_utils.py:53-73 materialises
 * it block by block).  workspace_dev: hrag_sim_topk_workspace_bytes(rows, batch) bytes, ZEROED ONCE by the caller
 * before the first call (the call leaves it reusable). */
int64_t hrag_sim_topk_workspace_bytes(int64_t rows, int32_t batch);
hrag_status hrag_sim_topk(const uint16_t *emb_dev, int64_t rows, int32_t dim, const uint16_t *q_dev, int32_t batch,
                          int32_t k, int32_t dtype, void *workspace_dev, int64_t workspace_bytes,
                          int32_t *idx_out_dev, float *val_out_dev, hrag_stream stream);
/* The THRESHOLDED form (ABI 0.8; add_synonymy_edges reads neighbours down to synonymy_edge_sim_threshold only,
 * HippoRAG.py:1004-1007): the caller trusts results at or above min_score and nothing below it.
 *   - tiles whose maximum is below min_score - margin are never rescored;
 *   - approx_dim > 0: pass 1 (the tile maxima) runs over the first approx_dim elements of every row and query only; the
 *     caller states in `margin` a bound on |full product - prefix product|.  For the split layout of hrag_split_f32 with
 *     approx_dim = dim / 3 the prefix is hi . qhi and the rest, lo . qhi + hi . qlo, is at most 2 * 2^-11 |x| |q| = 9.8e-4
 *     for unit vectors: margin 1.2e-3, a third of the MFMA work.  approx_dim = 0: pass 1 over all of dim (margin may be 0);
 *   - pass 3 rescores the tiles that reach the cut over all `dim` elements: every returned score is the exact chain of
 *     hrag_sim_gemm, and every row with score >= min_score is returned (in score order, ahead of anything below
 *     min_score) UNLESS overflow_out_dev[b] != 0 (more than k tiles reached the cut) or the k-th returned score is itself
 *     >= min_score: for those queries the caller takes hrag_sim_gemm + hrag_topk_rows.  Entries below min_score are
 *     NOT the global ranking (rows of unselected tiles are missing among them).
 * k <= 16; workspace as hrag_sim_topk; overflow_out_dev int32 [B]. */
hrag_status hrag_sim_topk_min_score(const uint16_t *emb_dev, int64_t rows, int32_t dim, const uint16_t *q_dev, int32_t batch,
                                    int32_t k, int32_t dtype, int32_t approx_dim, float min_score, float margin,
                                    void *workspace_dev, int64_t workspace_bytes, int32_t *idx_out_dev,
                                    float *val_out_dev, int32_t *overflow_out_dev, hrag_stream stream);
/* fp32 [rows, dim] -> the fp16 [rows, 3 * dim] layout of an HRAG_F32_SPLIT engine: [hi | lo | hi] for embedding rows,
 * [hi | hi | lo] with as_query != 0 (new rows for hrag_engine_gather_embeddings; the engine converts its own inputs);
 * normalize != 0: rows are L2-normalised first (x / max(||x||, 1e-12), the KNN's F.normalize). */
hrag_status hrag_split_f32(const float *x_dev, int64_t rows, int32_t dim, int32_t as_query, int32_t normalize,
                           uint16_t *out_dev, hrag_stream stream);

/* Measurement hook: run `n` PPR SpMM sweeps over the engine's current state buffers for
 * `batch` right-hand sides (state is whatever the last hrag_retrieve / hrag_ppr left).
 * flags bit0: fp32 CSR path: main kernel only (skip its long-row and seed kernels); no effect on the SELL-8 kernels
 *             (bits 1-3), which finish long rows inside the sweep kernel;
 * flags bit1: the fp16-state kernel of the two-stage scheme (mode H) instead of the fp32 one
 *             (HRAG_EINVAL when the engine has no fp16 state: sharded, max_batch <= 8, F32_STATE);
 * flags bit2: the small-batch kernel (batch <= 8, state fp32 [V][1|2|4|8]);
 * flags bit3: the fp8-state kernel over the buffers of the last hrag_retrieve (HRAG_EINVAL without col_sum /
 *             max_batch <= 64 / batch <= 64 / no preceding hrag_retrieve); bits 4-5 pick the instantiation
 *             (0 stage sweep C, 1 boundary B, 2 final F, 3 first boundary B0), bits 6-7 the residual form of
 *             B / F (0 fp32, 2 fp32 in / 3-byte out, 3 3-byte in / out; F: 1 = 3-byte in);
 *             with bit 3, bit 8 (256) = GATHER REPLAY: the state-row gathers of a stage sweep and nothing else
 *             (same matrix, same state, same launch geometry; nothing is written) -- the floor of any sweep that
 *             fetches one state row per matrix slot (bench.py roofline.gather_replay_ms). */
hrag_status hrag_ppr_sweeps(hrag_engine *e, int32_t batch, int32_t n, float damping, int32_t flags,
                            hrag_stream stream);

/* ------------------------------------------------------------------------------------------
 * Stage-level operators.  hrag_retrieve() is exactly the composition of these on engine-owned
 * buffers; they are exported so that a host can interleave its own exchange steps (the
 * multi-GPU row-shard mode of hipporag_amd/dist.py all-gathers x between hrag_stage_ppr_step
 * calls) -- every buffer is caller-owned device memory.
 *
 * PPR state layout ("slab layout"): x is [n_slabs][V][bc] fp32; query q lives in slab q / bc,
 * column q % bc; hrag_ppr_layout() returns (bc, n_slabs) for a batch on this engine.
 * seed arrays: seed_vtx int32 [B][HRAG_SEED_STRIDE], seed_w fp32 [B][HRAG_SEED_STRIDE], seed_cnt int32 [B].
 * ------------------------------------------------------------------------------------------ */
#define HRAG_SEED_STRIDE 32

hrag_status hrag_ppr_layout(hrag_engine *e, int32_t batch, int32_t *slab_width_out,
                            int32_t *n_slabs_out);

/* min / max of every row (what min_max_normalize reduces, misc_utils.py:131-132). */
hrag_status hrag_row_minmax(const float *scores_dev, int32_t batch, int64_t n, int64_t ld,
                            float *min_out_dev, float *max_out_dev, hrag_stream stream);

/* seeds == HippoRAG.py:1574-1623 + :1505-1542 on fact ids (needs the engine's fact_desc).
 * flags_dev int32 [B] is read-modify-written (bits 0 and 2). */
hrag_status hrag_stage_seeds(hrag_engine *e, const int32_t *kept_idx_dev, const float *kept_score_dev,
                             const int32_t *kept_count_dev, int32_t kf, int32_t link_top_k,
                             int32_t batch, int32_t *seed_vtx_dev, float *seed_w_dev,
                             int32_t *seed_cnt_dev, int32_t *flags_dev, hrag_stream stream);

/* passage prior == HippoRAG.py:1626-1635: tele[slab][p][c] = minmax(S[q][p]) * passage_node_weight
 * for ALL Np passages (scores_dev fp32 [B, ld] raw cosine scores in passage order, min/max per
 * row); rows of queries whose flags bit0 is set (DPR fallback) are zero.
 * tele_out_dev: [n_slabs][Np][bc]. */
hrag_status hrag_stage_teleport(hrag_engine *e, const float *scores_dev, int64_t ld,
                                const float *min_dev, const float *max_dev, float passage_node_weight,
                                const int32_t *flags_dev, int32_t batch, float *tele_out_dev,
                                hrag_stream stream);

/* x0 = v on the rows this engine owns (rows [row_offset, row_offset + n_rows) of every slab). */
hrag_status hrag_stage_ppr_init(hrag_engine *e, const float *tele_dev, const int32_t *seed_vtx_dev,
                                const float *seed_w_dev, const int32_t *seed_cnt_dev, int32_t batch,
                                float *x_dev, hrag_stream stream);

/* one sweep y = damping * P x + (1 - damping) * v on the owned rows (reads all of x). */
hrag_status hrag_stage_ppr_step(hrag_engine *e, const float *tele_dev, const int32_t *seed_vtx_dev,
                                const float *seed_w_dev, const int32_t *seed_cnt_dev, int32_t batch,
                                float damping, const float *x_dev, float *y_dev, hrag_stream stream);

/* per-query sum of x over the owned rows -> sums_out_dev double [B] (all-reduce it across shards).
 * workspace_dev: hrag_colsum_workspace_bytes(e, batch) bytes. */
hrag_status hrag_stage_colsum(hrag_engine *e, const float *x_dev, int32_t batch, void *workspace_dev,
                              double *sums_out_dev, hrag_stream stream);
int64_t hrag_colsum_workspace_bytes(hrag_engine *e, int32_t batch);

/* doc_scores == pagerank_scores[passage_node_idxs] / sum (HippoRAG.py:1745) for ALL Np passages
 * from a complete x; queries with flags bit0 get minmax(scores_dev) (DPR fallback, :467-469);
 * sets flags bit1 where sums <= 0 on a non-fallback query.  out_dev fp32 [B, out_ld]. */
hrag_status hrag_stage_doc_scores(hrag_engine *e, const float *x_dev, const double *sums_dev,
                                  int32_t batch, const float *scores_dev, int64_t ld,
                                  const float *min_dev, const float *max_dev, int32_t *flags_dev,
                                  float *out_dev, int64_t out_ld, hrag_stream stream);

/* ------------------------------------------------------------------------------------------
 * Row-sharded hot path (multi-GPU): one engine per GPU, each created with the row shard
 * [row_offset, row_offset + n_rows) of the graph, the passage embeddings of the passages whose
 * vertex lies in that range (hrag_embed_desc.row_offset / rows must describe exactly those: the
 * passage prior then never crosses GPUs) and any contiguous slice of the fact embeddings.
 * The reference has no analogue (HippoRAG.py:459 is a serial loop over queries); this is the layout
 * BASELINE.json's north star names.  The PPR iterate is the staged e4m3 state of csrc/ppr8.hip,
 * replicated; after every sweep the host exchanges the owners' row blocks (1 byte per vertex and
 * query on the wire) -- hipporag_amd/dist.py does it with one RCCL all-gather per exchange group.
 * Needs hrag_graph_desc.col_sum; damping^ppr_iters <= 2^-20, 16 <= ppr_iters <= 30.
 *
 * State buffers (caller-owned, three of them, zero-initialised once): e4m3
 *   [n_groups][num_vertices + 1][slabs_per_group][128]; query q lives in slab q / 128; slab s in
 *   group s / slabs_per_group.  Row num_vertices of every group stays zero.  The owned rows of one
 *   group are ONE contiguous block (own_offset, own_bytes inside the group): with equal-sized shards
 *   in rank order (own_offset == rank * own_bytes) an in-place all-gather over the first
 *   world * own_bytes bytes of the group completes the buffer.
 *
 * Per batch:  hrag_shard_score_facts -> gather + merge candidates, min / max all-reduce
 *             hrag_shard_passage_scores -> all-reduce MIN / MAX of mn / mx
 *             hrag_stage_seeds (replicated)
 *             hrag_shard_prior_stats -> all-reduce MAX of zmax, SUM of mass
 *             hrag_shard_ppr_begin -> exchange every group of state[0]
 *             n_steps x n_groups x hrag_shard_ppr_sweep -> exchange group g of state[*exchange_out]
 *                 (n_steps = ppr_iters without a tolerance; measuring steps: est all-reduce + hrag_shard_ppr_decide)
 *             hrag_shard_finish -> gather + merge the local top-k lists
 * ------------------------------------------------------------------------------------------ */
#define HRAG_FLAG_FP8_SATURATED 8

typedef struct hrag_shard_layout {
    int32_t n_slabs;         /* 128-query slabs of the batch                                   */
    int32_t n_groups;        /* exchange groups                                                */
    int32_t slabs_per_group;
    int32_t reserved;
    int64_t state_bytes;     /* size of ONE state buffer = n_groups * group_bytes               */
    int64_t group_bytes;     /* (num_vertices + 1) * slabs_per_group * 128                      */
    int64_t own_offset;      /* byte offset of the owned rows inside a group                    */
    int64_t own_bytes;       /* n_rows * slabs_per_group * 128                                  */
} hrag_shard_layout;

/* want_groups: 0 = the narrowest groups; otherwise the number of exchange groups asked for.  The engine may
 * return a different number: a group is limited to 4 GiB and 2^24 vertices, and its width is kept EVEN
 * (the sweep kernels gather the two adjacent slabs of a vertex together), so the narrowest group holds two slabs. */
hrag_status hrag_shard_layout_query(hrag_engine *e, int32_t batch, int32_t want_groups,
                                    hrag_shard_layout *out);

/* local phase A: the k best facts of the owned fact rows per query, RAW cosine scores (global fact
 * ids, -1 beyond the shard's size) + the local min / max of every score row. */
hrag_status hrag_shard_score_facts(hrag_engine *e, const uint16_t *q_fact_dev, int32_t batch, int32_t k,
                                   int32_t *idx_out_dev, float *score_out_dev, float *min_out_dev,
                                   float *max_out_dev, hrag_stream stream);

/* cosine scores of the owned passages (kept inside the engine) + their local min / max fp32 [B]. */
hrag_status hrag_shard_passage_scores(hrag_engine *e, const uint16_t *q_pass_dev, int32_t batch,
                                      float *min_out_dev, float *max_out_dev, hrag_stream stream);

/* given the GLOBAL min / max: zmax_out_dev fp32 [B] (max over the owned passages of the normalised
 * score / weighted degree) and mass_out_dev double [2 * B] (prior mass, prior mass on isolated
 * passages) -- all-reduce MAX / SUM them.  flags_dev: bit0 queries carry no prior. */
hrag_status hrag_shard_prior_stats(hrag_engine *e, const float *min_dev, const float *max_dev,
                                   float passage_node_weight, const int32_t *flags_dev, int32_t batch,
                                   float *zmax_out_dev, double *mass_out_dev, hrag_stream stream);

/* reset vector on the owned rows + c_0 into the owned rows of state[0] (all groups). */
hrag_status hrag_shard_ppr_begin(hrag_engine *e, const float *min_dev, const float *max_dev,
                                 const float *zmax_dev, const double *mass_dev, float passage_node_weight,
                                 const int32_t *seed_vtx_dev, const float *seed_w_dev,
                                 const int32_t *seed_cnt_dev, int32_t *flags_dev, int32_t batch,
                                 float damping, int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol,
                                 int32_t n_groups, void *state0_dev, void *state1_dev, void *state2_dev,
                                 int32_t *n_steps_out, hrag_stream stream);

/* sweep `sweep` (0 .. ppr_iters - 1, in order per group) on the owned rows of exchange group `group`;
 * *exchange_out = index of the state buffer whose owned block of that group was written and must be
 * exchanged before the group's next sweep (-1 after the last sweep: nothing to exchange).
 * Issue the groups of one sweep in ASCENDING order (0 .. n_groups - 1) before any group of the next sweep: on an
 * engine that owns every row and measures its stage scales (damping < 0.46, HRAG_OPT_ACCEL) the boundary sweeps fold
 * each group's maximum into one running value and the launch of the last group turns it into the next scale. */
hrag_status hrag_shard_ppr_sweep(hrag_engine *e, int32_t sweep, int32_t group, int32_t *exchange_out,
                                 int32_t *checkpoint_out, hrag_stream stream);

/* The convergence contract of hrag_retrieve on row shards (ppr_tol > 0 in hrag_shard_ppr_begin; *n_steps_out then
 * counts the conditional steps too: run ALL of them, the gate words decide on the device which ones do anything --
 * and every shard decides alike because the measure is all-reduced):
 *   after a step whose *checkpoint_out was set (a final sweep that measures; all groups swept):
 *        hrag_shard_ppr_est(e, est, 0) -> all-reduce MAX of est fp32 [B] over the shards ->
 *        hrag_shard_ppr_est(e, est, 1) -> hrag_shard_ppr_decide(e, step)
 *   before hrag_shard_finish: the same get / all-reduce MAX / set once more (a session without extension stages has no
 *   measuring step), so that residual_out, iters_out and flags bit 4 come out identical on every shard. */
hrag_status hrag_shard_ppr_est(hrag_engine *e, float *est_dev, int32_t set, hrag_stream stream);
hrag_status hrag_shard_ppr_decide(hrag_engine *e, int32_t sweep, hrag_stream stream);
/* *open_out = 1 when step `step` of the session will do something (no gate, or its gate word says go), 0 when a
 * decision has closed its gate (every later step is then closed too) or `step` is past the last one.  The ONLY shard
 * entry point that synchronises the stream: a host loop that pays a collective per step calls it once after each
 * hrag_shard_ppr_decide and stops issuing steps -- and their exchanges -- when it returns 0. */
hrag_status hrag_shard_ppr_gate(hrag_engine *e, int32_t step, int32_t *open_out, hrag_stream stream);

/* doc scores of the owned passages (PPR probability, or the normalised DPR score on the fallback)
 * and their local top-k: idx_out_dev int32 [B, k] GLOBAL passage positions, score_out_dev fp32 [B, k].
 * flags_dev is read-modify-written (bit 1). */
hrag_status hrag_shard_finish(hrag_engine *e, const float *min_dev, const float *max_dev, int32_t *flags_dev,
                              int32_t batch, int32_t k, int32_t *idx_out_dev, float *score_out_dev,
                              float *residual_out_dev, int32_t *iters_out_dev, hrag_stream stream);

/* ------------------------------------------------------------------------------------------
 * ONE CALL PER PHASE on a row shard (round 6).  The steps above leave the order of the local steps, the exchanges and the
 * all-reduces to the host (hipporag_amd/dist.py ShardedRetriever is that host loop, ~100 lines per phase).  The two
 * drivers below run exactly that loop INSIDE the library and call the host back only for the collectives, so that a
 * multi-GPU host -- one process per GPU, one engine per process -- needs four small functions (RCCL one-liners on the
 * given stream) and two calls.  Results are bit-identical to the host loop (tests/test_gpu_multi.py, two processes).
 *
 * hrag_comm: every function works IN PLACE on DEVICE memory, must be ordered after the work already enqueued on `stream`
 * and before work enqueued on it later (an RCCL call on that stream is; a host-staged implementation synchronises it),
 * and returns 0 on success (anything else aborts the driver with HRAG_EINVAL).  world == 1: the pointers may be NULL.
 *   all_reduce      buf_dev: `count` elements of dtype HRAG_COMM_F32 / _F64 / _I32, op HRAG_COMM_MIN / _MAX / _SUM
 *   all_gather      recv_dev = world blocks of bytes_per_rank in rank order; block `rank` = send_dev
 *   exchange_begin  the per-sweep state exchange of one exchange group: region_dev = world blocks of own_bytes, block
 *                   `rank` holds this shard's fresh rows; start the all-gather in place (it may run asynchronously on a
 *                   stream of the host's as long as it is ordered behind `stream` at the time of the call) ...
 *   exchange_wait   ... and make `stream` wait for the exchange of `group` (every block then holds its owner's rows).
 *                   At most one exchange per group is in flight; group g's exchange overlaps the sweeps of the others.
 *                   When a driver returns an error it first calls exchange_wait for every exchange it had begun (rc
 *                   ignored), so the host's collective handles are never left open behind a failed call.
 * ------------------------------------------------------------------------------------------ */
#define HRAG_COMM_F32 0
#define HRAG_COMM_F64 1
#define HRAG_COMM_I32 2
#define HRAG_COMM_MIN 0
#define HRAG_COMM_MAX 1
#define HRAG_COMM_SUM 2
typedef struct hrag_comm {
    void *user;             /* passed back to every function */
    int32_t rank, world;
    int32_t (*all_reduce)(void *user, void *buf_dev, int64_t count, int32_t dtype, int32_t op, hrag_stream stream);
    int32_t (*all_gather)(void *user, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, hrag_stream stream);
    int32_t (*exchange_begin)(void *user, void *region_dev, int64_t own_bytes, int32_t group, hrag_stream stream);
    int32_t (*exchange_wait)(void *user, int32_t group, hrag_stream stream);
} hrag_comm;

/* bytes of caller-owned device scratch the two drivers need for `batch` queries and lists of k entries */
int64_t hrag_shard_workspace_bytes(hrag_engine *e, int32_t world, int32_t batch, int32_t k);

/* phase A over all shards: hrag_shard_score_facts + all-gather of the candidates + MIN / MAX all-reduce + the merge:
 * idx_out_dev int32 [B, k] global fact ids (-1 beyond), score_out_dev fp32 [B, k] min-max normalised with the GLOBAL
 * row minimum / maximum -- replicated on every shard, equal to hrag_score_facts of the unsharded engine. */
hrag_status hrag_shard_score_facts_all(hrag_engine *e, const hrag_comm *comm, const uint16_t *q_fact_dev, int32_t batch,
                                       int32_t k, void *workspace_dev, int64_t workspace_bytes, int32_t *idx_out_dev,
                                       float *score_out_dev, hrag_stream stream);

/* phase B over all shards = hrag_retrieve of the unsharded engine (same arguments, same outputs, replicated on every
 * shard): passage scores, statistics all-reduced, hrag_shard_ppr_begin, every sweep of every exchange group with its
 * exchange, the contract's all-reduced measure and decisions (ppr_tol > 0), hrag_shard_finish, the merged top-k.
 * state0..2_dev: three state buffers of hrag_shard_layout_query(e, batch, n_groups).state_bytes, zeroed ONCE by the
 * caller (row V of every group must stay zero) and reusable across calls; flags_out_dev int32 [B] also carries bit 3
 * (HRAG_FLAG_FP8_SATURATED) of EVERY shard; residual_out_dev / iters_out_dev may be NULL (written when ppr_tol > 0). */
hrag_status hrag_shard_retrieve(hrag_engine *e, const hrag_comm *comm, const uint16_t *q_pass_dev, int32_t batch,
                                const int32_t *kept_idx_dev, const float *kept_score_dev, const int32_t *kept_count_dev,
                                int32_t kf, int32_t link_top_k, float damping, float passage_node_weight,
                                int32_t ppr_iters, int32_t ppr_max_iters, float ppr_tol, int32_t k, int32_t n_groups,
                                void *state0_dev, void *state1_dev, void *state2_dev, void *workspace_dev,
                                int64_t workspace_bytes, int32_t *doc_idx_out_dev, float *doc_score_out_dev,
                                int32_t *flags_out_dev, float *residual_out_dev, int32_t *iters_out_dev,
                                hrag_stream stream);

/* Index update with the embeddings staying on the device (incremental index() / delete(), HippoRAG.py:262-411:
 * the reference re-reads everything from its stores in prepare_retrieval_objects): compose the embedding
 * matrix of the NEXT engine from the rows this engine already holds and the rows that are new,
 *   out[i] = src_rows_dev[i] >= 0 ? held row src_rows_dev[i] : new_rows_dev[-src_rows_dev[i] - 1],
 * which: 0 = facts, 1 = passages; rows are dim 16-bit elements (3 * dim on an HRAG_F32_SPLIT engine: new rows in the
 * layout of hrag_split_f32); out_dev [n, dim] is caller-owned and can be
 * handed to hrag_engine_create as hrag_embed_desc.data (device pointers are accepted there). */
hrag_status hrag_engine_gather_embeddings(hrag_engine *e, int32_t which, const int32_t *src_rows_dev, int64_t n,
                                          const void *new_rows_dev, void *out_dev, hrag_stream stream);

/* set / clear HRAG_OPT_* tuning bits after creation (e.g. HRAG_OPT_NO_FP8 to rerun a batch that
 * reported HRAG_FLAG_FP8_SATURATED on the fp16 / fp32 state). */
hrag_status hrag_engine_set_flags(hrag_engine *e, int32_t flags, int32_t on);

hrag_status hrag_get_timings(hrag_engine *e, hrag_timings *out); /* synchronises the events */
hrag_status hrag_set_profiling(hrag_engine *e, int32_t enabled);

#ifdef __cplusplus
}
#endif
#endif /* HRAG_H_ */
