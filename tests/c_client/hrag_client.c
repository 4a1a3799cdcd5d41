/* A reference-side host in PLAIN C: no Python, no torch -- include/hrag.h, libhrag.so and the HIP runtime only.
 * What a maintainer who binds the library from another language sees.  Reads one binary case file
 * (written by tests/test_gpu_c_client.py), stages the index with hrag_engine_create, runs
 *   phase A  hrag_score_facts                          (HippoRAG.get_fact_scores + rerank_facts candidates, :1427-1465, :1683-1688)
 *   phase B  hrag_retrieve under the contract          (graph_search_with_fact_entities + run_ppr, :1544-1656, :1709-1749)
 *   the same phase B on a second WORKSPACE handle      (hrag_workspace_create)
 *   phase A + B through the one-call row-shard drivers at world 1 (hrag_shard_score_facts_all / hrag_shard_retrieve, NULL callbacks)
 * and writes every result to the output file; the test compares them with the Python wrapper's, bit for bit.
 *   gcc -std=c11 hrag_client.c -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -L hipporag_amd -lhrag -L /opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hrag.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define HR(x) do { hrag_status s_ = (x); if (s_ != HRAG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, (int)s_, hrag_last_error()); return 3; } } while (0)

static void *slurp(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); }
    return p;
}
static void *to_dev(const void *h, size_t bytes) {
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) exit(5);
    return d;
}
static void dump(FILE *o, const void *dev, size_t bytes) {
    void *h = malloc(bytes);
    if (hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) exit(6);
    fwrite(h, 1, bytes, o);
    free(h);
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hd[8];   /* V, nnz, Np, F, D, B, k_docs, k_facts */
    if (fread(hd, 8, 8, f) != 8) return 1;
    const int64_t V = hd[0], nnz = hd[1], Np = hd[2], F = hd[3], D = hd[4], B = hd[5], K = hd[6], KF = hd[7];
    int32_t *row_ptr = slurp(f, (size_t)(V + 1) * 4), *col = slurp(f, (size_t)nnz * 4);
    float *val = slurp(f, (size_t)nnz * 4);
    double *col_sum = slurp(f, (size_t)V * 8);
    int32_t *pv = slurp(f, (size_t)Np * 4);
    uint16_t *pemb = slurp(f, (size_t)Np * D * 2), *femb = slurp(f, (size_t)F * D * 2);
    int32_t *subj = slurp(f, (size_t)F * 4), *obj = slurp(f, (size_t)F * 4), *nch = slurp(f, (size_t)V * 4);
    uint16_t *qf = slurp(f, (size_t)B * D * 2), *qp = slurp(f, (size_t)B * D * 2);
    fclose(f);

    hrag_graph_desc g = {V, 0, V, nnz, row_ptr, col, val, Np, pv, col_sum};
    hrag_embed_desc pe = {Np, 0, (int32_t)D, HRAG_BF16, pemb}, fe = {F, 0, (int32_t)D, HRAG_BF16, femb};
    hrag_fact_desc fd = {F, subj, obj, nch};
    hrag_opts op;
    memset(&op, 0, sizeof op);
    op.max_batch = (int32_t)B; op.max_topk = (int32_t)K; op.device = -1;
    hrag_engine *e = NULL, *w = NULL;
    HR(hrag_engine_create(&g, &fe, &pe, &fd, &op, &e));          /* host arrays are copied: the caller may free them now */
    free(row_ptr); free(col); free(val);

    hipStream_t s;
    HIP(hipStreamCreate(&s));
    uint16_t *d_qf = to_dev(qf, (size_t)B * D * 2), *d_qp = to_dev(qp, (size_t)B * D * 2);
    int32_t *f_idx, *cnt, *d_idx, *flags, *used;
    float *f_sc, *d_sc, *resid;
    HIP(hipMalloc((void **)&f_idx, (size_t)B * KF * 4)); HIP(hipMalloc((void **)&f_sc, (size_t)B * KF * 4));
    HIP(hipMalloc((void **)&d_idx, (size_t)B * K * 4)); HIP(hipMalloc((void **)&d_sc, (size_t)B * K * 4));
    HIP(hipMalloc((void **)&flags, (size_t)B * 4)); HIP(hipMalloc((void **)&used, (size_t)B * 4)); HIP(hipMalloc((void **)&resid, (size_t)B * 4));
    int32_t *h_cnt = malloc((size_t)B * 4);
    for (int64_t i = 0; i < B; ++i) h_cnt[i] = (int32_t)KF;      /* the identity "recognition memory" filter: all candidates kept */
    cnt = to_dev(h_cnt, (size_t)B * 4);

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 1;
    /* ---- phase A, phase B on the engine */
    HR(hrag_score_facts(e, d_qf, (int32_t)B, (int32_t)KF, f_idx, f_sc, s));
    HR(hrag_retrieve(e, d_qp, (int32_t)B, f_idx, f_sc, cnt, (int32_t)KF, (int32_t)KF, 0.5f, 0.05f, 20, 29, 1.5e-6f, (int32_t)K,
                     d_idx, d_sc, flags, resid, used, s));
    HIP(hipStreamSynchronize(s));
    dump(o, f_idx, (size_t)B * KF * 4); dump(o, f_sc, (size_t)B * KF * 4);
    dump(o, d_idx, (size_t)B * K * 4); dump(o, d_sc, (size_t)B * K * 4); dump(o, flags, (size_t)B * 4);
    dump(o, resid, (size_t)B * 4); dump(o, used, (size_t)B * 4);
    /* ---- the same phase B on a second workspace of the same index */
    HR(hrag_workspace_create(e, &w));
    HIP(hipMemset(d_idx, 0xff, (size_t)B * K * 4));
    HR(hrag_retrieve(w, d_qp, (int32_t)B, f_idx, f_sc, cnt, (int32_t)KF, (int32_t)KF, 0.5f, 0.05f, 20, 29, 1.5e-6f, (int32_t)K,
                     d_idx, d_sc, flags, resid, used, s));
    HIP(hipStreamSynchronize(s));
    dump(o, d_idx, (size_t)B * K * 4); dump(o, d_sc, (size_t)B * K * 4);
    hrag_stats st;
    HR(hrag_engine_stats(w, &st));
    if (!st.is_workspace || st.calls_retrieve != 1 || st.index_bytes <= 0) { fprintf(stderr, "unexpected hrag_stats\n"); return 7; }
    if (hrag_engine_destroy(e) == HRAG_OK) { fprintf(stderr, "an engine with a live workspace must not be destroyed\n"); return 8; }
    HR(hrag_engine_destroy(w));
    /* ---- the one-call row-shard drivers at world 1 (this engine owns every row; no collective is called) */
    if (B > 64) {
        hrag_comm cm;
        memset(&cm, 0, sizeof cm);
        cm.rank = 0; cm.world = 1;
        hrag_shard_layout lay;
        HR(hrag_shard_layout_query(e, (int32_t)B, 2, &lay));
        void *st8[3], *ws;
        for (int i = 0; i < 3; ++i) { HIP(hipMalloc(&st8[i], (size_t)lay.state_bytes)); HIP(hipMemset(st8[i], 0, (size_t)lay.state_bytes)); }
        const int64_t wsb = hrag_shard_workspace_bytes(e, 1, (int32_t)B, (int32_t)K);
        HIP(hipMalloc(&ws, (size_t)wsb));
        HR(hrag_shard_score_facts_all(e, &cm, d_qf, (int32_t)B, (int32_t)KF, ws, wsb, f_idx, f_sc, s));
        HR(hrag_shard_retrieve(e, &cm, d_qp, (int32_t)B, f_idx, f_sc, cnt, (int32_t)KF, (int32_t)KF, 0.5f, 0.05f, 20, 29, 1.5e-6f,
                               (int32_t)K, lay.n_groups, st8[0], st8[1], st8[2], ws, wsb, d_idx, d_sc, flags, resid, used, s));
        HIP(hipStreamSynchronize(s));
        dump(o, f_idx, (size_t)B * KF * 4); dump(o, f_sc, (size_t)B * KF * 4);
        dump(o, d_idx, (size_t)B * K * 4); dump(o, d_sc, (size_t)B * K * 4);
    }
    fclose(o);
    HR(hrag_engine_destroy(e));
    printf("hrag_client OK: version %d, %lld queries\n", hrag_version(), (long long)B);
    return 0;
}
