"""Index-time entity KNN: hipporag_amd.knn.retrieve_knn against an fp64 restatement of
reference src/hipporag/utils/embed_utils.py:6-94 (normalize -> mm -> topk)."""

import numpy as np
import pytest

from tests.helpers import tie_aware_equal

pytestmark = pytest.mark.gpu


def _ref_knn(q, keys, k):
    qn = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    kn = keys / np.maximum(np.linalg.norm(keys, axis=1, keepdims=True), 1e-12)
    s = qn.astype(np.float64) @ kn.astype(np.float64).T
    order = np.argsort(s, axis=1, kind="stable")[:, ::-1][:, :k]
    return order, np.take_along_axis(s, order, axis=1)


@pytest.mark.parametrize("nq,nk,dim,k,qb", [(37, 3000, 128, 50, 16), (5, 4100, 768, 2047, 1000), (3, 7, 64, 2047, 2),
                                            (130, 2500, 200, 10, 64)])
def test_retrieve_knn_matches_fp64(gpu_device, nq, nk, dim, k, qb):
    from hipporag_amd.knn import retrieve_knn
    rng = np.random.default_rng(nq + nk)
    keys = rng.standard_normal((nk, dim)).astype(np.float32) * rng.uniform(0.1, 10, (nk, 1)).astype(np.float32)
    q = (keys[rng.integers(0, nk, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    idx, sc = retrieve_knn([f"q{i}" for i in range(nq)], [f"k{i}" for i in range(nk)], q, keys, k=k,
                           query_batch_size=qb, return_arrays=True)
    want_idx, want_sc = _ref_knn(q, keys, k)
    assert idx.shape == want_idx.shape
    for i in range(nq):
        assert tie_aware_equal(idx[i], want_idx[i], want_sc[i], abs_gap=4e-6), i
        got_true = _ref_knn(q[i:i + 1], keys, nk)      # true score of the returned ids
        full = np.empty(nk); full[got_true[0][0]] = got_true[1][0]
        np.testing.assert_allclose(sc[i], full[idx[i]], rtol=0, atol=3e-6)
    # dict form, like the reference returns it
    res = retrieve_knn([f"q{i}" for i in range(2)], [f"k{i}" for i in range(nk)], q[:2], keys, k=min(k, 5),
                       query_batch_size=qb)
    ids0, sc0 = res["q0"]
    assert ids0 == [f"k{j}" for j in idx[0][:len(ids0)]] and np.allclose(sc0, sc[0][:len(sc0)])


def test_retrieve_knn_single_pass_bf16_is_coarser(gpu_device):
    from hipporag_amd.knn import retrieve_knn
    rng = np.random.default_rng(1)
    keys = rng.standard_normal((2000, 256)).astype(np.float32)
    q = rng.standard_normal((8, 256)).astype(np.float32)
    _, sc3 = retrieve_knn(None, None, q, keys, k=20, return_arrays=True)
    _, sc1 = retrieve_knn(None, None, q, keys, k=20, precision="bf16", return_arrays=True)
    _, want = _ref_knn(q, keys, 20)
    assert np.abs(sc3 - want).max() < 3e-6 < np.abs(sc1 - want).max() < 1e-2
    assert retrieve_knn([], [], np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32)) == {}


def test_synonymy_candidates_rules(gpu_device):
    """HippoRAG.py:992-1018: threshold, self-match and short-phrase skips."""
    from hipporag_amd.knn import synonymy_candidates
    rng = np.random.default_rng(2)
    base = rng.standard_normal((6, 64)).astype(np.float32)
    embs = np.concatenate([base, base[:3] + 0.01 * rng.standard_normal((3, 64)).astype(np.float32)])
    texts = ["alpha", "beta", "ab", "delta", "epsilon", "zeta", "alpha2", "beta2", "ab2"]
    keys = [f"entity-{i}" for i in range(9)]
    edges = synonymy_candidates(keys, texts, embs, topk=8, sim_threshold=0.8)
    pairs = {(a, b) for a, b, _ in edges}
    assert ("entity-0", "entity-6") in pairs and ("entity-6", "entity-0") in pairs
    assert ("entity-1", "entity-7") in pairs
    assert not any(a == "entity-2" for a, _, _ in edges)          # "ab": <= 2 alphanumerics -> skipped as a query
    assert ("entity-8", "entity-2") in pairs                       # ... but still a valid neighbour
    assert all(a != b and s >= 0.8 for a, b, s in edges)


@pytest.mark.parametrize("n_dup", [3, 40])
def test_thresholded_knn_matches_the_full_lists(gpu_device, n_dup):
    """retrieve_knn(min_score=t): exactly the prefix of every full top-k list that lies above t -- through the fused
    top-16 (no score block) when a query has fewer than 16 such neighbours (n_dup = 3), through the dense path for the
    queries that have more (n_dup = 40: clusters of 40 near-duplicates)."""
    from hipporag_amd.knn import retrieve_knn
    rng = np.random.default_rng(7)
    base = rng.standard_normal((60, 96)).astype(np.float32)
    keys = np.concatenate([base] + [base[:20] + 0.02 * rng.standard_normal((20, 96)).astype(np.float32) for _ in range(n_dup)])
    keys = np.concatenate([keys, rng.standard_normal((3000, 96)).astype(np.float32)])
    q = keys[:120]
    full_i, full_s = retrieve_knn(None, None, q, keys, k=64, query_batch_size=50, return_arrays=True)
    thr_i, thr_s = retrieve_knn(None, None, q, keys, k=64, query_batch_size=50, return_arrays=True, min_score=0.8)
    assert (full_s[:20] >= 0.8).sum(1).min() >= min(n_dup, 64) - 1         # the duplicated rows really have that many
    for r in range(len(q)):
        n = int((full_s[r] >= 0.8).sum())
        np.testing.assert_array_equal(thr_i[r, :n], full_i[r, :n])
        np.testing.assert_array_equal(thr_s[r, :n], full_s[r, :n])
        assert np.all(thr_i[r, n:] == -1) and np.all(thr_s[r, n:] == 0)


def test_index_from_openie_computes_its_synonymy_edges_on_the_gpu(gpu_device):
    """HippoRAG.index_from_openie(synonymy="knn"): add_synonymy_edges (:959-1020) as part of the call -- the KNN over all
    entities the store holds afterwards on the GPU (knn.synonymy_candidates) -- gives the graph the same call gives with
    the edges computed by an fp64 KNN under the reference's selection rules and passed through synonymy=<callable>
    (the contract tools/soak_incremental_vs_reference.py checks against the real reference on the CPU)."""
    import re
    from hipporag_amd.retriever import HippoRAG, RetrievalConfig
    from tests.golden.make_golden import MockEmbeddingModel

    def cpu_candidates(keys, texts, embs, *, topk=2047, sim_threshold=0.8):
        e = np.asarray(embs, np.float64)
        e = e / np.linalg.norm(e, axis=1, keepdims=True)
        s = e @ e.T
        out = []
        for i, t in enumerate(texts):
            if len(re.sub("[^A-Za-z0-9]", "", t)) <= 2:
                continue
            n = 0
            for j in np.argsort(-s[i], kind="stable")[:topk]:
                if s[i, j] < sim_threshold or n > 100:
                    break
                if j != i and texts[j] != "":
                    out.append((keys[i], keys[j], float(s[i, j])))
                    n += 1
        return out

    names = [f"{a} {b} works {i % 7}" for i, (a, b) in enumerate(zip("alpha beta gamma delta kappa sigma omega theta".split() * 5,
                                                                     "river stone cloud field".split() * 10))]
    names += [n + " group" for n in names[:12]]                      # near-duplicates: cosine >= 0.8 under the mock model
    docs = [f"document {d} about {names[d % len(names)]}" for d in range(30)]
    triples = [[(names[(3 * d) % len(names)], "relates to", names[(5 * d + 1) % len(names)]),
                (names[(7 * d + 2) % len(names)], "mentions", names[(11 * d + 3) % len(names)])] for d in range(30)]
    graphs = []
    for how in ("knn", cpu_candidates):
        rag = HippoRAG(RetrievalConfig(max_batch=4, embedding_precision="bf16"), embedding_model=MockEmbeddingModel())
        rag.index_from_openie(docs[:20], triples[:20], synonymy=how)
        rag.index_from_openie(docs[15:], triples[15:], synonymy=how)             # incremental: old pairs gain parallel edges
        g = rag._graph
        s, d, w = g.edge_list()
        tot = {}
        for a, b, x in zip(s.tolist(), d.tolist(), w.tolist()):
            k = tuple(sorted((g.names[a], g.names[b])))
            tot[k] = tot.get(k, 0.0) + float(x)
        graphs.append(tot)
    syn = {k: v for k, v in graphs[1].items() if k[0].startswith("entity-") and k[1].startswith("entity-") and v != round(v)}
    assert len(syn) >= 12                                             # the near-duplicate names are linked
    assert set(graphs[0]) == set(graphs[1])
    for k in graphs[1]:
        assert abs(graphs[0][k] - graphs[1][k]) <= 4e-6 * max(1.0, abs(graphs[1][k])), k


def test_thresholded_knn_prefix_first_pass_straddling_the_threshold(gpu_device):
    """retrieve_knn(min_score=t) at a shape where the thresholded fused top-16 runs its first pass over the hi . qhi third
    of the split layout (include/hrag.h hrag_sim_topk_min_score: 256-row GEMM, dim % 64 == 0, batch > 64): neighbours
    planted with cosines within 5e-3 of the threshold on BOTH sides -- inside and outside the 1.2e-3 margin of the prefix
    pass -- scattered over many 128-row tiles; queries with more than 16 qualifying tiles (the overflow flag) and with
    more than 16 qualifying rows (the dense path).  Every prefix above the threshold must equal the full lists'."""
    from hipporag_amd.knn import retrieve_knn
    rng = np.random.default_rng(11)
    dim, n_base, n_fill = 128, 200, 12000

    def unit(x):
        return x / np.linalg.norm(x, axis=-1, keepdims=True)

    base = unit(rng.standard_normal((n_base, dim)))
    rows = [base]
    for i in range(n_base):
        n_dup = (3, 10, 24, 40)[i % 4]                       # 24 / 40: more than 16 rows (and tiles) above the threshold
        u = unit(rng.standard_normal((n_dup, dim)))
        u = unit(u - (u @ base[i])[:, None] * base[i])       # orthogonal to the base vector
        c = rng.uniform(0.795, 0.805, n_dup)[:, None]        # cosines straddling 0.8
        rows.append(c * base[i] + np.sqrt(1 - c * c) * u)
    keys = np.concatenate(rows + [unit(rng.standard_normal((n_fill, dim)))]).astype(np.float32)
    perm = rng.permutation(len(keys))                        # the neighbours of a query land in many tiles
    keys = keys[perm]
    q = keys[np.argsort(perm)[:n_base]]                      # the base vectors, wherever they went
    full_i, full_s = retrieve_knn(None, None, q, keys, k=64, query_batch_size=100, return_arrays=True)
    thr_i, thr_s = retrieve_knn(None, None, q, keys, k=64, query_batch_size=100, return_arrays=True, min_score=0.8)
    n_above = (full_s >= 0.8).sum(1)
    assert n_above.min() >= 1 and n_above.max() > 16 and ((n_above > 1) & (n_above < 16)).sum() > 20
    near = np.abs(full_s - 0.8) < 1.2e-3                     # scores inside the prefix pass's margin exist on both sides
    assert (near & (full_s >= 0.8)).any() and (near & (full_s < 0.8)).any()
    for r in range(len(q)):
        n = int(n_above[r])
        np.testing.assert_array_equal(thr_i[r, :n], full_i[r, :n])
        np.testing.assert_array_equal(thr_s[r, :n], full_s[r, :n])
        assert np.all(thr_i[r, n:] == -1) and np.all(thr_s[r, n:] == 0)
