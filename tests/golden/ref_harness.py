"""Runs the REAL reference (OSU-NLP-Group/HippoRAG, /root/reference/src/hipporag) in the authoring
container so that its own code produces golden vectors for the retrieval hot path.

TEST INFRASTRUCTURE, generator side only: this module needs /root/reference and is never imported
on the GPU box (tests read the committed .npz fixtures; a CPU test re-runs it when the reference is
present to prove the fixtures are reproducible).

What is real and what is substituted
  real (imported unmodified from /root/reference/src):  HippoRAG.__init__, index(), add_fact_edges,
      add_passage_edges, add_synonymy_edges (+ utils/embed_utils.retrieve_knn on torch CPU),
      augment_graph, prepare_retrieval_objects, get_query_embeddings, get_fact_scores, rerank_facts,
      graph_search_with_fact_entities, get_top_k_weights, dense_passage_retrieval, run_ppr, retrieve,
      retrieve_dpr, EmbeddingStore (parquet), min_max_normalize, compute_mdhash_id, text_processing ...
  substituted (absent from this image, no network):
      igraph        -> the small in-memory multigraph below; its personalized_pagerank() solves with
                       oracle/prpack_port.c (a restatement of PRPACK, python_igraph==0.11.8 is where
                       the reference gets it) -- so PRPACK's ARITHMETIC is the one thing these
                       fixtures do not pin; everything around it is the reference's own code;
      OpenIE (LLM)  -> hand-written triples;  DSPyFilter (LLM) -> identity filter;
      embedding model -> deterministic mock (tests/golden/make_golden.py recipe);
      openai / litellm / tenacity / boto3 / botocore / gritlm / sentence_transformers -> empty stubs
      (imported at module top by files that are not on this path).
"""

from __future__ import annotations

import importlib.machinery
import os
import pickle
import sys
import types

import numpy as np

REFERENCE_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STUBBED = ["openai", "tenacity", "litellm", "boto3", "botocore", "botocore.auth", "botocore.awsrequest",
           "botocore.exceptions", "gritlm", "sentence_transformers"]


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "hipporag"))


class _Passthrough:
    """Stand-in for any class / decorator factory of a stubbed package."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]                     # used as a decorator: leave the function alone
        return self


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        t = type(name, (_Passthrough,), {})
        setattr(self, name, t)
        return t


# ----------------------------------------------------------------------------- igraph stand-in
class _Vertex:
    def __init__(self, g, i):
        self._g, self.index = g, i

    def __getitem__(self, attr):
        return self._g._vattr[attr][self.index]

    def attributes(self):
        return {k: v[self.index] for k, v in self._g._vattr.items()}


class _VertexSeq:
    def __init__(self, g):
        self._g = g

    def __call__(self):
        return self

    def __len__(self):
        return self._g._n

    def __iter__(self):
        return (_Vertex(self._g, i) for i in range(self._g._n))

    def __getitem__(self, key):
        if isinstance(key, str):
            return list(self._g._vattr[key])
        return _Vertex(self._g, range(self._g._n)[key])

    def attribute_names(self):
        return list(self._g._vattr)


class _EdgeSeq:
    def __init__(self, g):
        self._g = g

    def __call__(self):
        return self

    def __len__(self):
        return len(self._g._edges)

    def __getitem__(self, key):
        return list(self._g._eattr[key])

    def attribute_names(self):
        return list(self._g._eattr)


class Graph:
    """The subset of igraph.Graph the reference touches (HippoRAG.py:233-236, 408, 887-888, 1169-1229,
    1304-1329, 1577-1579, 1736-1743).  Undirected multigraph, vertex / edge attributes as columns."""

    def __init__(self, directed=False):
        assert not directed, "the reference builds an undirected graph (config_utils.py is_directed_graph=False)"
        self._n = 0
        self._vattr = {}
        self._edges = []
        self._eattr = {}
        self._solver = None

    # construction ---------------------------------------------------------------------------------
    def add_vertices(self, n, attributes=None):
        attributes = attributes or {}
        for k in set(self._vattr) | set(attributes):
            col = self._vattr.setdefault(k, [None] * self._n)
            vals = attributes.get(k)
            col.extend(list(vals) if vals is not None else [None] * n)
            assert len(col) == self._n + n
        self._n += n
        self._solver = None

    def add_edges(self, es, attributes=None):
        attributes = attributes or {}
        name_to_idx = None
        new = []
        for u, v in es:
            if isinstance(u, str) or isinstance(v, str):      # igraph resolves vertex names
                if name_to_idx is None:
                    name_to_idx = {nm: i for i, nm in enumerate(self._vattr["name"])}
                u, v = name_to_idx[u], name_to_idx[v]
            new.append((int(u), int(v)))
        for k in set(self._eattr) | set(attributes):
            col = self._eattr.setdefault(k, [None] * len(self._edges))
            vals = attributes.get(k)
            col.extend(list(vals) if vals is not None else [None] * len(new))
        self._edges.extend(new)
        self._solver = None

    def delete_vertices(self, ids):
        name_to_idx = {nm: i for i, nm in enumerate(self._vattr.get("name", []))}
        drop = {name_to_idx[i] if isinstance(i, str) else int(i) for i in ids}
        keep = [i for i in range(self._n) if i not in drop]
        remap = {old: new for new, old in enumerate(keep)}
        self._vattr = {k: [v[i] for i in keep] for k, v in self._vattr.items()}
        kept_e = [j for j, (u, v) in enumerate(self._edges) if u in remap and v in remap]
        self._eattr = {k: [v[j] for j in kept_e] for k, v in self._eattr.items()}
        self._edges = [(remap[self._edges[j][0]], remap[self._edges[j][1]]) for j in kept_e]
        self._n = len(keep)
        self._solver = None

    # queries --------------------------------------------------------------------------------------
    @property
    def vs(self):
        return _VertexSeq(self)

    @property
    def es(self):
        return _EdgeSeq(self)

    def vcount(self):
        return self._n

    def ecount(self):
        return len(self._edges)

    def get_edgelist(self):
        return list(self._edges)

    def is_directed(self):
        return False

    def write_pickle(self, fname):
        with open(fname, "wb") as f:
            pickle.dump((self._n, self._vattr, self._edges, self._eattr), f)

    @classmethod
    def Read_Pickle(cls, fname):
        g = cls()
        with open(fname, "rb") as f:
            g._n, g._vattr, g._edges, g._eattr = pickle.load(f)
        return g

    # the one numerical entry point (HippoRAG.py:1736-1743) ------------------------------------------
    def personalized_pagerank(self, vertices=None, directed=True, damping=0.85, reset=None, weights=None,
                              implementation="prpack"):
        import oracle
        from oracle.prpack_port import PrpackCSR
        assert implementation == "prpack" and directed is False
        if self._solver is None:
            src = [u for u, _ in self._edges]
            dst = [v for _, v in self._edges]
            w = self._eattr[weights] if isinstance(weights, str) else (weights or [1.0] * len(src))
            a = oracle.build_symmetric_csr(self._n, src, dst, w)
            self._solver = PrpackCSR(oracle.column_normalize(a))
        reset = np.asarray(reset, dtype=np.float64)
        if reset.shape != (self._n,) or np.isnan(reset).any() or (reset < 0).any() or not reset.sum() > 0:
            raise ValueError("igraph: invalid reset vector")
        x, _ = self._solver.solve(reset, float(damping), "prpack")
        idx = list(range(self._n)) if vertices is None else list(vertices)
        return [float(x[i]) for i in idx]


def install_stubs() -> None:
    for name in STUBBED:
        if name not in sys.modules:
            m = _StubModule(name)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
            sys.modules[name] = m
    if "igraph" not in sys.modules:
        ig = types.ModuleType("igraph")
        ig.Graph = Graph
        ig.__spec__ = importlib.machinery.ModuleSpec("igraph", None)
        sys.modules["igraph"] = ig


def import_reference():
    """import hipporag from /root/reference/src with the stubs above."""
    if not reference_available():
        raise RuntimeError("reference sources not present at " + REFERENCE_SRC)
    install_stubs()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import hipporag  # noqa: F401
    return hipporag


# ----------------------------------------------------------------------------- substituted LLM parts
class FixedOpenIE:
    """Replaces information_extraction.OpenIE.batch_openie (LLM) with hand-written triples."""

    def __init__(self, doc_to_triples):
        self._t = doc_to_triples

    def batch_openie(self, chunks):
        from hipporag.utils.misc_utils import NerRawOutput, TripleRawOutput
        ner, tri = {}, {}
        for key, row in chunks.items():
            triples = [list(t) for t in self._t[row["content"]]]
            ents = list(dict.fromkeys(e for t in triples for e in (t[0], t[2])))
            ner[key] = NerRawOutput(chunk_id=key, response=None, unique_entities=ents, metadata={})
            tri[key] = TripleRawOutput(chunk_id=key, response=None, triples=triples, metadata={})
        return ner, tri


def identity_filter(query, candidate_items, candidate_indices, len_after_rerank=None):
    """DSPyFilter.__call__ (rerank.py:108-131) with an LLM that keeps every candidate."""
    return candidate_indices, candidate_items, {"confidence": None}


class _NoLLM:
    def infer(self, *a, **k):
        raise RuntimeError("no LLM in the golden harness")


def build_reference_rag(save_dir, docs, triples, embedding_model, **config_overrides):
    """A real reference HippoRAG object, indexed with the reference's own index()."""
    import_reference()
    from hipporag import HippoRAG
    from hipporag.utils.config_utils import BaseConfig
    cfg = BaseConfig()
    cfg.save_dir = save_dir
    cfg.openie_mode = "online"
    cfg.force_index_from_scratch = True
    cfg.force_openie_from_scratch = True
    cfg.rerank_dspy_file_path = None
    for k, v in config_overrides.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    rag = HippoRAG(global_config=cfg, extraction_llm=_NoLLM(), qa_llm=_NoLLM(), embedding_model=embedding_model)
    rag.openie = FixedOpenIE({d: t for d, t in zip(docs, triples)})
    rag.rerank_filter = identity_filter
    rag.index(docs)
    return rag


def capture(rag, queries, num_to_retrieve=None):
    """Runs the reference's retrieve() and records, per query, what each stage of the hot path produced
    (by wrapping the reference's own methods -- nothing is recomputed here)."""
    log = {"fact_scores": [], "reset": [], "ppr_ids": [], "ppr_scores": [], "dpr_ids": [], "dpr_scores": [],
           "top_fact_idx": []}
    orig_ppr, orig_dpr, orig_fs, orig_rr = rag.run_ppr, rag.dense_passage_retrieval, rag.get_fact_scores, rag.rerank_facts

    def run_ppr(reset_prob, damping=0.5):
        ids, sc = orig_ppr(reset_prob, damping=damping)
        log["reset"].append(np.array(reset_prob, dtype=np.float64))
        log["ppr_ids"].append(np.array(ids)); log["ppr_scores"].append(np.array(sc, dtype=np.float64))
        return ids, sc

    def dpr(query):
        ids, sc = orig_dpr(query)
        log["dpr_ids"].append(np.array(ids)); log["dpr_scores"].append(np.array(sc))
        return ids, sc

    def fs(query):
        s = orig_fs(query)
        log["fact_scores"].append(np.array(s))
        return s

    def rr(query, scores):
        out = orig_rr(query, scores)
        log["top_fact_idx"].append(np.array(out[0], dtype=np.int64))
        return out

    rag.run_ppr, rag.dense_passage_retrieval, rag.get_fact_scores, rag.rerank_facts = run_ppr, dpr, fs, rr
    try:
        sols = rag.retrieve(list(queries), num_to_retrieve=num_to_retrieve)
    finally:
        rag.run_ppr, rag.dense_passage_retrieval, rag.get_fact_scores, rag.rerank_facts = orig_ppr, orig_dpr, orig_fs, orig_rr
    return sols, log
