"""CPU oracle for the HippoRAG retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy/scipy (+ one small C file) restatement of the
reference's algorithm for the path

    fact scores -> top-k facts -> (filter) -> seed/reset vector ->
    dense passage scores -> Personalized PageRank -> ranked passage ids

Every function cites the reference file:line it follows (paths relative to
the upstream repository root).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker / baseline*, never as
the thing measured or shipped.  ``hipporag_amd`` (the product) must never
import it; the product path fails loudly when the HIP library is missing.

PARITY STATUS.  The reference has no test, golden vector or known-answer fixture for
any function on this path.  What pins this oracle instead:

  * PINNED against the reference's own code, run in the authoring container: the fixtures
    ``tests/golden/ref_*.npz`` were produced by importing ``/root/reference/src/hipporag``
    (``tests/golden/ref_harness.py`` / ``make_ref_golden.py``; real: index(), the edge rules,
    prepare_retrieval_objects, get_fact_scores, rerank_facts, graph_search_with_fact_entities,
    get_top_k_weights, dense_passage_retrieval, run_ppr's wrapper, retrieve, retrieve_dpr;
    substituted: igraph, the LLM calls, the embedding model).  ``tests/test_ref_golden.py``
    checks every stage of this oracle against those recordings (fact scores, candidate
    facts, seed weights, reset vectors bit-for-bit, DPR ranking, final ranking).
  * UNPINNED: the PPR arithmetic itself.  It lives in ``python_igraph==0.11.8`` (PRPACK),
    which is neither vendored in /root/reference nor installable here, so the stand-in
    igraph of the harness solves with this oracle's own PRPACK restatement.  That part is
    cross-checked against independent formulations instead (``networkx.pagerank`` --
    networkx 3.4.2 is the version the reference pins --, a sparse direct solve of
    (I - alpha P) x = v, and the C port on both sides of PRPACK's 128-vertex solver switch),
    see ``tests/test_oracle_ppr.py``.
"""

from .hipporag_ref import (  # noqa: F401
    min_max_normalize,
    fact_scores,
    dense_passage_scores,
    topk_desc,
    rerank_facts,
    seed_weights,
    reset_vector,
    run_ppr,
    retrieve_one,
    retrieve_dpr_one,
    RefIndex,
)
from .ppr import (  # noqa: F401
    build_symmetric_csr,
    column_normalize,
    ppr_exact,
    ppr_power,
)
