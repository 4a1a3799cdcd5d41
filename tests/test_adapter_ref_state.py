"""reference_adapter.attach() with the REAL engine on REAL reference state: the object a real ``hipporag.HippoRAG``
was after its own index() + prepare_retrieval_objects() (recorded in the authoring container, restored by
tests/restored_reference.py), answering the queries the reference's own retrieve() answered
(src/hipporag/HippoRAG.py:413-499) -- fp32 embeddings as the stores hold them, mixed LLM-filter outcomes incl. the DPR
fallback.  (tests/test_adapter_real_reference.py covers the live class on the CPU with an oracle-backed engine.)"""
import numpy as np
import pytest

from tests.helpers import tie_aware_equal
from tests.restored_reference import restore



def _check_attach_on_restored_state():
    from hipporag_amd.reference_adapter import attach, detach
    rag, st = restore("synth_f32")
    queries = st["queries"]
    text_to_pos = {st["chunk_rows"][k]["content"]: i for i, k in enumerate(st["passage_node_keys"])}
    attach(rag, max_batch=16)            # defaults: fp32-faithful similarity, convergence contract at 3e-6
    assert rag._mi355x is not None
    sols = rag.retrieve(list(queries))
    assert len(sols) == len(queries)
    n_dpr = 0
    for q, (sol, want) in enumerate(zip(sols, st["reference_retrieve"])):
        assert sol.question == queries[q]
        want_ids = [text_to_pos[d] for d in want["docs"]]
        got_ids = [text_to_pos[d] for d in sol.docs]
        assert len(got_ids) == len(want_ids)
        ws = np.asarray(want["doc_scores"], np.float64)
        is_dpr = ws.max() == 1.0 and ws.min() == 0.0          # min-max normalised DPR ranking (HippoRAG.py:467-469)
        n_dpr += int(is_dpr)
        if is_dpr:
            assert tie_aware_equal(got_ids, want_ids, ws, abs_gap=4e-6), q
            np.testing.assert_allclose(sol.doc_scores, ws, rtol=0, atol=4e-6)
        else:
            assert tie_aware_equal(got_ids, want_ids, ws, rel_gap=2e-5), q
            by_pos = dict(zip(want_ids, ws))
            ref = np.array([by_pos[i] for i in got_ids])
            # the 1e-5 bar where the normalised prior is well conditioned; passages whose whole score is a prior of
            # cancelled fp32 similarities carry the similarity's 3e-6 absolute error (tests/test_ref_golden.py synth_f32)
            got = np.asarray(sol.doc_scores, np.float64)
            pos = ref > 0                                        # an isolated passage whose prior is the row minimum: 0
            assert np.all(got[~pos] == 0.0)
            rel = np.abs(got[pos] - ref[pos]) / ref[pos]
            assert np.median(rel) < 1e-5 and rel.max() < 2e-4, (q, rel.max())
    assert n_dpr == 3                                            # the recorded run's DPR fallbacks
    # per-query seams on the same object
    fs = rag.get_fact_scores(queries[0])
    assert fs.shape == (len(st["fact_node_keys"]),) and fs.max() == 1.0 and fs.min() == 0.0
    d_ids, d_sc = rag.dense_passage_retrieval(queries[3])
    want = st["reference_retrieve"][3]                           # query 3: the filter kept nothing -> DPR ranking
    assert tie_aware_equal(d_ids[: len(want["docs"])], [text_to_pos[d] for d in want["docs"]],
                           np.asarray(want["doc_scores"]), abs_gap=4e-6)
    detach(rag)
    assert rag._mi355x is None


@pytest.mark.gpu
def test_attach_on_restored_reference_state_answers_like_the_reference(gpu_device):
    _check_attach_on_restored_state()


def test_restored_reference_state_is_coherent_on_the_cpu(monkeypatch):
    """The same check with the engine replaced by the oracle-backed stand-in of tests/support: the restored object
    carries everything attach() reads, and the recorded answers are reachable from it (no GPU, no reference package)."""
    import importlib.util
    import os
    from hipporag_amd import engine as engine_mod
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "adapter_on_real_reference.py")
    spec = importlib.util.spec_from_file_location("adapter_on_real_reference", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(engine_mod, "HippoRAGEngine", mod.OracleEngine)
    _check_attach_on_restored_state()
