"""Shared test scaffolding: small synthetic indexes in both the oracle's and the engine's form."""

from __future__ import annotations

import numpy as np

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float


from oracle.checks import (  # noqa: E402,F401  (the checkers live with the oracle: bench.py and smoke() use them too)
    ID_GAP_FLOOR, make_case, prior_noise_allowance, ranked_parity, tie_aware_equal, tie_aware_report, ulp4_report,
)


def write_test_report(name: str, record: dict) -> None:
    """Leave a small JSON record of a GPU test's parity figures under gpurun_out/test_reports/ (merged back from the
    GPU box) and on stdout (pytest -s); never fails the test."""
    import json
    import os
    print(f"[{name}] {json.dumps(record)}")
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out", "test_reports")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(record, f)
    except OSError:
        pass
