"""Seeded slices of the randomised differential soaks (tools/soak_random.py, tools/soak_shards.py) inside the driver-run
suite (round-4 review, next #8): the soaks found four real bugs in round 4 but only ever ran when the builder started
them.  A FIXED number of seeded cases each (the same cases on every box; ~25 s): random small indices x batch sizes x damping x plain / accelerated x fixed count / convergence contract
x filter outcomes x embedding kinds x engine knobs through hrag_score_facts / hrag_retrieve / hrag_ppr, and random
shard counts x exchange groups through the row-shard entry points, every case against the oracle.  A failing case
prints its parameters: `python tools/soak_random.py --replay '<json>'` replays it."""

import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_seeded_slice_of_the_random_soak(gpu_device, capsys):
    out = os.path.join(ROOT, "gpurun_out", "test_reports", "soak_random_slice.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    rc = _tool("soak_random").main(["--seconds", "150", "--cases", "160", "--seed", "50501", "--out", out])
    text = capsys.readouterr().out
    assert rc == 0 and "SOAK OK" in text, text[-3000:]


def test_seeded_slice_of_the_shard_soak(gpu_device, capsys):
    out = os.path.join(ROOT, "gpurun_out", "test_reports", "soak_shards_slice.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    rc = _tool("soak_shards").main(["--seconds", "150", "--cases", "200", "--seed", "50502", "--out", out])
    text = capsys.readouterr().out
    assert rc == 0 and "SOAK OK" in text, text[-3000:]
