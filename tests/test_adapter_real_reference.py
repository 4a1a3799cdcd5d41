"""reference_adapter.attach() / detach() against the REAL reference class (CPU, authoring container only:
/root/reference does not travel).  See tests/support/adapter_on_real_reference.py."""

import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/hipporag"), reason="reference sources not present")
def test_attach_and_detach_on_the_real_reference_object(tmp_path):
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "support", "adapter_on_real_reference.py")], env=env,
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ADAPTER_ON_REAL_REFERENCE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/hipporag"), reason="reference sources not present")
def test_the_mirror_class_against_the_reference_on_fresh_random_corpora():
    """A short run of tools/soak_mirror_vs_reference.py (500 corpora in the round's record): the mirror's retrieve /
    retrieve_dpr / retrieve_ircot -- host logic over the oracle-backed engine stand-in -- against the real reference's
    own methods, the IRCoT reasoner replaced by the same deterministic function on both sides."""
    env = dict(os.environ, PYTHONHASHSEED="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_mirror_vs_reference.py"), "--cases", "5", "--seed", "9"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "5 cases; SOAK OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
