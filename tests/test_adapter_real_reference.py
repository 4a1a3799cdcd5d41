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
