"""`python bench.py --gpus N` is its own launcher (hipporag_amd/launch.py): the driver runs exactly that command, with
no torchrun in front.  CPU tests: the launcher with a stub worker (environment, rank 0's JSON line last, a dying rank
ends the job), bench.py itself on this GPU-less box (must fail in seconds with ONE device-count line, before any rank starts), and the
choice of the leg that becomes `value` at N > 1."""

import io
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

from hipporag_amd import launch
from bench_dist import pick_value_leg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    p = tmp_path / "worker.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_spawn_ranks_gives_every_rank_the_torchrun_environment_and_prints_rank0s_line_last(tmp_path):
    cmd = _stub(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        print(f"chatter from rank {r}")                      # must never end up after the JSON line
        if r == 0:
            print(json.dumps({"metric": "stub", "world": w, "port": os.environ["MASTER_PORT"]}))
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(3, cmd, out=out, err=err, timeout_s=60)
    assert rc == 0
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["metric"] == "stub" and rec["world"] == 3
    for r in range(3):
        assert f"chatter from rank {r}" in err.getvalue()


def test_spawn_ranks_a_dying_rank_ends_the_job_with_its_exit_code(tmp_path):
    cmd = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit("rank 1: needs 2 devices, 0 visible")   # SystemExit(str) -> exit code 1, message on stderr
        time.sleep(600)                                       # the survivors would hang in a collective
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(2, cmd, out=out, err=err, timeout_s=120)
    assert rc == 1
    assert "rank 1 exited with code 1" in err.getvalue()
    assert out.getvalue() == ""


def test_spawn_ranks_times_out(tmp_path):
    cmd = _stub(tmp_path, "import time; time.sleep(600)")
    out, err = io.StringIO(), io.StringIO()
    assert launch.spawn_ranks(2, cmd, out=out, err=err, timeout_s=1.0) == 124


def _bench_gpus_2(extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HRAG_FORCE_DIST")}
    env.update(extra_env or {})
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--config", "tiny"], env=env, capture_output=True, text=True, timeout=600)
    return p, time.monotonic() - t0


def test_bench_gpus_2_fails_fast_with_one_clear_line_where_there_are_fewer_gpus():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs (this container: none; a 1-GPU box: see the gpu test
    below): ONE clear line on stderr, a non-zero exit code, nothing on stdout, in seconds -- the device count is taken
    through ctypes before any rank (or torch) is started (round-5 review, item 4b)."""
    p, dt = _bench_gpus_2()
    assert p.returncode != 0
    assert "needs 2 GPUs on this node" in p.stderr, p.stderr[-2000:]
    assert len([ln for ln in p.stderr.splitlines() if ln.strip()]) == 1, p.stderr[-2000:]
    assert "torch.distributed.run" not in p.stderr
    assert p.stdout.strip() == ""
    assert dt < 10.0, dt


@pytest.mark.gpu
def test_bench_gpus_2_fails_fast_on_a_one_gpu_box():
    """The same command on the GPU box the driver runs the suite on: with exactly one device visible it must end the
    same way (skipped where 2+ devices are visible: there the command is a real run)."""
    from hipporag_amd.launch import visible_gpu_count
    if (visible_gpu_count() or 0) >= 2:
        pytest.skip("2+ GPUs visible: `bench.py --gpus 2` is a real run here")
    p, dt = _bench_gpus_2()
    assert p.returncode != 0 and "needs 2 GPUs on this node" in p.stderr and p.stdout.strip() == "", (p.returncode, p.stderr[-1000:])
    assert dt < 10.0, dt


def test_the_ranks_still_check_the_device_count_themselves_under_an_external_launcher():
    """With WORLD_SIZE set by a launcher the pre-check is not taken; rank 0 must refuse with the same message."""
    env = {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(launch.free_port())}
    p, _ = _bench_gpus_2(env)
    assert p.returncode != 0 and "needs 2 GPUs on this node" in p.stderr, p.stderr[-2000:]


def test_value_leg_prefers_a_parity_green_corpus_sharded_leg():
    green = {"value": 1.0, "ms_per_step": 1.0, "parity": {"ok": True}}
    red = {"value": 9.0, "ms_per_step": 1.0, "parity": {"ok": False}}
    failed = {"error": "boom"}
    assert pick_value_leg("auto", green, green) == "rowshard"       # the north star's layout is the primary figure
    assert pick_value_leg("auto", red, green) == "rowshard"
    assert pick_value_leg("auto", green, red) == "hybrid"
    assert pick_value_leg("hybrid", green, green) == "hybrid"
    assert pick_value_leg("auto", failed, red) == "replica"
    assert pick_value_leg("auto", None, {"skipped": True}) == "replica"
    assert pick_value_leg("rowshard", green, green) == "rowshard"
    assert pick_value_leg("hybrid", red, green) == "replica"
    assert pick_value_leg("replica", green, green) == "replica"


def test_a_stopped_launcher_takes_its_ranks_along(tmp_path):
    """SIGTERM to the launcher (a driver-side timeout) must not leave N orphaned ranks holding the GPUs."""
    import signal
    import time
    pidfile = tmp_path / "pids"
    worker = tmp_path / "worker.py"
    worker.write_text(textwrap.dedent(f"""
        import os, time
        open({str(pidfile)!r}, "a").write(str(os.getpid()) + "\\n")
        time.sleep(600)
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from hipporag_amd import launch
        sys.exit(launch.spawn_ranks(2, [sys.executable, {str(worker)!r}]))
    """))
    p = subprocess.Popen([sys.executable, str(driver)], stderr=subprocess.DEVNULL)
    for _ in range(200):
        if pidfile.exists() and len(pidfile.read_text().split()) == 2:
            break
        time.sleep(0.05)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    time.sleep(0.3)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, pid
