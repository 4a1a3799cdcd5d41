"""One process per GPU without an external launcher: `python bench.py --gpus N` spawns its own N ranks.

The reference has no distributed code (src/hipporag/HippoRAG.py:459 is a serial loop over the queries); this is
the process model of every multi-GPU mode of hipporag_amd/dist.py.  `python -m torch.distributed.run` sets RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* and starts the script N times; a plain `python bench.py --gpus N` does the same
here: spawn_ranks() re-executes the command N times with that environment (rendezvous on 127.0.0.1 -- the container
hostname may not resolve), keeps rank 0's standard output, forwards everything else to stderr, and prints rank 0's
LAST line (the JSON line of bench.py) as the last line of its own stdout.  A rank that dies takes the others with it
(they are ended by PID, never by pattern) and its exit code becomes the launcher's.
"""

from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """The environment torch.distributed.run would give rank `rank` of a one-node job."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HRAG_SELF_SPAWNED="1")
    # the host driver only supports dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _pump(stream, sink: List[str], echo) -> None:
    for line in iter(stream.readline, ""):
        sink.append(line)
        if echo is not None:
            echo.write(line)
            echo.flush()
    stream.close()


def _pump_rank0(stream, held: List[str], echo) -> None:
    """Rank 0: every non-empty line is forwarded to `echo` AS IT ARRIVES except the most recent one, which is held back
    (held[0]) -- it may be the JSON line that belongs on stdout.  Nothing accumulates, and a long run shows its
    progress while it runs."""
    for line in iter(stream.readline, ""):
        if not line.strip():
            continue
        if held:
            echo.write(held[0])
            echo.flush()
            held[0] = line
        else:
            held.append(line)
    stream.close()


def spawn_ranks(world: int, cmd: Sequence[str], *, timeout_s: Optional[float] = None, port: Optional[int] = None,
                env: Optional[Dict[str, str]] = None, out=None, err=None) -> int:
    """Run `cmd` as `world` ranks of one node; returns the job's exit code (0 = every rank exited 0).

    Rank 0's last non-empty stdout line is written to `out` (default sys.stdout) when the job ends; its earlier lines
    and every other rank's stdout go to `err` (default sys.stderr) as they arrive (rank 0's with a lag of one line:
    the most recent line is held back, it may be the last).  All ranks share
    this process' stderr.  timeout_s: end the job (exit code 124) if it has not finished by then."""
    if world < 1:
        raise ValueError("world must be >= 1")
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port() if port is None else port
    procs: List[subprocess.Popen] = []
    lines0: List[str] = []
    pumps: List[threading.Thread] = []

    def on_signal(signum, frame):            # the launcher is being stopped (a driver-side timeout): take the ranks along
        raise KeyboardInterrupt(f"signal {signum}")

    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT):
            old_handlers[sig] = signal.signal(sig, on_signal)
    try:
        for r in range(world):
            p = subprocess.Popen(list(cmd), env=rank_env(r, world, port, env), stdout=subprocess.PIPE, stderr=None,
                                 text=True, bufsize=1)
            procs.append(p)
            t = (threading.Thread(target=_pump_rank0, args=(p.stdout, lines0, err), daemon=True) if r == 0 else
                 threading.Thread(target=_pump, args=(p.stdout, [], err), daemon=True))
            t.start()
            pumps.append(t)
        deadline = None if timeout_s is None else time.monotonic() + timeout_s
        code = 0
        alive = set(range(world))
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and code == 0:
                    code = rc
                    err.write(f"[launch] rank {r} exited with code {rc}; ending the other ranks\n")
            if code != 0 or (deadline is not None and time.monotonic() > deadline):
                if code == 0:
                    code = 124
                    err.write(f"[launch] job exceeded {timeout_s:.0f} s; ending all ranks\n")
                break
            if alive:
                time.sleep(0.05)
        return code
    finally:
        for p in procs:                      # exact PIDs of the children started above
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for t in pumps:
            t.join(timeout=5)
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
        err.flush()
        if lines0:                           # the line rank 0 printed last
            out.write(lines0[0] if lines0[0].endswith("\n") else lines0[0] + "\n")
            out.flush()


def visible_gpu_count() -> Optional[int]:
    """HIP devices visible to this process, WITHOUT importing torch (the first `import torch` on a fresh box pages the
    image in for a minute or two): hipGetDeviceCount through ctypes.  None when libamdhip64 cannot be loaded (the caller
    then lets the ranks find out)."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            lib = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        try:
            rc = lib.hipGetDeviceCount(ctypes.byref(n))
        except Exception:
            return None
        return int(n.value) if rc == 0 else 0
    return None


def self_spawn(world: int, script: str, argv: Sequence[str], **kw) -> int:
    """`python script argv...` as `world` ranks (what bench.py does for --gpus N when no launcher set WORLD_SIZE)."""
    return spawn_ranks(world, [sys.executable, script, *argv], **kw)
