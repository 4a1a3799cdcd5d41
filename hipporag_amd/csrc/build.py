"""Build libhrag.so for gfx950 with hipcc (no cmake, no torch extension machinery).

    python -m hipporag_amd.csrc.build        # or: from hipporag_amd.csrc.build import build

The library is written IN-TREE (hipporag_amd/libhrag.so) so that it travels with the repository
snapshot to the GPU box.  hipcc cross-compiles for gfx950 without a GPU present.
"""

from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(HERE, "_obj")
LIB = os.path.join(PKG, "libhrag.so")
SOURCES = ["errors.cpp", "ppr_spmm.hip", "ppr16.hip", "ppr8.hip", "ppr_sv.hip", "layout.hip", "sim_gemm.hip", "sim_gemm256.hip", "sim_gemv.hip", "topk.hip", "knn.hip", "seeds.hip",
           "engine.hip", "shard.hip", "shard_driver.hip"]
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(HERE, "engine_impl.h"), os.path.join(INCLUDE, "hrag.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + INCLUDE, "-I" + HERE]


class ToolchainMissing(RuntimeError):
    """No hipcc on this machine (a deployment box that received the built library)."""


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise ToolchainMissing("hipcc not found: libhrag.so cannot be built (ROCm toolchain required)")
    return exe


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what changed and relink.  Safe under concurrent callers (the N ranks of a torchrun job all import
    the package): the whole build runs under an exclusive file lock, objects and the library are written to a
    temporary name and renamed into place, so a peer never dlopens a half-written file and the ranks that waited
    find everything up to date."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(hipcc, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc: str, force: bool, verbose: bool) -> str:
    hdr_digest = _digest(HEADERS)

    def compile_one(src: str):
        src_path = os.path.join(HERE, src)
        obj = os.path.join(OBJ_DIR, src + ".o")
        stamp = obj + ".sha"
        want = _digest([src_path]) + hdr_digest
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
            return obj, False
        tmp = obj + f".tmp{os.getpid()}"
        cmd = [hipcc, *FLAGS, "-x", "hip", "-c", src_path, "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
        with open(stamp, "w") as f:
            f.write(want)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not os.path.exists(LIB):
        tmp = LIB + f".tmp{os.getpid()}"
        # -z defs: an internal launcher that is declared but not defined (or defined in an anonymous namespace) fails
        # HERE, on the CPU build box, instead of at the first call on the GPU box (the library is dlopen'ed RTLD_LAZY)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", tmp, *objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
