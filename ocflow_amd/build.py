"""Build libocflow_hip.so (the C-ABI shared library of include/ocflow_hip.h) for gfx950 with hipcc.

In-tree build: objects under ocflow_amd/_build/, library at ocflow_amd/lib/libocflow_hip.so (git-ignored,
travels to the GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(PKG, "_build")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libocflow_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the OCFlow MI355X kernels cannot be built")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp() -> str:
    h = hashlib.sha256()
    for p in _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            [os.path.join(INCLUDE, "ocflow_hip.h"), os.path.abspath(__file__)]:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode()); h.update(f.read())  # path-independent: no rebuild on the GPU box
    return h.hexdigest()


def _compile(src: str, extra) -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    cmd = [_hipcc(), *FLAGS, *extra, "-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp_file = os.path.join(OBJ_DIR, "stamp")
    stamp = _stamp() + " " + " ".join(extra_flags)

    def fresh():
        return os.path.exists(LIB_PATH) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp

    if not force and fresh():
        return LIB_PATH
    # one builder at a time: the ranks of a multi-GPU launch all import the package at once
    with open(os.path.join(OBJ_DIR, "lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():  # another rank built it while this one waited
            return LIB_PATH
        with cf.ThreadPoolExecutor(max_workers=4) as ex:
            objs = list(ex.map(lambda s: _compile(s, list(extra_flags)), _sources()))
        tmp = LIB_PATH + f".tmp{os.getpid()}"
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB_PATH)
        with open(stamp_file, "w") as f:
            f.write(stamp)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
