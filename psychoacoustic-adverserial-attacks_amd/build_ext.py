"""Builds libpaa_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Staleness is decided by CONTENT, not by mtime: a sha256 over every file under csrc/ and include/ plus the compiler flags is
stored next to the library (libpaa_hip.so.key); a checkout or rsync that reorders mtimes can neither hide a source change nor
force a rebuild.  Object files carry their own keys (source + the headers of csrc/ and include/ + flags), so a one-file edit
recompiles one file."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["proj_kernels.hip", "spec_kernels.hip", "gemm.hip", "gemm_ring.hip", "gemm_ring2.hip", "model_kernels.hip", "conv0_dgrad.hip", "attention.hip", "model.hip"]
# PAA_EXTRA_HIPCC_FLAGS: diagnostic builds of tools/ (e.g. "-DPAA_EXPERIMENTS -DPAA_ABL=3"); never set by __graft_entry__.build()
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value"] + os.environ.get("PAA_EXTRA_HIPCC_FLAGS", "").split()
# a diagnostic build lives NEXT to the shipped library (its own file and object directory), never in its place
VARIANT = "_exp" if os.environ.get("PAA_EXTRA_HIPCC_FLAGS", "").strip() else ""
LIB = os.path.join(HERE, f"libpaa_hip{VARIANT}.so")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for x in extra:
        h.update(str(x).encode() + b"\0")
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _headers():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    out += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return out


def source_key() -> str:
    """Content hash of everything the library is built from."""
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return _digest(files, FLAGS + SOURCES)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def is_current() -> bool:
    return os.path.exists(LIB) and _read(LIB + ".key") == source_key()


def build(force: bool = False, verbose: bool = True) -> str:
    key = source_key()
    if not force and os.path.exists(LIB) and _read(LIB + ".key") == key:
        return LIB
    objdir = os.path.join(HERE, "build" + VARIANT)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = _headers()

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        okey = _digest([os.path.join(CSRC, src)] + hdrs, FLAGS)
        if not force and os.path.exists(obj) and _read(obj + ".key") == okey:
            return obj
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        with open(obj + ".key", "w") as f:
            f.write(okey)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(LIB + ".key", "w") as f:
        f.write(key)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
