"""Build libbuzzdetect_hip.so (gfx950) in-tree with hipcc.

    python -m buzzdetect_amd.build [--force]

The shared object lands next to its sources (``buzzdetect_amd/csrc/``) so that it travels
with the repository snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "libbuzzdetect_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ("engine.hip", "frontend.hip", "cnn.hip")
HEADERS = ("bd_internal.h", os.path.join("..", "..", "include", "buzzdetect_hip.h"))
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


STAMP_PATH = LIB_PATH + ".srchash"      # sha256 of the sources the library was built from (travels with it)


def source_hash() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def needs_build() -> bool:
    """True when there is no library or it was built from other sources (content hash, not mtimes: a snapshot copied
    to another machine keeps neither order nor times)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-o", LIB_PATH + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[buzzdetect_amd.build]", " ".join(cmd), flush=True)
    stamp = source_hash()
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(STAMP_PATH, "w") as f:
        f.write(stamp + "\n")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
