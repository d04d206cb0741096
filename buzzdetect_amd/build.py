"""Build libbuzzdetect_hip.so (gfx950) in-tree with hipcc.

    python -m buzzdetect_amd.build [--force]

The shared object lands next to its sources (``buzzdetect_amd/csrc/``) so that it travels
with the repository snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "libbuzzdetect_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ("engine.hip", "frontend.hip", "resample.hip", "sepchip.hip", "sepchipf32.hip", "sepmid.hip", "sepmidf32.hip", "septail.hip", "stemreg.hip", "stemregf32.hip", "l4regf32.hip", "cnn.hip", "rowfmt.hip")
# extra compiler flags of single files (part of the source hash below, like the sources themselves)
FILE_FLAGS = {
    # sep_chip_kernel lives at the 256-register limit of two waves per SIMD: the default machine scheduler spills 4-28 of its
    # long-lived values (address bases, pending tiles) depending on unrelated edits; the occupancy-driven one keeps them
    "sepchip.hip": ("-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"),
    "sepmid.hip": ("-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"),
    "sepchipf32.hip": ("-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"),
    "sepmidf32.hip": ("-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"),
}
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
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
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
    """Compile the library unless it is current.  Several processes may find it stale at once (every rank of a
    ``torchrun`` launch imports the package): the rebuild is serialised by a lock file next to the sources, each
    builder compiles into its own temporary name, and whoever gets the lock second finds the work done.  The command
    line goes to stderr - stdout belongs to the caller (bench.py prints one JSON line there)."""
    if not force and not needs_build():
        return LIB_PATH
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():          # another process built it while this one waited
                return LIB_PATH
            fd, tmp = tempfile.mkstemp(prefix=LIB_NAME + ".", suffix=".tmp", dir=CSRC)
            os.close(fd)
            objdir = tempfile.mkdtemp(prefix=".obj.", dir=CSRC)
            common = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
                      "-Wno-unused-function"]
            jobs = [(common + ["-c", os.path.join(CSRC, s), "-o", os.path.join(objdir, s + ".o")] + list(FILE_FLAGS.get(s, ())))
                    for s in SOURCES]
            cmd = common + ["-shared", "-o", tmp] + [j[j.index("-o") + 1] for j in jobs]
            if verbose:
                for j in jobs:
                    print("[buzzdetect_amd.build]", " ".join(j), file=sys.stderr, flush=True)
                print("[buzzdetect_amd.build]", " ".join(cmd), file=sys.stderr, flush=True)
            stamp = source_hash()
            try:
                # one translation unit per process, side by side (the kernels are independent files), then one link
                procs = [subprocess.Popen(j, stdout=sys.stderr) for j in jobs]
                rcs = [p.wait() for p in procs]
                if any(rcs):
                    raise subprocess.CalledProcessError(next(r for r in rcs if r), jobs[[bool(r) for r in rcs].index(True)])
                subprocess.run(cmd, check=True, stdout=sys.stderr)
                os.chmod(tmp, 0o755)
                with open(STAMP_PATH + ".tmp", "w") as f:
                    f.write(stamp + "\n")
                # the stamp goes first: a reader that sees the new library always sees its stamp
                if os.path.exists(STAMP_PATH):
                    os.remove(STAMP_PATH)
                os.replace(tmp, LIB_PATH)
                os.replace(STAMP_PATH + ".tmp", STAMP_PATH)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
                shutil.rmtree(objdir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
