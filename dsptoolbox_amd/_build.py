"""Build the HIP library in-tree:  python -m dsptoolbox_amd._build [--force]."""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdsptoolbox_amd.so")
SOURCES = [os.path.join(CSRC, "api.hip")]
# -fno-slp-vectorize: hipcc's SLP pass turns the complex butterflies into v_pk_* pairs glued
# together with ~270 v_mov per FFT; scalar fp32 VALU code is 25 % faster here (measured on
# MI355X, profiles/).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]


HASH_PATH = os.path.join(LIB_DIR, "libdsptoolbox_amd.srchash")


def _source_hash() -> str:
    """sha256 over the contents of every source the library is built from (names included).
    Content, not modification times: a checkout or a copy to another machine touches every file
    without changing anything, and must not trigger a two-minute rebuild."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for dp, _, names in os.walk(root):
            for f in names:
                if f.endswith((".hip", ".hpp", ".h")):
                    files.append(os.path.join(dp, f))
    for path in sorted(files, key=lambda q: os.path.relpath(q, HERE)):
        h.update(os.path.relpath(path, HERE).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _up_to_date() -> bool:
    if not (os.path.exists(LIB_PATH) and os.path.exists(HASH_PATH)):
        return False
    with open(HASH_PATH) as fh:
        return fh.read().strip() == _source_hash()


def build_library(force: bool = False, verbose: bool = True) -> str:
    """hipcc --offload-arch=gfx950 -> dsptoolbox_amd/lib/libdsptoolbox_amd.so

    Safe to call from several ranks at once: one process compiles (file lock) into a
    temporary name and renames it into place, the others wait and reuse the result."""
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and _up_to_date():
        return LIB_PATH
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date():  # another rank built it while we waited
                return LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            cmd = [hipcc] + FLAGS + ["-o", tmp] + SOURCES + ["-ldl", "-pthread"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr, flush=True)
            try:
                subprocess.check_call(cmd, stdout=sys.stderr)
                os.replace(tmp, LIB_PATH)
                with open(HASH_PATH, "w") as fh:
                    fh.write(_source_hash() + "\n")
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
