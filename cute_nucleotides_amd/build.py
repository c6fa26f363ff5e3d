"""Compile libcute_nt_hip.so (the C-ABI library, include/cute_nt.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so it travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcute_nt_hip.so")
SOURCES = ["cute_nt.hip"]
# every file the one translation unit includes: a non-forced build() must notice an edit to any of them
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".h"))) + [os.path.join("..", "..", "include", "cute_nt.h")]
ARCH = "gfx950"


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: cannot build libcute_nt_hip.so")
    return p


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Build the library if missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
