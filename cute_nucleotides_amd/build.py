"""Compile libcute_nt_hip.so (the C-ABI library, include/cute_nt.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so it travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "hip")
LIB = os.path.join(HERE, "libcute_nt_hip.so")
# the same translation unit with -DCNT_LAB_VARIANTS: every measured kernel variant + the process-global tuning knobs that
# select them (cnt_set_tuning).  Bench / test infrastructure, kept out of the product package on purpose.
LAB_LIB = os.path.join(os.path.dirname(HERE), "bench", "libcute_nt_hip_lab.so")
# ... and with -DCNT_TEST_HOOKS: the product's code plus the cnt_test_* hooks (fold shards onto fewer devices, run
# the huge-page advice alone, print the fused launch plan).  Test infrastructure: lives under tests/, loaded only by tests
# (and by the bench forms when a test folds them onto one GPU); the product exports none of it.
HOOKS_LIB = os.path.join(os.path.dirname(HERE), "tests", "libcute_nt_hip_hooks.so")
SOURCES = ["cute_nt.hip"]
# every file the one translation unit includes: a non-forced build() must notice an edit to any of them
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".h"))) + [os.path.join("..", "include", "cute_nt.h")]
ARCH = "gfx950"


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: cannot build libcute_nt_hip.so")
    return p


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(lib, defines, verbose):
    cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC"] + defines + \
          ["-o", lib + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    return lib


def build(force=False, verbose=False):
    """Build the PRODUCT library (defaults + the any-alignment kernels, no run-time kernel selection) if missing or older
    than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    return _compile(LIB, [], verbose)


def build_lab(force=False, verbose=False):
    """Build bench/libcute_nt_hip_lab.so: the same sources with -DCNT_LAB_VARIANTS (all kernel variants + cnt_set_tuning) and
    -DCNT_TEST_HOOKS (the bench A/B tools print launch plans and fold shards too)."""
    if not force and not _stale(LAB_LIB):
        return LAB_LIB
    return _compile(LAB_LIB, ["-DCNT_LAB_VARIANTS", "-DCNT_TEST_HOOKS"], verbose)


def build_hooks(force=False, verbose=False):
    """Build tests/libcute_nt_hip_hooks.so: the product's sources with -DCNT_TEST_HOOKS (same kernels, plus cnt_test_*)."""
    if not force and not _stale(HOOKS_LIB):
        return HOOKS_LIB
    return _compile(HOOKS_LIB, ["-DCNT_TEST_HOOKS"], verbose)


if __name__ == "__main__":
    import sys
    import time

    t0 = time.time()
    print(build(force=True, verbose=True), "%.1f s" % (time.time() - t0))
    if "--lab" in sys.argv:
        t0 = time.time()
        print(build_lab(force=True, verbose=True), "%.1f s" % (time.time() - t0))
    if "--hooks" in sys.argv:
        t0 = time.time()
        print(build_hooks(force=True, verbose=True), "%.1f s" % (time.time() - t0))
