"""The host tier's staging-copy team (hip/copy_pool.hpp) is lock-free where it matters -- helpers steal blocks from one
generation-tagged counter and spin between the copies of a call -- so it gets its own stress test on the CPU box, under
ThreadSanitizer when the toolchain has it: a data race or a lost block in there would corrupt a drop-in call's output
silently, and the GPU tests exercise only a few hundred copies."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "copy_pool_stress.cpp")


def _build(tmp_path, flags):
    exe = str(tmp_path / "copy_pool_stress")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", *flags, "-o", exe, SRC], capture_output=True, text=True)
    return exe if r.returncode == 0 else None, r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_copy_pool_stress_plain(tmp_path):
    exe, err = _build(tmp_path, [])
    assert exe, err
    r = subprocess.run([exe, "500", "3", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_copy_pool_stress_thread_sanitizer(tmp_path):
    exe, err = _build(tmp_path, ["-fsanitize=thread"])
    if not exe:
        pytest.skip("no ThreadSanitizer runtime in this toolchain: " + err[-200:])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    r = subprocess.run([exe, "150", "2", "3"], capture_output=True, text=True, timeout=600, env=env)
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_copy_pool_stalled_helper_cannot_mix_two_jobs(tmp_path):
    """a helper parked between its loads of a job's fields, across the end of that job and the
    publication of the next one (different block size), must neither touch the old job's memory nor the new job's
    completion counter."""
    exe, err = _build(tmp_path, [])
    assert exe, err
    r = subprocess.run([exe, "stall"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok stall" in r.stdout, r.stdout + r.stderr
