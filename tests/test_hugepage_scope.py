"""What the host tier does to CALLER memory: madvise(MADV_HUGEPAGE) reaches only the 2-MiB
units of an output in which no page exists yet -- a fresh allocation -- and never warm memory: a reused output's VMA keeps
its flags, is not split, and its resident set does not change.  Read from /proc/self/smaps around
cnt_test_advise_output (the advice step of a host-tier call, runnable without a GPU) and, on the GPU box, around real
cnt_bits_to_n calls."""
import ctypes
import mmap
import os
import re

import numpy as np
import pytest

MIB = 1 << 20


def _vmas(lo, hi):
    """[(start, end, vmflags)] of the mappings that overlap [lo, hi)"""
    out, cur = [], None
    for line in open("/proc/self/smaps"):
        m = re.match(r"^([0-9a-f]+)-([0-9a-f]+) ", line)
        if m:
            cur = [int(m.group(1), 16), int(m.group(2), 16), ""]
            if cur[0] < hi and cur[1] > lo:
                out.append(cur)
            else:
                cur = None
        elif cur is not None and line.startswith("VmFlags:"):
            cur[2] = line.split(":", 1)[1].split()
    return [tuple(v) for v in out]


def _anon(nbytes):
    m = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    return m, addr


def _thp_mode():
    try:
        return re.search(r"\[(\w+)\]", open("/sys/kernel/mm/transparent_hugepage/enabled").read()).group(1)
    except OSError:
        return None


@pytest.fixture(scope="module")
def L():
    """the TEST-HOOKS build (tests/libcute_nt_hip_hooks.so, -DCNT_TEST_HOOKS): the product library exports no cnt_test_* symbol"""
    from cute_nucleotides_amd import _lib, build

    build.build_hooks()
    prev = _lib.use_build("hooks")
    lib = _lib.lib()
    _lib.use_build(prev)
    return lib


@pytest.fixture(scope="module")
def P():
    """the PRODUCT library, for the real host-tier calls"""
    from cute_nucleotides_amd import _lib

    assert _lib.active_build() == "product"
    return _lib.lib()


needs_smaps = pytest.mark.skipif(not os.path.exists("/proc/self/smaps"), reason="needs /proc/self/smaps")


@needs_smaps
def test_warm_output_is_left_alone(L):
    m, addr = _anon(64 * MIB)
    np.frombuffer(m, dtype=np.uint8)[:] = 1  # every page exists: a reused output
    before = _vmas(addr, addr + 64 * MIB)
    assert L.cnt_test_advise_output(ctypes.c_void_p(addr), 64 * MIB) == 0
    after = _vmas(addr, addr + 64 * MIB)
    assert after == before, (before, after)  # same mappings, same boundaries, same VmFlags
    assert all("hg" not in flags for _, _, flags in after)


@needs_smaps
def test_fresh_output_is_advised_and_a_half_warm_one_only_where_it_is_fresh(L):
    if _thp_mode() not in ("madvise", "always"):
        pytest.skip("transparent huge pages are off on this host: the advice is a no-op")
    m, addr = _anon(64 * MIB)
    assert L.cnt_test_advise_output(ctypes.c_void_p(addr), 64 * MIB) == 0
    lo = (addr + 2 * MIB - 1) & ~(2 * MIB - 1)
    hi = (addr + 64 * MIB) & ~(2 * MIB - 1)
    advised = [(s, e) for s, e, flags in _vmas(addr, addr + 64 * MIB) if "hg" in flags]
    assert advised and min(s for s, _ in advised) <= lo and max(e for _, e in advised) >= hi, advised  # the whole 2-MiB-aligned interior
    # second mapping: first half touched (warm), second half never -> only the second half carries the flag
    m2, a2 = _anon(64 * MIB)
    np.frombuffer(m2, dtype=np.uint8)[: 32 * MIB] = 1
    assert L.cnt_test_advise_output(ctypes.c_void_p(a2), 64 * MIB) == 0
    for s, e, flags in _vmas(a2, a2 + 64 * MIB):
        if "hg" in flags:
            assert max(s, a2) >= a2 + 32 * MIB - 2 * MIB, (hex(s), hex(e))  # nothing of the warm half (up to the unit that straddles)
    assert any("hg" in flags for s, e, flags in _vmas(a2 + 40 * MIB, a2 + 64 * MIB))
    # small outputs are never advised
    m3, a3 = _anon(4 * MIB)
    assert L.cnt_test_advise_output(ctypes.c_void_p(a3), 4 * MIB) == 0
    assert all("hg" not in flags for _, _, flags in _vmas(a3, a3 + 4 * MIB))
    assert L.cnt_test_advise_output(None, 64 * MIB) != 0


@needs_smaps
@pytest.mark.gpu
def test_real_calls_leave_a_warm_output_alone_and_advise_a_fresh_one(P, oracle):
    L = P
    n_len = 1 << 26
    n = oracle.fill_random_acgt(n_len, 5)
    bits = oracle.n_to_bits_lut(n)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    m, addr = _anon(n_len)
    warm = np.frombuffer(m, dtype=np.uint8)
    warm[:] = 0
    before = _vmas(addr, addr + n_len)
    for _ in range(2):
        assert L.cnt_bits_to_n(p(bits), bits.size, n_len, ctypes.c_void_p(addr)) == 0
    assert np.array_equal(warm, n) and _vmas(addr, addr + n_len) == before  # a reused output: flags and layout untouched
    if _thp_mode() in ("madvise", "always"):
        m2, a2 = _anon(n_len)
        assert L.cnt_bits_to_n(p(bits), bits.size, n_len, ctypes.c_void_p(a2)) == 0
        assert np.array_equal(np.frombuffer(m2, dtype=np.uint8), n)
        assert any("hg" in flags for _, _, flags in _vmas(a2, a2 + n_len))  # a fresh output: advised as before
