"""Pinned caller memory (round 6): a side of a host-slice call that lies in pinned memory is used in place by the copy engines
(hip/host_tier.inc `in_direct` / `out_direct`; include/cute_nt.h "pinned caller memory").  Results must not depend on which sides
are pinned, at any offset inside the pinned buffers, for both codecs, the validated forms and the sharded tier; the hooks build's
trace says which sides went unstaged."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(1 << 20) + 1, (1 << 21) + 12345, 3 << 20, (1 << 23) + 7, (1 << 25) + 31, 1 << 26]


def _mk(kind, count, dtype, cn, off=0):
    """an array of `count` elements that is pinned (cnt_host_alloc), torch-pinned, registered-in-place or ordinary, starting
    `off` elements into its allocation"""
    if kind == "alloc":
        return cn.pinned_empty(count + off, dtype)[off:]
    if kind == "torch":
        import torch

        t = torch.empty((count + off) * np.dtype(dtype).itemsize, dtype=torch.uint8, pin_memory=True)
        return t.numpy().view(dtype)[off:]
    return np.empty(count + off, dtype)[off:]


@pytest.mark.parametrize("kind_in,kind_out", [("alloc", "alloc"), ("alloc", "plain"), ("plain", "alloc"), ("torch", "torch"), ("plain", "plain")])
def test_results_do_not_depend_on_which_sides_are_pinned(oracle, kind_in, kind_out):
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import n_to_bits2 as n2

    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    alpha5 = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
    for i, n_len in enumerate(SIZES[:5]):
        off_in, off_out = [0, 1, 13, 64, 5][i], [0, 3, 1, 8, 2][i]
        words = (n_len + 31) // 32
        n = _mk(kind_in, n_len, np.uint8, cn, off_in)
        n[:] = alpha[rng.integers(0, 10, n_len)]
        assert cn.is_pinned(n) == (kind_in != "plain")
        out = _mk(kind_out, words + 2, np.uint64, cn, off_out)
        out[:] = 0x5A5A5A5A5A5A5A5A
        got = cn.n_to_bits_hip_into(n, out)
        want = oracle.n_to_bits_lut(n)
        assert np.array_equal(got, want), (n_len, kind_in, kind_out)
        assert (out[words:] == 0x5A5A5A5A5A5A5A5A).all()
        # decode a prefix back: packed words pinned like `out`, letters pinned like `n`
        length = n_len - int(rng.integers(0, 40))
        back = _mk(kind_in, n_len + 16, np.uint8, cn, off_in)
        back[:] = 0x2A
        got = cn.bits_to_n_hip_into(out[:words], length, back)
        assert np.array_equal(got, oracle.bits_to_n_lut(want, length)), (n_len, length, kind_in, kind_out)
        assert (back[length:] == 0x2A).all()
        # validated: a handful of strays, counted by the same pass
        where = rng.integers(0, n_len, 7)
        n[where] = 0x21
        w, bad = cn.n_to_bits_hip_checked(n, strict_lut=True)
        assert bad == oracle.validate(n) and np.array_equal(w, oracle.n_to_bits_lut(n)), (n_len, kind_in)
        # 5-letter codec
        n[:] = alpha5[rng.integers(0, 12, n_len)]
        w5 = _mk(kind_out, (n_len + 26) // 27, np.uint64, cn, off_out)
        got5 = n2.n_to_bits2_hip_into(n, w5)
        want5 = oracle.n_to_bits2_lut(n)
        assert np.array_equal(got5, want5), (n_len, kind_in, kind_out)
        got = n2.bits_to_n2_hip_into(got5, length, back)
        assert np.array_equal(got, oracle.bits_to_n2_lut(want5, length)), (n_len, length)


def test_registered_in_place_and_released(oracle):
    """cnt_host_register pins an ordinary array for the block; inside it the tier uses it in place, after it stages again"""
    import cute_nucleotides_amd as cn

    rng = np.random.default_rng(12)
    n_len = (1 << 22) + 77
    n = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_len)].copy()
    want = oracle.n_to_bits_lut(n)
    out = np.empty(want.size, dtype=np.uint64)
    assert not cn.is_pinned(n)
    with cn.host_registered(n), cn.host_registered(out):
        assert cn.is_pinned(n) and cn.is_pinned(out) and cn.is_pinned(n[5:1000])
        assert np.array_equal(cn.n_to_bits_hip_into(n, out), want)
    assert not cn.is_pinned(n) and not cn.is_pinned(out)
    out[:] = 0
    assert np.array_equal(cn.n_to_bits_hip_into(n, out), want)


def test_a_range_that_leaves_its_pinned_allocation_is_staged(oracle):
    """is_pinned asks about the WHOLE range: an ordinary array is not pinned because its first byte happens to be, and freeing a
    pinned buffer makes later calls on that address range ordinary again"""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    a = cn.pinned_empty(1 << 16, np.uint8)
    p = a.ctypes.data
    assert L.cnt_host_is_pinned(ctypes.c_void_p(p), a.size) == 1
    assert L.cnt_host_is_pinned(ctypes.c_void_p(p + 100), a.size - 100) == 1
    assert L.cnt_host_is_pinned(ctypes.c_void_p(p), a.size + (4 << 20)) == 0
    assert L.cnt_host_is_pinned(None, 10) == 0 and L.cnt_host_is_pinned(ctypes.c_void_p(p), 0) == 0
    assert L.cnt_host_alloc(None, 10) == _lib.CNT_EINVAL and L.cnt_host_register(None, 10) == _lib.CNT_EINVAL
    q = ctypes.c_void_p()
    assert L.cnt_host_alloc(ctypes.byref(q), 0) == _lib.CNT_EINVAL
    assert L.cnt_host_free(None) == 0 and L.cnt_host_unregister(None) == _lib.CNT_EINVAL
    assert L.cnt_host_unregister(ctypes.c_void_p(np.empty(4096, np.uint8).ctypes.data)) < 0  # never registered: the runtime's error


def test_trace_shows_which_sides_went_unstaged(oracle, hooks_build):
    """the hooks build stamps 5 / 6 = input / output used in place, and no staging copy (tag 3 follows tag 2 within a
    microsecond) when the input is pinned"""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    n_len = (1 << 25) + 8192  # above the single-kernel lane of calls with both sides pinned (2^25 nt): the pipeline, traced
    rng = np.random.default_rng(13)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_len)]
    for pin_in, pin_out in ((True, True), (True, False), (False, True), (False, False)):
        n = cn.pinned_empty(n_len, np.uint8) if pin_in else np.empty(n_len, np.uint8)
        n[:] = letters
        out = cn.pinned_empty(n_len // 32, np.uint64) if pin_out else np.empty(n_len // 32, np.uint64)
        want = oracle.n_to_bits_lut(letters)
        assert np.array_equal(cn.n_to_bits_hip_into(n, out), want)
        tags = (ctypes.c_int * 4096)()
        us = (ctypes.c_double * 4096)()
        k = L.cnt_test_host_trace(tags, us, 4096)
        flags = {tags[i]: us[i] for i in range(k) if tags[i] in (5, 6)}
        assert flags == {5: float(pin_in), 6: float(pin_out)}, (pin_in, pin_out, flags)
        back = cn.pinned_empty(n_len, np.uint8) if pin_in else np.empty(n_len, np.uint8)
        assert np.array_equal(cn.bits_to_n_hip_into(out, n_len, back), oracle.bits_to_n_lut(out, n_len))
        k = L.cnt_test_host_trace(tags, us, 4096)
        flags = {tags[i]: us[i] for i in range(k) if tags[i] in (5, 6)}
        assert flags == {5: float(pin_out), 6: float(pin_in)}, (pin_in, pin_out, flags)


def test_sharded_host_tier_takes_pinned_slices(oracle, hooks_build):
    """every shard's worker asks about ITS part of the caller's slices"""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import sharding

    sharding.alias_devices(True)
    try:
        rng = np.random.default_rng(14)
        n_len = (1 << 23) + 999
        n = cn.pinned_empty(n_len, np.uint8)
        n[:] = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)[rng.integers(0, 10, n_len)]
        want = oracle.n_to_bits_lut(n)
        for ndev in (1, 2, 3, 8):
            assert np.array_equal(cn.n_to_bits_hip_sharded(n, ndev=ndev), want), ndev
            assert np.array_equal(cn.bits_to_n_hip_sharded(want, n_len - 3, ndev=ndev), oracle.bits_to_n_lut(want, n_len - 3)), ndev
    finally:
        sharding.alias_devices(False)


def test_a_failed_runtime_call_does_not_resurface_behind_the_next_launch(oracle):
    """The runtime keeps a failed call's code as the thread's "last error", and the check behind every kernel launch here reads
    exactly that: a status the library has handed to its caller (an unregister of memory that was never registered) must not
    come back as the error of the next, healthy call on an already initialised thread."""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(15)
    n = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (1 << 21) + 5)]
    want = oracle.n_to_bits_lut(n)
    assert np.array_equal(cn.n_to_bits_hip(n), want)  # streams, ring, chip info: all set up
    junk = np.empty(8192, np.uint8)
    for _ in range(3):
        assert L.cnt_host_unregister(ctypes.c_void_p(junk.ctypes.data)) < 0
        assert np.array_equal(cn.n_to_bits_hip(n), want)
        assert np.array_equal(cn.n_to_bits_hip(n[:40000]), want[:1250])  # the small path's launch check as well
        assert L.cnt_host_free(ctypes.c_void_p(junk.ctypes.data)) < 0  # not a pinned allocation
        import torch

        d = torch.from_numpy(n.copy()).cuda()
        assert np.array_equal(cn.n_to_bits_dev(d).cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("n_len", [1 << 16, (1 << 16) + 1, 100003, (1 << 18) - 31, 1 << 19, (1 << 20) - 5, 1 << 20])
def test_small_calls_between_pinned_slices_use_them_in_place(oracle, n_len):
    """2^16 .. 2^20 nt with both slices pinned: ONE kernel reads and writes the caller's buffers over the link (no memcpy on
    either side) -- any alignment inside the pinned allocations, every flag mode, both codecs, the validated forms, guards intact"""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import n_to_bits2 as n2

    rng = np.random.default_rng(n_len)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    alpha5 = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
    for off_in, off_out in ((0, 0), (1, 1), (13, 3), (64, 0), (127, 15)):
        words = (n_len + 31) // 32
        n = cn.pinned_empty(n_len + off_in + 32, np.uint8)[off_in : off_in + n_len]
        n[:] = alpha[rng.integers(0, 10, n_len)]
        whole = cn.pinned_empty(words + off_out + 4, np.uint64)
        whole[:] = 0x5A5A5A5A5A5A5A5A
        out = whole[off_out:]
        for strict, tail in ((False, False), (True, False), (False, True)):
            got = cn.n_to_bits_hip_into(n, out, strict_lut=strict, tail_lut=tail)
            assert np.array_equal(got, oracle.n_to_bits_lut(n)), (n_len, off_in, off_out, strict, tail)
        assert (whole[:off_out] == 0x5A5A5A5A5A5A5A5A).all() and (out[words:] == 0x5A5A5A5A5A5A5A5A).all()
        want = oracle.n_to_bits_lut(n)
        length = n_len - int(rng.integers(0, 33))
        bwhole = cn.pinned_empty(n_len + off_in + 64, np.uint8)
        bwhole[:] = 0x2A
        back = bwhole[off_in:]
        got = cn.bits_to_n_hip_into(out[:words], length, back)
        assert np.array_equal(got, oracle.bits_to_n_lut(want, length)), (n_len, length, off_in, off_out)
        assert (bwhole[:off_in] == 0x2A).all() and (back[length:] == 0x2A).all()
        dirty = n.copy()
        dirty[rng.integers(0, n_len, 9)] = 0x7F
        pd = cn.pinned_empty(n_len, np.uint8)
        pd[:] = dirty
        # the returning form allocates an ordinary output: staged; the C entry point with a pinned output: in place
        import ctypes

        from cute_nucleotides_amd import _lib

        bad = ctypes.c_uint64(0)
        assert _lib.lib().cnt_n_to_bits_checked(ctypes.c_void_p(pd.ctypes.data), n_len, ctypes.c_void_p(out.ctypes.data), words, _lib.CNT_STRICT_LUT, ctypes.byref(bad)) == 0
        assert bad.value == oracle.validate(dirty) and np.array_equal(out[:words], oracle.n_to_bits_lut(dirty)), (n_len, off_in, off_out)
        n5 = cn.pinned_empty(n_len + off_in, np.uint8)[off_in:]
        n5[:] = alpha5[rng.integers(0, 12, n_len)]
        w5 = cn.pinned_empty((n_len + 26) // 27 + off_out, np.uint64)[off_out:]
        got5 = n2.n_to_bits2_hip_into(n5, w5)
        want5 = oracle.n_to_bits2_lut(n5)
        assert np.array_equal(got5, want5), (n_len, off_in, off_out)
        assert np.array_equal(n2.bits_to_n2_hip_into(got5, length, back), oracle.bits_to_n2_lut(want5, length)), (n_len, length)


@pytest.mark.parametrize("n_len", [1, 31, 4097, (1 << 16) + 3, (1 << 22) + 77])
def test_packed_ops_on_pinned_slices(oracle, n_len):
    """cnt_hamming / cnt_complement / cnt_reverse_complement / cnt_validate with their slices in pinned memory run as one kernel
    over the link (no copies, no device scratch); same answers as from ordinary memory and as the oracle's definitions; an
    output that overlaps its input keeps the old path"""
    import ctypes

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib, packed_ops as po

    L = _lib.lib()
    rng = np.random.default_rng(77 + n_len)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    letters = alpha[rng.integers(0, 10, n_len)]
    letters2 = alpha[rng.integers(0, 10, n_len)]
    a, b = oracle.n_to_bits_lut(letters), oracle.n_to_bits_lut(letters2)
    words = a.size
    for off in (0, 1, 5):
        pa = cn.pinned_empty(words + off, np.uint64)[off:]
        pb = cn.pinned_empty(words + off + 2, np.uint64)[off + 2:]
        pa[:], pb[:] = a, b
        assert po.hamming_hip(pa, pb, n_len) == oracle.hamming(a, b, n_len) == po.hamming_hip(a, b, n_len), (n_len, off)
        out = cn.pinned_empty(words + off + 3, np.uint64)
        out[:] = 0x5A5A5A5A5A5A5A5A
        view = out[off : off + words]
        p = lambda x: ctypes.c_void_p(x.ctypes.data)
        assert L.cnt_complement(p(pa), n_len, p(view)) == 0
        assert np.array_equal(view, oracle.complement(a, n_len)), (n_len, off)
        assert L.cnt_reverse_complement(p(pa), n_len, p(view)) == 0
        assert np.array_equal(view, oracle.reverse_complement(a, n_len)), (n_len, off)
        assert (out[:off] == 0x5A5A5A5A5A5A5A5A).all() and (out[off + words :] == 0x5A5A5A5A5A5A5A5A).all()
        # in place (output == input): the staged path, as always
        keep = pa.copy()
        assert L.cnt_complement(p(pa), n_len, p(pa)) == 0 and np.array_equal(pa, oracle.complement(keep, n_len))
        pn = cn.pinned_empty(n_len + off, np.uint8)[off:]
        pn[:] = letters
        pn[rng.integers(0, n_len, 3)] = 0x2D
        assert po.validate_hip(pn) == oracle.validate(pn) and po.validate_hip(pn, allow_n=True) == oracle.validate(pn, allow_n=True), (n_len, off)
