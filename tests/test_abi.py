"""C-ABI surface checks that need no GPU: the library loads, exports every symbol
include/cute_nt.h declares, and its argument validation mirrors the reference's error
behaviour.  No compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from cute_nucleotides_amd import build

    build.build()
    from cute_nucleotides_amd import _lib

    return _lib.lib()


def _declared(hooks=False):
    """symbols include/cute_nt.h declares: outside its #ifdef CNT_TEST_HOOKS section (the product's ABI), or inside it"""
    text = open(os.path.join(ROOT, "include", "cute_nt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = "".join(re.findall(r"#ifdef CNT_TEST_HOOKS(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef CNT_TEST_HOOKS.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cnt_\w+)\s*\(", inside if hooks else outside)))


def _exported(path):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("cnt_")})


def test_every_declared_symbol_is_exported_and_bound(L):
    from cute_nucleotides_amd import _lib

    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(L, name), "header declares %s but the library does not export it" % name
        assert name in _lib.SIGNATURES, "%s not bound in _lib.SIGNATURES" % name
    assert sorted(_lib.SIGNATURES) == names
    assert _exported(_lib.LIB_PATH) == names  # ... and the library exports nothing the header does not declare


def test_product_exports_no_test_hook():
    """cnt_test_alias_devices / cnt_test_advise_output / cnt_test_round_trip_plan are declared under
    #ifdef CNT_TEST_HOOKS and exist only in tests/libcute_nt_hip_hooks.so (and the lab build): the product's dynamic symbol
    table has no cnt_test_* entry, and the hooks build adds exactly those to the product's ABI."""
    from cute_nucleotides_amd import _lib, build

    hooks = _declared(hooks=True)
    assert hooks == sorted(_lib.TEST_HOOK_SIGNATURES) == ["cnt_test_advise_output", "cnt_test_alias_devices", "cnt_test_decode_plan", "cnt_test_host_trace", "cnt_test_pipeline_pieces", "cnt_test_round_trip_plan"]
    product = _exported(build.build())
    assert not [n for n in product if n.startswith("cnt_test_")], product
    assert _exported(build.build_hooks()) == sorted(product + hooks)
    text = open(os.path.join(ROOT, "include", "cute_nt.h")).read()
    assert "CNT_TEST_HOOKS" not in open(os.path.join(ROOT, "tests", "c_link_check.c")).read()  # a C caller of the product never sees them
    assert text.count("#ifdef CNT_TEST_HOOKS") == 1


def test_product_never_imports_oracle():
    seen = 0
    for top in ("cute_nucleotides_amd", "hip", "include"):  # the Python package, the kernels + C-ABI shim, the header
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".inc")) or f == "Makefile":
                    src = open(os.path.join(dirpath, f)).read()
                    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("#", "//", "*", '"""')))
                    assert not re.search(r"^\s*(from|import)\s+oracle", code, re.M), f
                    assert "cnt_oracle" not in code and "libcnt_oracle" not in code, f
                    seen += 1
    assert seen >= 20


def test_words_for(L):
    for n in (0, 1, 31, 32, 33, 64, 2**34, 2**36 + 5):
        assert L.cnt_words_for(n) == (n + 31) // 32
        assert L.cnt_words2_for(n) == (n + 26) // 27


def test_strerror_carries_the_reference_panic_text(L):
    from cute_nucleotides_amd import _lib

    assert _lib.strerror(_lib.CNT_ELEN) == "The length is greater than the number of nucleotides!"
    assert _lib.strerror(0) == "ok"
    for s in (1, 2, 4, 5, -1, -100, 99):
        assert isinstance(_lib.strerror(s), str) and _lib.strerror(s)


def test_argument_errors_before_any_device_work(L):
    from cute_nucleotides_amd import _lib

    n = np.frombuffer(b"ACGT" * 16, dtype=np.uint8)
    out = np.zeros(1, dtype=np.uint64)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    # capacity: 64 nt need 2 words
    assert L.cnt_n_to_bits(p(n), 64, p(out), 1) == _lib.CNT_ECAP
    assert L.cnt_n_to_bits_dev(p(n), 64, p(out), 1, 0, None) == _lib.CNT_ECAP
    assert L.cnt_n_to_bits2(p(n), 64, p(out), 2) == _lib.CNT_ECAP
    # len guard (reference panic): 1 word holds 32 nt / 27 nt
    buf = np.zeros(64, dtype=np.uint8)
    assert L.cnt_bits_to_n(p(out), 1, 33, p(buf)) == _lib.CNT_ELEN
    assert L.cnt_bits_to_n_dev(p(out), 1, 33, p(buf), 0, None) == _lib.CNT_ELEN
    assert L.cnt_bits_to_n2(p(out), 1, 28, p(buf)) == _lib.CNT_ELEN
    assert L.cnt_bits_to_n_sharded(p(out), 1, 33, p(buf), 1) == _lib.CNT_ELEN
    # empty input -> empty output, success, no device needed (SURVEY 8a iv)
    assert L.cnt_n_to_bits(None, 0, None, 0) == _lib.CNT_OK
    assert L.cnt_bits_to_n(None, 0, 0, None) == _lib.CNT_OK
    assert L.cnt_n_to_bits2(None, 0, None, 0) == _lib.CNT_OK
    assert L.cnt_n_to_bits_dev(None, 0, None, 0, 0, None) == _lib.CNT_OK
    # null with non-zero size
    assert L.cnt_n_to_bits(None, 64, p(out), 2) == _lib.CNT_EINVAL
    # unknown flag
    assert L.cnt_n_to_bits_dev(p(n), 32, p(out), 1, 0x80, None) == _lib.CNT_EINVAL
    # input and output of one call never share memory (the reference returns a fresh Vec): every tier refuses
    big = np.zeros(4096, dtype=np.uint8)
    q = lambda off: ctypes.c_void_p(big.ctypes.data + off)
    assert L.cnt_n_to_bits(q(0), 64, q(56), 2) == _lib.CNT_EINVAL  # out starts inside n
    assert L.cnt_n_to_bits(q(8), 64, q(0), 2) == _lib.CNT_EINVAL  # out ends inside n
    assert L.cnt_n_to_bits_dev(q(0), 64, q(0), 2, 0, None) == _lib.CNT_EINVAL
    assert L.cnt_n_to_bits2_dev(q(0), 54, q(48), 2, 0, None) == _lib.CNT_EINVAL
    assert L.cnt_n_to_bits_sharded(q(0), 64, q(0), 2, 1) == _lib.CNT_EINVAL
    assert L.cnt_bits_to_n(q(0), 2, 64, q(8)) == _lib.CNT_EINVAL
    assert L.cnt_bits_to_n_dev(q(64), 2, 64, q(8), 0, None) == _lib.CNT_EINVAL  # out's last byte is bits' first
    assert L.cnt_bits_to_n2_dev(q(0), 2, 54, q(8), 0, None) == _lib.CNT_EINVAL
    assert L.cnt_bits_to_n_sharded(q(0), 2, 64, q(15), 1) == _lib.CNT_EINVAL
    assert L.cnt_round_trip_dev(q(0), 64, q(1024), 2, q(32), 0, None) == _lib.CNT_EINVAL  # back inside n
    assert L.cnt_round_trip_dev(q(0), 64, q(56), 2, q(2048), 0, None) == _lib.CNT_EINVAL  # packed inside n
    assert L.cnt_round_trip_dev(q(0), 64, q(1024), 2, q(1032), 0, None) == _lib.CNT_EINVAL  # back inside packed
    # ... adjacent is fine: past the argument checks these need a device (none here: CNT_ENODEV; on a GPU box they would run
    # on host pointers, so they are only tried without one)
    count = ctypes.c_int(-1)
    assert L.cnt_device_count(ctypes.byref(count)) == _lib.CNT_OK
    if count.value == 0:
        assert L.cnt_n_to_bits(q(0), 64, q(64), 2) == _lib.CNT_ENODEV
        assert L.cnt_bits_to_n(q(64), 2, 64, q(0)) == _lib.CNT_ENODEV
    # generator offsets must sit on block boundaries
    assert L.cnt_fill_random_acgt_dev(p(buf), 5, 32, 1, None) == _lib.CNT_ERANGE
    assert L.cnt_fill_random_acgtn_dev(p(buf), 5, 27, 1, None) == _lib.CNT_ERANGE


def test_round_6_entry_points_check_their_arguments_before_any_device_work(L):
    """the validated encode's counter is never optional; an explicit device list is held to its contract BEFORE the device count is
    asked (bogus lists are CNT_EINVAL on any box, an index past the visible devices CNT_ENODEV like everywhere else); a handle
    that is not a live queue is CNT_EINVAL"""
    from cute_nucleotides_amd import _lib

    n = np.frombuffer(b"ACGT" * 16, dtype=np.uint8)
    out = np.zeros(2, dtype=np.uint64)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    bad = ctypes.c_uint64(99)
    assert L.cnt_n_to_bits_checked(p(n), 64, p(out), 1, 0, ctypes.byref(bad)) == _lib.CNT_ECAP and bad.value == 0
    assert L.cnt_n_to_bits_checked(p(n), 64, p(out), 2, 0, None) == _lib.CNT_EINVAL
    assert L.cnt_n_to_bits2_checked(p(n), 64, p(out), 2, 0, ctypes.byref(bad)) == _lib.CNT_ECAP
    assert L.cnt_n_to_bits_checked(p(n), 64, p(out), 2, 0x80, ctypes.byref(bad)) == _lib.CNT_EINVAL
    assert L.cnt_n_to_bits_checked_dev(p(n), 64, p(out), 2, 0, None, None) == _lib.CNT_EINVAL
    assert L.cnt_round_trip_checked_dev(p(n), 64, p(out), 2, p(n), 0, None, None) == _lib.CNT_EINVAL
    q = ctypes.c_void_p(1)
    devs = lambda *v: (ctypes.c_int * len(v))(*v)
    assert L.cnt_sharded_dev_open_on_devices(0, devs(0), 0, ctypes.byref(q)) == _lib.CNT_EINVAL and not q.value
    assert L.cnt_sharded_dev_open_on_devices(2, None, 0, ctypes.byref(q)) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_open_on_devices(2, devs(0, -1), 0, ctypes.byref(q)) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_open_on_devices(65, devs(*([0] * 65)), 0, ctypes.byref(q)) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_open_on_devices(1, devs(0), 2, ctypes.byref(q)) == _lib.CNT_EINVAL  # unknown queue flag
    assert L.cnt_sharded_dev_open_on_devices(1, devs(0), 0, None) == _lib.CNT_EINVAL
    count = ctypes.c_int(-1)
    assert L.cnt_device_count(ctypes.byref(count)) == _lib.CNT_OK
    # a well-formed list naming a device that is not there: "device index out of range" (on this box: no device at all)
    assert L.cnt_sharded_dev_open_on_devices(2, devs(0, count.value + 3), 0, ctypes.byref(q)) == _lib.CNT_ENODEV and not q.value
    d = ctypes.c_int(7)
    assert L.cnt_sharded_dev_device(ctypes.c_void_p(0xDEAD0), 0, ctypes.byref(d)) == _lib.CNT_EINVAL and d.value == 7
    nt = ctypes.c_uint64(0)
    assert L.cnt_chip_cache_nt(0, None) == _lib.CNT_EINVAL
    assert L.cnt_chip_cache_nt(count.value + 1, ctypes.byref(nt)) == _lib.CNT_ENODEV


def test_pinned_memory_entry_points_without_a_device(L):
    """argument errors first, then CNT_ENODEV: nothing can be pinned for a device that is not there, nothing is pinned, and
    freeing nothing is fine (this box has no GPU; tests/test_gpu_pinned.py is the other half)"""
    from cute_nucleotides_amd import _lib

    buf = np.zeros(8192, dtype=np.uint8)
    p = ctypes.c_void_p(buf.ctypes.data)
    q = ctypes.c_void_p(0x1234)
    assert L.cnt_host_alloc(None, 4096) == _lib.CNT_EINVAL
    assert L.cnt_host_alloc(ctypes.byref(q), 0) == _lib.CNT_EINVAL
    assert L.cnt_host_register(None, 4096) == _lib.CNT_EINVAL and L.cnt_host_register(p, 0) == _lib.CNT_EINVAL
    assert L.cnt_host_unregister(None) == _lib.CNT_EINVAL and L.cnt_host_free(None) == _lib.CNT_OK
    count = ctypes.c_int(-1)
    assert L.cnt_device_count(ctypes.byref(count)) == _lib.CNT_OK
    if count.value == 0:
        assert L.cnt_host_alloc(ctypes.byref(q), 4096) == _lib.CNT_ENODEV and not q.value
        assert L.cnt_host_register(p, buf.size) == _lib.CNT_ENODEV
        assert L.cnt_host_is_pinned(p, buf.size) == 0
    assert L.cnt_host_is_pinned(None, 16) == 0 and L.cnt_host_is_pinned(p, 0) == 0


def test_product_build_has_no_kernel_selection(L):
    """the product library contains the default kernels only and nothing that selects code at
    run time -- every cnt_set_tuning key is CNT_EINVAL, the variant tables hold one entry, the getters answer constants."""
    from cute_nucleotides_amd import _lib, devutil

    assert not _lib.is_lab_build() and devutil.get_tuning("lab_build") == 0
    for key in ("encode", "decode", "encode2", "decode2", "round_trip_shape", "round_trip_cap", "reduce_persistent", "reduce_xi",
                "xcd_shift", "small_nt", "launch_tiles", "nonsense"):
        assert L.cnt_set_tuning(key.encode(), 0) == _lib.CNT_EINVAL, key
    for key in ("encode", "decode", "encode2", "decode2"):
        assert devutil.get_tuning(key) == 0 and devutil.get_tuning(key + "_variants") == 1
        assert L.cnt_tuning_name(key.encode(), 0) and L.cnt_tuning_name(key.encode(), 1) is None
    assert devutil.get_tuning("small_nt") == 1 << 17 and devutil.get_tuning("reduce_persistent") == 1
    with pytest.raises(_lib.CuteNtError, match="use_lab_build"):
        devutil.set_tuning("encode", 0)


def test_tuning_knobs(L, lab_build):
    from cute_nucleotides_amd import _lib, devutil

    L = _lib.lib()  # the lab build
    for key in ("encode", "decode"):
        old = devutil.get_tuning(key)
        assert old == 0  # the shipped default
        n = devutil.get_tuning(key + "_variants")
        assert n >= 2
        try:
            for v in range(n):
                devutil.set_tuning(key, v)
                assert devutil.get_tuning(key) == v
                assert L.cnt_tuning_name(key.encode(), v)
            assert L.cnt_set_tuning(key.encode(), n) == 1
            assert L.cnt_set_tuning(key.encode(), -1) == 1
            assert L.cnt_tuning_name(key.encode(), n) is None
        finally:
            devutil.set_tuning(key, old)
    assert L.cnt_set_tuning(b"nonsense", 0) == 1


def test_host_tier_fails_loudly_without_a_gpu(L):
    """On a box with no HIP device the product path errors out; it never computes on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cute_nucleotides_amd import n_to_bits_hip
    from cute_nucleotides_amd._lib import CuteNtError

    with pytest.raises(CuteNtError):
        n_to_bits_hip(b"ACGT")


def test_python_mirror_len_guard_is_a_value_error():
    from cute_nucleotides_amd import bits_to_n2_hip, bits_to_n_hip

    with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
        bits_to_n_hip(np.zeros(1, dtype=np.uint64), 33)
    with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
        bits_to_n2_hip(np.zeros(1, dtype=np.uint64), 28)


def test_c_partition_equals_the_python_partition(L):
    """cnt_shard_range (what cnt_*_sharded cut by) against cute_nucleotides_amd/sharding.py (what bench.py's
    ranks cut by): same ranges for both codecs, every shard on a word boundary, only the last non-empty one
    ragged, empty shards at (n, n), full coverage.  Pure arithmetic: runs without a device."""
    from cute_nucleotides_amd import _lib, sharding

    sizes = [0, 1, 26, 27, 31, 32, 33, 13823, 13824, 13825, 16383, 16384, 16385, 100003, 8 * 16384, 8 * 13824 + 1,
             (1 << 20) + 13, 1 << 34, (1 << 35) * 8, (1 << 36) + 7, (1 << 62) + 12345]
    for unit, gran in ((32, sharding.SHARD_GRAN_NT), (27, sharding.SHARD_GRAN5_NT)):
        assert gran % unit == 0
        for n_len in sizes:
            for ndev in (1, 2, 3, 4, 5, 7, 8, 64):
                c = [sharding.shard_range_c(n_len, ndev, k, unit) for k in range(ndev)]
                assert c == sharding.partition(n_len, ndev, gran), (unit, n_len, ndev)
                assert c[0][0] == 0 and c[-1][1] == n_len
                nonempty = [(lo, hi) for lo, hi in c if hi > lo]
                for lo, hi in nonempty:
                    assert lo % unit == 0 and lo % gran == 0
                for lo, hi in nonempty[:-1]:
                    assert (hi - lo) % gran == 0
                for (lo, hi), (lo2, hi2) in zip(c, c[1:]):
                    assert hi == lo2
                # output words: shard k writes [lo/unit, lo/unit + ceil((hi-lo)/unit)) -- disjoint, covering ceil(n/unit)
                w = [(lo // unit, lo // unit + -(-(hi - lo) // unit)) for lo, hi in nonempty]
                for (a, b), (c2, d) in zip(w, w[1:]):
                    assert b == c2
                if nonempty:
                    assert w[-1][1] == -(-n_len // unit)
    # the metric / configs[4] shapes: 8 x 2^35 and 8 x 2^34 cut exactly
    assert [sharding.shard_range_c(8 << 35, 8, k) for k in range(8)] == [(k << 35, (k + 1) << 35) for k in range(8)]
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    for bad in ((10, 0, 0, 32), (10, 2, 2, 32), (10, 2, -1, 32), (10, 2, 0, 16)):
        assert L.cnt_shard_range(*bad, ctypes.byref(lo), ctypes.byref(hi)) == _lib.CNT_EINVAL
    assert L.cnt_shard_range(10, 2, 0, 32, None, ctypes.byref(hi)) == _lib.CNT_EINVAL
    # saturating near SIZE_MAX instead of wrapping
    assert sharding.shard_range_c(2**64 - 1, 1, 0) == (0, 2**64 - 1)
    assert sharding.shard_range_c(2**64 - 1, 3, 2)[1] == 2**64 - 1


def test_worker_info_needs_a_sharded_call_first(L):
    from cute_nucleotides_amd import _lib

    assert L.cnt_shard_worker_info(0, None, None, None, None) == _lib.CNT_EINVAL


def test_every_tier_fails_loudly_without_a_device(L):
    """There is no CPU fallback: on a box without a HIP device every compute entry point returns an error status
    (CNT_ENODEV from the host / sharded tiers, -(hipErrorNoDevice) from the device tier) -- it never computes on the
    host and never crashes.  Argument errors and empty inputs are still answered."""
    from cute_nucleotides_amd import _lib

    count = ctypes.c_int(-1)
    assert L.cnt_device_count(ctypes.byref(count)) == 0
    if count.value != 0:
        pytest.skip("a HIP device is visible here; this test is about the box without one")
    n = np.frombuffer(b"ACGT" * 16, dtype=np.uint8)
    out = np.full(2, 0x5555555555555555, dtype=np.uint64)
    buf = np.full(64, 0x2A, dtype=np.uint8)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    for rc in (L.cnt_n_to_bits(p(n), 64, p(out), 2), L.cnt_bits_to_n(p(out), 2, 64, p(buf)), L.cnt_n_to_bits2(p(n), 64, p(out), 3),
               L.cnt_bits_to_n2(p(out), 2, 54, p(buf)), L.cnt_n_to_bits_sharded(p(n), 64, p(out), 2, 2), L.cnt_bits_to_n2_sharded(p(out), 2, 54, p(buf), 0)):
        assert rc == _lib.CNT_ENODEV, rc
    assert L.cnt_n_to_bits_dev(p(n), 64, p(out), 2, 0, None) < 0 and L.cnt_round_trip_dev(p(n), 64, p(out), 2, p(buf), 0, None) != 0
    ptrs, sizes = (ctypes.c_void_p * 1)(n.ctypes.data), (ctypes.c_size_t * 1)(64)
    outs, caps = (ctypes.c_void_p * 1)(out.ctypes.data), (ctypes.c_size_t * 1)(2)
    assert L.cnt_n_to_bits_sharded_dev(ptrs, sizes, outs, caps, 1, 0, None) == _lib.CNT_ENODEV
    d = ctypes.c_void_p()
    assert L.cnt_dev_alloc(ctypes.byref(d), 64) < 0 and not d.value
    assert L.cnt_device_pci_bus_id(0, ctypes.create_string_buffer(64), 64) == _lib.CNT_ENODEV
    assert (out == 0x5555555555555555).all() and (buf == 0x2A).all()  # nothing was computed anywhere
    # still answered without a device: empty inputs, argument errors, the partition, shutdown
    assert L.cnt_n_to_bits(None, 0, None, 0) == 0 and L.cnt_bits_to_n_sharded(None, 0, 0, None, 4) == 0
    assert L.cnt_n_to_bits(p(n), 64, p(out), 1) == _lib.CNT_ECAP and L.cnt_bits_to_n(p(out), 1, 33, p(buf)) == _lib.CNT_ELEN
    assert L.cnt_shutdown() == 0
    with pytest.raises(_lib.CuteNtError):
        import cute_nucleotides_amd as cn

        cn.n_to_bits_hip(b"ACGT")
