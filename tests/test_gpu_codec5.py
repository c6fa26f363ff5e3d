"""GPU parity tests for the 5-letter {A,C,G,T/U,N} codec (reference src/n_to_bits2.rs) against
the CPU oracle; bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALPHA = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
SIZES = [1, 2, 3, 4, 5, 26, 27, 28, 53, 54, 55, 80, 81, 82, 6911, 6912, 6913, 6912 * 3 + 5, 100003, 27 * 40000,
         (1 << 22) + 11]


@pytest.fixture()
def no_small_path(lab_build):
    """small ragged inputs normally take the generic kernel alone: switch that off to reach the tiles"""
    from cute_nucleotides_amd import devutil

    saved = devutil.get_tuning("small_nt")
    devutil.set_tuning("small_nt", 0)
    yield
    devutil.set_tuning("small_nt", saved)


@pytest.fixture(scope="module")
def cn():
    import torch

    assert torch.cuda.is_available()
    import cute_nucleotides_amd as cn

    return cn


@pytest.mark.parametrize("n_len", SIZES)
def test_encode_decode_host_tier(cn, oracle, n_len):
    n = ALPHA[np.random.default_rng(n_len).integers(0, ALPHA.size, n_len)]
    want = oracle.n_to_bits2_lut(n)
    got = cn.n_to_bits2_hip(n)
    assert np.array_equal(got, want)
    back = cn.bits_to_n2_hip(got, n_len)
    assert np.array_equal(back, oracle.bits_to_n2_lut(want, n_len))
    assert bytes(back) == bytes(n).upper().replace(b"U", b"T")


@pytest.mark.parametrize("small_nt", [0, 1 << 17], ids=["lab: tiles at every size", "product build"])
def test_device_tier_aligned_unaligned_and_overrun(cn, oracle, request, small_nt):
    import torch

    from cute_nucleotides_amd import devutil

    if small_nt == 0:  # forcing the tile path for small inputs needs the lab build's knob; the other case is the product as shipped
        request.getfixturevalue("lab_build")
        devutil.set_tuning("small_nt", 0)
    n_len = 6912 * 5 + 100
    n = oracle.fill_random_acgtn(n_len, 5)
    want = oracle.n_to_bits2_lut(n)
    buf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
    for off in (0, 1, 16, 27):
        v = buf[off : off + n_len]
        v.copy_(torch.from_numpy(n))
        out = torch.full((want.size + 3,), -1, dtype=torch.int64, device="cuda")
        cn.n_to_bits2_dev(v, out=out)
        got = out.cpu().numpy()
        assert (got[want.size :] == -1).all()
        assert np.array_equal(got[: want.size].view(np.uint64), want), off
    dbits = torch.from_numpy(want.view(np.int64)).cuda()
    for length in (0, 1, 26, 27, 28, 6912, 6913, n_len - 1, n_len):
        for off in (0, 3):
            out = torch.full((n_len + 64,), 0x5A, dtype=torch.uint8, device="cuda")
            cn.bits_to_n2_dev(dbits, length, out=out[off:])
            got = out.cpu().numpy()
            assert (got[off + length :] == 0x5A).all() and (got[:off] == 0x5A).all()
            assert np.array_equal(got[off : off + length], oracle.bits_to_n2_lut(want, length)), (length, off)
    with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
        cn.bits_to_n2_dev(dbits, want.size * 27 + 1)
    if small_nt == 0:
        devutil.set_tuning("small_nt", 1 << 17)


@pytest.mark.parametrize("strict", [False, True])
def test_encode_alignment_matrix(cn, oracle, no_small_path, strict):
    """Any input byte phase x output word phase (head peel + n_to_bits2_window), sizes around the
    peel / tile / slack boundaries, guard words around the output."""
    import torch

    sizes = [3456 + 27 * 15 + 127, 3456 + 27 * 15 + 128 + 27, 3456 * 3 + 500, 40003, 100003]
    if strict:
        big = np.random.default_rng(31).integers(0, 256, max(sizes), dtype=np.uint8)
    else:
        big = ALPHA[np.random.default_rng(30).integers(0, ALPHA.size, max(sizes))]
    ibuf = torch.zeros(big.size + 256, dtype=torch.uint8, device="cuda")
    obuf = torch.empty(big.size // 27 + 64, dtype=torch.int64, device="cuda")
    for n_len in sizes:
        n = big[:n_len]
        want = oracle.n_to_bits2_lut(n)
        for io in [0, 1, 2, 3, 5, 8, 15, 16, 17, 27, 33, 64, 77, 100, 127]:
            view = ibuf[io : io + n_len]
            view.copy_(torch.from_numpy(n))
            for oo in (0, 1, 2, 3, 5, 7):
                obuf.fill_(-1)
                cn.n_to_bits2_dev(view, out=obuf[8 + oo : 8 + oo + want.size], strict_lut=strict)
                got = obuf.cpu().numpy()
                assert (got[: 8 + oo] == -1).all() and (got[8 + oo + want.size :] == -1).all(), (n_len, io, oo)
                assert np.array_equal(got[8 + oo : 8 + oo + want.size].view(np.uint64), want), (n_len, io, oo)


def test_decode_alignment_matrix(cn, oracle, no_small_path):
    """Every output phase mod 128 (the head is 19 * (-phase) mod 128 whole words) x packed-word phase."""
    import torch

    words = 1500
    bits = np.random.default_rng(32).integers(0, 2**63, words, dtype=np.uint64)
    want_full = oracle.bits_to_n2_lut(bits, words * 27)
    dbuf = torch.zeros(words + 8, dtype=torch.int64, device="cuda")
    obuf = torch.empty(words * 27 + 512, dtype=torch.uint8, device="cuda")
    for wo in (0, 1):
        d = dbuf[wo : wo + words]
        d.copy_(torch.from_numpy(bits.view(np.int64)))
        for oo in range(128):
            for length in (words * 27, words * 27 - 3456 - 5, 3456 + 127 * 27 + 1, 3456 + 127 * 27 - 1):
                obuf.fill_(0x2A)
                cn.bits_to_n2_dev(d, length, out=obuf[128 + oo : 128 + oo + length])
                got = obuf.cpu().numpy()
                assert (got[: 128 + oo] == 0x2A).all() and (got[128 + oo + length :] == 0x2A).all(), (wo, oo, length)
                assert np.array_equal(got[128 + oo : 128 + oo + length], want_full[:length]), (wo, oo, length)


def test_sharded_tier_single_device(cn, oracle):
    from cute_nucleotides_amd import n_to_bits2 as n2

    for n_len in (1, 3456 * 4 - 1, 3456 * 4, 3456 * 9 + 5, (1 << 21) + 7):
        n = ALPHA[np.random.default_rng(n_len).integers(0, ALPHA.size, n_len)]
        want = oracle.n_to_bits2_lut(n)
        got = n2.n_to_bits2_hip_sharded(n, ndev=1)
        assert np.array_equal(got, want)
        assert np.array_equal(n2.bits_to_n2_hip_sharded(got, n_len, ndev=0), oracle.bits_to_n2_lut(want, n_len))
    with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
        n2.bits_to_n2_hip_sharded(got, got.size * 27 + 1)


def test_arbitrary_words_and_bytes(cn, oracle):
    """Words no encoder produces (7-bit fields 125..127, bit 63 set) decode like the oracle
    defines; strict mode encodes non-alphabet bytes as 0 like BYTE_LUT (n_to_bits2.rs:8-23)."""
    import torch

    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2**64, 256 * 7 + 13, dtype=np.uint64)
    d = torch.from_numpy(bits.view(np.int64)).cuda()
    got = cn.bits_to_n2_dev(d, bits.size * 27).cpu().numpy()
    assert np.array_equal(got, oracle.bits_to_n2_lut(bits, bits.size * 27))
    n = rng.integers(0, 256, 6912 * 2 + 40, dtype=np.uint8)
    dn = torch.from_numpy(n).cuda()
    strict = cn.n_to_bits2_dev(dn, strict_lut=True).cpu().numpy().view(np.uint64)
    assert np.array_equal(strict, oracle.n_to_bits2_lut(n))
    fast = cn.n_to_bits2_dev(dn).cpu().numpy().view(np.uint64)
    table = np.zeros(256, dtype=np.uint8)  # the reference SIMD table on the low 3 bits (n_to_bits2.rs:127-136)
    for c in range(128):
        table[c] = {1: 0, 3: 1, 4: 2, 5: 2, 6: 4, 7: 3}.get(c & 7, 0)
    codes = table[n].astype(np.uint64)
    codes = np.concatenate([codes, np.zeros((-codes.size) % 27, dtype=np.uint64)]).reshape(-1, 9, 3)
    vals = codes[:, :, 0] + 5 * codes[:, :, 1] + 25 * codes[:, :, 2]
    want = np.bitwise_or.reduce(vals << (np.arange(9, dtype=np.uint64) * np.uint64(7)), axis=1)
    assert np.array_equal(fast, want)


def test_large_round_trip(cn, oracle):
    import torch

    from cute_nucleotides_amd import devutil

    n_len = 27 * (1 << 22) + 5  # ~108 Mi nt
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgtn(d, 11)
    packed = cn.n_to_bits2_dev(d)
    back = cn.bits_to_n2_dev(packed, n_len)
    assert devutil.count_mismatch(d, back) == 0
    m = 27 * 100000
    host = oracle.fill_random_acgtn(m, 11)
    assert np.array_equal(packed[: m // 27].cpu().numpy().view(np.uint64), oracle.n_to_bits2_lut(host))


@pytest.mark.parametrize("key", ["encode2", "decode2"])
def test_every_variant(cn, oracle, lab_build, key):
    import torch

    from cute_nucleotides_amd import devutil

    n_len = 1728 * 64 * 9 + 1728 * 3 + 11  # whole wave tiles for every workgroup width + a ragged rest
    n = oracle.fill_random_acgtn(n_len, 21)
    want = oracle.n_to_bits2_lut(n)
    d = torch.from_numpy(n).cuda()
    dbits = torch.from_numpy(want.view(np.int64)).cuda()
    old = devutil.get_tuning(key)
    assert old == 0
    try:
        for v, name in devutil.variants(key):
            devutil.set_tuning(key, v)
            if key == "encode2":
                for strict in (False, True):
                    got = cn.n_to_bits2_dev(d, strict_lut=strict).cpu().numpy().view(np.uint64)
                    assert np.array_equal(got, want), (name, strict)
            else:
                for length in (n_len, n_len - 1, 1728 * 5, 1728 * 5 + 26):
                    got = cn.bits_to_n2_dev(dbits, length).cpu().numpy()
                    assert np.array_equal(got, oracle.bits_to_n2_lut(want, length)), (name, length)
    finally:
        devutil.set_tuning(key, old)


def test_large_ragged_size_64bit_indexing(cn, oracle):
    import torch

    from cute_nucleotides_amd import devutil

    n_len = 27 * ((1 << 28) + 12345) + 19  # ~7.2 Gi nt, > 2^32, ragged
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgtn(d, 31)
    packed = cn.n_to_bits2_dev(d)
    back = cn.bits_to_n2_dev(packed, n_len)
    assert devutil.count_mismatch(d, back) == 0
    tail_words = 5000
    first_word = packed.numel() - tail_words
    host = oracle.fill_random_acgtn(n_len - 27 * first_word, 31, first_nt=27 * first_word)
    assert np.array_equal(packed[first_word:].cpu().numpy().view(np.uint64), oracle.n_to_bits2_lut(host))


def test_caller_supplied_outputs_are_validated(cn):
    """The C ABI counts output capacity in WORDS and the decoders take no capacity at all, so the Python
    device tier must refuse an `out` that is too narrow, too short, non-contiguous or on the host -- both
    codecs, the fused call and the packed-domain ops (a uint8 `out` with numel == words would pass the C
    check and be written 8x out of bounds)."""
    import torch

    from cute_nucleotides_amd import n_to_bits2 as n2
    from cute_nucleotides_amd import packed_ops as po

    n = torch.full((27 * 100,), 65, dtype=torch.uint8, device="cuda")
    words5, words2 = 100, (27 * 100 + 31) // 32
    ok5 = n2.n_to_bits2_dev(n, out=torch.empty(words5 + 3, dtype=torch.int64, device="cuda"))
    assert ok5.numel() == words5
    bad_word_outs = [torch.empty(words5, dtype=torch.uint8, device="cuda"), torch.empty(words5 - 1, dtype=torch.int64, device="cuda"),
                     torch.empty(2 * words5, dtype=torch.int64, device="cuda")[::2], torch.empty(words5, dtype=torch.int64)]
    for bad in bad_word_outs:
        with pytest.raises(ValueError):
            n2.n_to_bits2_dev(n, out=bad)
        with pytest.raises(ValueError):
            cn.n_to_bits_dev(n, out=bad[: words2 - 1] if (bad.is_cuda and bad.is_contiguous() and bad.dtype == torch.int64) else bad)
    bad_byte_outs = [torch.empty(2700, dtype=torch.int8, device="cuda"), torch.empty(2699, dtype=torch.uint8, device="cuda"),
                     torch.empty(5400, dtype=torch.uint8, device="cuda")[::2], torch.empty(2700, dtype=torch.uint8)]
    bits2 = cn.n_to_bits_dev(n)
    for bad in bad_byte_outs:
        with pytest.raises(ValueError):
            n2.bits_to_n2_dev(ok5, 2700, out=bad)
        with pytest.raises(ValueError):
            cn.bits_to_n_dev(bits2, 2700, out=bad)
        with pytest.raises(ValueError):
            cn.round_trip_dev(n, out_n=bad)
    for bad in bad_word_outs[:1] + bad_word_outs[2:]:
        with pytest.raises(ValueError):
            po.complement_dev(bits2, 2700, out=bad)
        with pytest.raises(ValueError):
            po.reverse_complement_dev(bits2, 2700, out=bad)
        with pytest.raises(ValueError):
            cn.round_trip_dev(n, out_bits=bad)
    with pytest.raises(ValueError):
        po.complement_dev(bits2, 2700, out=torch.empty(words2 - 1, dtype=torch.int64, device="cuda"))
    # and the calling thread's HIP device is where it was
    assert torch.cuda.current_device() == 0
    from cute_nucleotides_amd import _lib
    import ctypes

    cur = ctypes.c_int(-1)
    assert _lib.lib().cnt_get_device(ctypes.byref(cur)) == 0 and cur.value == 0


# ---- CNT_TAIL_LUT: n_to_bits2_pext to the letter, on arbitrary bytes --------------------------------------------------
TAIL_SIZES5 = [1, 4, 5, 6, 26, 27, 28, 31, 32, 33, 53, 54, 58, 59, 60, 81, 3456, 3456 + 4, 3456 + 5, 3456 * 3, 3456 * 3 + 31, 3456 * 3 + 32,
               100003, 27 * 40000, 27 * 40000 + 5, (1 << 20) + 11, 27 * (1 << 19) + 13824 * 2, 27 * (1 << 19) + 13824 * 2 + 4]


def test_tail_lut_equals_n_to_bits2_pext_on_arbitrary_bytes(cn, oracle, lab_build):
    """n_to_bits2_pext runs its low-3-bit table over words [0, (len-5)/27) -- its 32-byte loads would over-read 5 bytes
    beyond that -- and hands every later word (one or two, whole or ragged) to n_to_bits2_lut (n_to_bits2.rs:120,179-185).
    On foreign bytes the two tables differ ('B' = 0x42 has low bits 010 -> 0 in both, but 'D' = 0x44 -> T in the fast
    table, 0 in BYTE_LUT), so neither the default nor CNT_STRICT_LUT alone equals it at every length; default |
    CNT_TAIL_LUT does: host tier (small path + chunked pipeline), device tier (tiles + generic), sharded tier."""
    import torch

    from cute_nucleotides_amd import devutil, n_to_bits2 as n2, sharding

    if not oracle.port_cpu_ok():
        pytest.skip("host CPU lacks AVX2/BMI2: the SIMD port cannot run")
    rng = np.random.default_rng(77)
    saved = devutil.get_tuning("small_nt")
    prev = sharding.alias_devices(True)
    try:
        for n_len in TAIL_SIZES5:
            n = rng.integers(0, 128, n_len, dtype=np.uint8)
            want = oracle.n_to_bits2_pext(n)
            assert np.array_equal(n2.n_to_bits2_hip(n, tail_lut=True), want), n_len
            d = torch.from_numpy(n).cuda()
            for small_nt in (0, 1 << 17):
                devutil.set_tuning("small_nt", small_nt)
                assert np.array_equal(n2.n_to_bits2_dev(d, tail_lut=True).cpu().numpy().view(np.uint64), want), (n_len, small_nt)
            off = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")  # input phase != 0: the window kernel's launch
            off[5 : 5 + n_len].copy_(d)
            assert np.array_equal(n2.n_to_bits2_dev(off[5 : 5 + n_len], tail_lut=True).cpu().numpy().view(np.uint64), want), n_len
            for ndev in (1, 3, 8):
                assert np.array_equal(n2.n_to_bits2_hip_sharded(n, ndev=ndev, tail_lut=True), want), (n_len, ndev)
            assert np.array_equal(n2.n_to_bits2_hip(n, strict_lut=True, tail_lut=True), oracle.n_to_bits2_lut(n)), n_len
    finally:
        devutil.set_tuning("small_nt", saved)
        sharding.alias_devices(prev)


def test_one_launch_per_call_at_any_size_and_alignment(cn, oracle, no_small_path):
    """the 5-letter codec's default kernels carry their head words and ragged end themselves (the last workgroups of the
    grid, behind their tile's stores -- the 2-bit codec's scheme): ONE node in a captured graph, results still the oracle's"""
    import torch

    from test_gpu_codec2 import _kernel_nodes_of

    n_len = 3456 * 300 + 1234
    host = oracle.fill_random_acgtn(n_len, 9)
    want = oracle.n_to_bits2_lut(host)
    want_back = oracle.bits_to_n2_lut(want, n_len)
    ibuf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
    pbuf = torch.zeros(want.size + 64, dtype=torch.int64, device="cuda")
    obuf = torch.zeros(n_len + 512, dtype=torch.uint8, device="cuda")
    for io, po, oo in ((0, 0, 0), (0, 3, 0), (5, 0, 77), (64, 1, 127), (127, 7, 1), (16, 0, 16)):
        view = ibuf[io : io + n_len]
        view.copy_(torch.from_numpy(host))
        packed = pbuf[po : po + want.size]
        out = obuf[oo : oo + n_len]
        assert _kernel_nodes_of(torch, lambda: cn.n_to_bits2_dev(view, out=packed)) == 1, (io, po)
        assert np.array_equal(packed.cpu().numpy().view(np.uint64), want), (io, po)
        assert _kernel_nodes_of(torch, lambda: cn.bits_to_n2_dev(packed, n_len, out=out)) == 1, (po, oo)
        assert np.array_equal(out.cpu().numpy(), want_back), (po, oo)
