"""Randomised differential test of the HIP path against the oracle: random lengths (biased to
tile edges), random 16-B/8-B/1-B pointer offsets, both codecs, both encode modes, device tier.
Deterministic seeds; a few hundred cases in a few seconds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# CNT_FUZZ_SEEDS=N multiplies the number of seeds of every test in this file (soak runs: `CNT_FUZZ_SEEDS=50
# python -m pytest tests/test_gpu_fuzz.py -m gpu -q`, a few minutes; profiles/r02_fuzz_soak.log is one such run)
SEEDS = max(1, int(os.environ.get("CNT_FUZZ_SEEDS", "1")))

EDGES = [0, 1, 2, 3, 26, 27, 31, 32, 33, 63, 64, 2047, 2048, 2049, 3455, 3456, 3457, 4095, 4096, 4097,
         32767, 32768, 32769, 65535, 65536]


def _lengths(rng, k):
    out = []
    for _ in range(k):
        base = int(rng.choice(EDGES)) * int(rng.integers(1, 40))
        out.append(max(0, base + int(rng.integers(-3, 4))))
    return out


@pytest.fixture(params=[0, 1 << 17], ids=["tiles", "small-path"])
def small_nt(request):
    """with the single-launch path for small ragged inputs (the product build as shipped) and without it (lab build,
    tuning key small_nt = 0: the tile kernels and their edge workgroups at every size)"""
    if request.param:
        yield request.param
        return
    request.getfixturevalue("lab_build")
    from cute_nucleotides_amd import devutil

    saved = devutil.get_tuning("small_nt")
    devutil.set_tuning("small_nt", 0)
    yield 0
    devutil.set_tuning("small_nt", saved)


@pytest.mark.parametrize("seed", range(4 * SEEDS))
def test_two_bit_codec_fuzz(oracle, small_nt, seed):
    import torch

    import cute_nucleotides_amd as cn

    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    for n_len in _lengths(rng, 40):
        off_in, off_out = int(rng.choice([0, 0, 16, 48, 8, 1, 5])), int(rng.choice([0, 0, 1, 2]))  # out offset in words
        anybytes = bool(rng.integers(0, 3) == 0)
        n = rng.integers(0, 256, n_len, dtype=np.uint8) if anybytes else alpha[rng.integers(0, 10, n_len)]
        buf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
        view = buf[off_in : off_in + n_len]
        view.copy_(torch.from_numpy(n))
        words = (n_len + 31) // 32
        obuf = torch.full((words + 4,), -1, dtype=torch.int64, device="cuda")
        strict = bool(rng.integers(0, 2))
        tail = not strict and bool(rng.integers(0, 2))  # CNT_TAIL_LUT: the SIMD encoders to the letter (table on the final partial word)
        if tail and anybytes:
            n = n & 0x7F  # the reference indexes BYTE_LUT[128] with the tail's bytes: 7-bit input only
            view.copy_(torch.from_numpy(n))
        got = cn.n_to_bits_dev(view, out=obuf[off_out:], strict_lut=strict, tail_lut=tail).cpu().numpy().view(np.uint64)
        if strict or not anybytes:
            want = oracle.n_to_bits_lut(n)
        elif tail:
            want = oracle.n_to_bits_bitextract(n)  # bits 1..2 on whole 32-nt blocks, BYTE_LUT on the ragged end (n_to_bits.rs:96-111)
        else:  # default mode on arbitrary bytes: (byte>>1)&3 everywhere, tail included
            codes = ((n >> 1) & 3).astype(np.uint64)
            pad = np.zeros((-n_len) % 32, dtype=np.uint64)
            want = np.bitwise_or.reduce(np.concatenate([codes, pad]).reshape(-1, 32) << (np.arange(32, dtype=np.uint64) * np.uint64(2)), axis=1) if words else np.empty(0, np.uint64)
        assert np.array_equal(got, want), (seed, n_len, off_in, off_out, strict, tail, anybytes)
        o = obuf.cpu().numpy()
        assert (o[:off_out] == -1).all() and (o[off_out + words :] == -1).all()
        # decode a random prefix of what was just packed, into an offset output
        length = int(rng.integers(0, n_len + 1))
        dout = torch.full((n_len + 80,), 0x5A, dtype=torch.uint8, device="cuda")
        off_d = int(rng.choice([0, 0, 16, 7]))
        cn.bits_to_n_dev(obuf[off_out : off_out + words], length, out=dout[off_d:])
        d = dout.cpu().numpy()
        assert np.array_equal(d[off_d : off_d + length], oracle.bits_to_n_lut(want, length)), (seed, n_len, length)
        assert (d[:off_d] == 0x5A).all() and (d[off_d + length :] == 0x5A).all()


@pytest.mark.parametrize("seed", range(3 * SEEDS))
def test_five_letter_codec_fuzz(oracle, small_nt, seed):
    import torch

    import cute_nucleotides_amd as cn

    rng = np.random.default_rng(100 + seed)
    alpha = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
    for n_len in _lengths(rng, 30):
        off_in = int(rng.choice([0, 0, 16, 27, 1]))
        n = alpha[rng.integers(0, 12, n_len)]
        buf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
        view = buf[off_in : off_in + n_len]
        view.copy_(torch.from_numpy(n))
        want = oracle.n_to_bits2_lut(n)
        got = cn.n_to_bits2_dev(view, strict_lut=bool(rng.integers(0, 2))).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), (seed, n_len, off_in)
        if oracle.port_cpu_ok() and n_len:  # CNT_TAIL_LUT on arbitrary 7-bit bytes == the port of n_to_bits2_pext
            junk = rng.integers(0, 128, n_len, dtype=np.uint8)
            view.copy_(torch.from_numpy(junk))
            got = cn.n_to_bits2_dev(view, tail_lut=True).cpu().numpy().view(np.uint64)
            assert np.array_equal(got, oracle.n_to_bits2_pext(junk)), (seed, n_len, off_in, "tail_lut")
        length = int(rng.integers(0, n_len + 1))
        dout = torch.full((n_len + 80,), 0x5A, dtype=torch.uint8, device="cuda")
        off_d = int(rng.choice([0, 0, 16, 3]))
        cn.bits_to_n2_dev(torch.from_numpy(want.view(np.int64)).cuda(), length, out=dout[off_d:])
        d = dout.cpu().numpy()
        assert np.array_equal(d[off_d : off_d + length], oracle.bits_to_n2_lut(want, length)), (seed, n_len, length)
        assert (d[:off_d] == 0x5A).all() and (d[off_d + length :] == 0x5A).all()


@pytest.mark.parametrize("seed", range(2 * SEEDS))
def test_large_decode_fuzz_over_packed_page_offsets(oracle, seed):
    """Calls of >= 2^20 nt peel the output to a 4-KiB page; the 0-3 FURTHER pages that place the XCD turns on the packed buffer's
    pages, and the window kernel, start above cnt_chip_cache_nt (2^30 nt on an SPX MI355X), so at these sizes the product runs
    the page peel + stream / shifted kernels (odd seeds: the lab build with the gate forced to 1 nt, i.e. turn placement +
    bits_to_n_window): random packed word offsets inside a page x random output byte offsets x random lengths, guards around the
    output, against bits_to_n_lut."""
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib, devutil

    if seed % 2:  # the past-the-cache plan at a size that fits any box: lab build, gate forced open
        prev = _lib.use_lab_build(True)
        devutil.set_tuning("decode_cache_log2", 0)
    try:
        _large_decode_fuzz(oracle, seed, torch, cn)
    finally:
        if seed % 2:
            devutil.set_tuning("decode_cache_log2", -1)
            _lib.use_lab_build(prev)


def _large_decode_fuzz(oracle, seed, torch, cn):
    rng = np.random.default_rng(9000 + seed)
    cap_words = ((1 << 21) + 70000) // 32
    bits = rng.integers(0, 2**64, cap_words, dtype=np.uint64)
    want = oracle.bits_to_n_lut(bits, cap_words * 32)
    pbuf = torch.zeros(cap_words + 1024, dtype=torch.int64, device="cuda")
    obuf = torch.empty(cap_words * 32 + 3 * 4096, dtype=torch.uint8, device="cuda")
    pb = ((-pbuf.data_ptr()) % 4096) // 8
    ob = 4096 + (-obuf.data_ptr()) % 4096
    for _ in range(6):
        p_off = int(rng.integers(0, 512))  # words: the packed pointer anywhere inside its page
        o_off = int(rng.integers(0, 4096))
        length = int(rng.integers(1 << 20, cap_words * 32 + 1))
        d = pbuf[pb + p_off : pb + p_off + cap_words]
        d.copy_(torch.from_numpy(bits.view(np.int64)))
        obuf.fill_(0x5A)
        cn.bits_to_n_dev(d, length, out=obuf[ob + o_off : ob + o_off + length])
        got = obuf.cpu().numpy()
        assert np.array_equal(got[ob + o_off : ob + o_off + length], want[:length]), (seed, p_off, o_off, length)
        assert (got[: ob + o_off] == 0x5A).all() and (got[ob + o_off + length :] == 0x5A).all(), (seed, p_off, o_off, length)


@pytest.fixture()
def alias(hooks_build):
    """the test-hooks build with cnt_test_alias_devices(1) for one test: shard k -> device k % count on the 1-GPU box (the
    product library has no such switch)"""
    from cute_nucleotides_amd import sharding

    sharding.alias_devices(True)
    yield
    sharding.alias_devices(False)


@pytest.mark.parametrize("seed", range(3 * SEEDS))
def test_host_and_sharded_tiers_fuzz(oracle, seed, alias):
    """host-slice entry points: the zero-copy small path (staged kernels + completion flag), the 2-slot pipeline, and
    the sharded tier with a random number of aliased shards -- random lengths around every path's limits, both codecs,
    arbitrary bytes for the 2-bit encoder in both modes"""
    import ctypes

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib, n_to_bits2 as n2

    L = _lib.lib()
    rng = np.random.default_rng(500 + seed)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    alpha5 = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
    limits = [1, 27, 32, 4096, 13824, 16384, 40000, 1 << 20, (1 << 20) + 1, 1 << 22, 3 << 20]
    for _ in range(10):
        n_len = max(1, int(rng.choice(limits)) + int(rng.integers(-40, 41)))
        strict = bool(rng.integers(0, 2))
        n = rng.integers(0, 256, n_len, dtype=np.uint8) if strict else alpha[rng.integers(0, 10, n_len)]
        want = oracle.n_to_bits_lut(n)
        assert np.array_equal(cn.n_to_bits_hip(n, strict_lut=strict), want), (seed, n_len, strict)
        length = int(rng.integers(0, n_len + 1))
        assert np.array_equal(cn.bits_to_n_hip(want, length), oracle.bits_to_n_lut(want, length)), (seed, n_len, length)
        ndev = int(rng.integers(1, 10))
        v = alpha[rng.integers(0, 10, n_len)]
        wv = oracle.n_to_bits_lut(v)
        assert np.array_equal(cn.n_to_bits_hip_sharded(v, ndev=ndev), wv), (seed, n_len, ndev)
        assert np.array_equal(cn.bits_to_n_hip_sharded(wv, length, ndev=ndev), oracle.bits_to_n_lut(wv, length)), (seed, n_len, ndev, length)
        n5 = alpha5[rng.integers(0, 12, n_len)]
        w5 = oracle.n_to_bits2_lut(n5)
        assert np.array_equal(n2.n_to_bits2_hip(n5), w5), (seed, n_len)
        assert np.array_equal(n2.n_to_bits2_hip_sharded(n5, ndev=ndev), w5), (seed, n_len, ndev)
        assert np.array_equal(n2.bits_to_n2_hip_sharded(w5, length, ndev=ndev), oracle.bits_to_n2_lut(w5, length)), (seed, n_len, ndev, length)
        # capacity / length errors still come first
        assert L.cnt_n_to_bits_sharded(ctypes.c_void_p(v.ctypes.data), n_len, ctypes.c_void_p(wv.ctypes.data), max(wv.size - 1, 0), ndev) == _lib.CNT_ECAP


@pytest.mark.parametrize("seed", range(2 * SEEDS))
def test_fused_and_packed_ops_fuzz(oracle, seed):
    """fused round trip == encode then decode (oracle), and the packed-domain ops against their oracle definitions,
    on random lengths and pointer phases"""
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import packed_ops as po

    rng = np.random.default_rng(900 + seed)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    for n_len in _lengths(rng, 12):
        if n_len == 0:
            continue
        strict = bool(rng.integers(0, 2))
        n = rng.integers(0, 256, n_len, dtype=np.uint8) if strict else alpha[rng.integers(0, 10, n_len)]
        off = int(rng.choice([0, 0, 128, 1, 16]))
        buf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
        view = buf[off : off + n_len]
        view.copy_(torch.from_numpy(n))
        # all three pointers at random phases (round_trip_window off the 128-B grid), guards around both outputs
        po_w, bo = int(rng.integers(0, 17)), int(rng.choice([0, 0, 128, 1, 16, 77, 4095]))
        words = (n_len + 31) // 32
        pbuf = torch.full((words + po_w + 8,), -1, dtype=torch.int64, device="cuda")
        bbuf = torch.full((n_len + bo + 64,), 0x2A, dtype=torch.uint8, device="cuda")
        bits, back = cn.round_trip_dev(view, out_bits=pbuf[po_w : po_w + words], out_n=bbuf[bo : bo + n_len], strict_lut=strict)
        want = oracle.n_to_bits_lut(n)
        assert np.array_equal(bits.cpu().numpy().view(np.uint64), want), (seed, n_len, off, po_w, bo, strict)
        assert np.array_equal(back.cpu().numpy(), oracle.bits_to_n_lut(want, n_len)), (seed, n_len, off, po_w, bo, strict)
        pb, bb = pbuf.cpu().numpy(), bbuf.cpu().numpy()
        assert (pb[:po_w] == -1).all() and (pb[po_w + words :] == -1).all() and (bb[:bo] == 0x2A).all() and (bb[bo + n_len :] == 0x2A).all(), (seed, n_len)
        other = oracle.n_to_bits_lut(alpha[rng.integers(0, 10, n_len)])
        d_other = torch.from_numpy(other.view(np.int64)).cuda()
        assert int(po.hamming_dev(bits, d_other, n_len).item()) == oracle.hamming(want, other, n_len), (seed, n_len)
        assert np.array_equal(po.complement_dev(bits, n_len).cpu().numpy().view(np.uint64), oracle.complement(want, n_len)), (seed, n_len)
        assert np.array_equal(po.reverse_complement_dev(bits, n_len).cpu().numpy().view(np.uint64), oracle.reverse_complement(want, n_len)), (seed, n_len)
        assert int(po.validate_dev(view).item()) == oracle.validate(n), (seed, n_len)


def _dirty(rng, alpha, n_len):
    """valid letters with nothing / a few / many / only arbitrary bytes mixed in"""
    n = alpha[rng.integers(0, alpha.size, n_len)]
    kind = int(rng.integers(0, 4))
    if kind == 1 and n_len:
        where = rng.integers(0, n_len, int(rng.integers(1, 6)))
        n[where] = rng.integers(0, 256, where.size, dtype=np.uint8)
    elif kind == 2:
        mask = rng.integers(0, 3, n_len) == 0
        n[mask] = rng.integers(0, 256, int(mask.sum()), dtype=np.uint8)
    elif kind == 3:
        n = rng.integers(0, 256, n_len, dtype=np.uint8)
    return n


@pytest.mark.parametrize("seed", range(3 * SEEDS))
def test_validated_encoders_fuzz(oracle, small_nt, seed):
    """The one-pass validated encoders (round 6) on random lengths, pointer phases, flag modes and dirt densities, one counter
    or CNT_SPREAD_COUNT: the words are the unvalidated call's bit for bit (and BYTE_LUT's in strict mode), the count is
    oracle.validate's, and an accumulator that already holds a number keeps it."""
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import n_to_bits2 as n2

    rng = np.random.default_rng(7000 + seed)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    alpha5 = np.frombuffer(b"ACGTUNacgtun", dtype=np.uint8)
    for n_len in _lengths(rng, 24):
        off_in, off_out = int(rng.choice([0, 0, 16, 48, 8, 1, 5, 127])), int(rng.choice([0, 0, 1, 2, 15]))
        strict = bool(rng.integers(0, 2))
        tail = not strict and bool(rng.integers(0, 2))
        spread = bool(rng.integers(0, 2))
        n = _dirty(rng, alpha, n_len)
        if tail:
            n = n & 0x7F
        buf = torch.zeros(n_len + 192, dtype=torch.uint8, device="cuda")
        view = buf[off_in : off_in + n_len]
        view.copy_(torch.from_numpy(n))
        words = (n_len + 31) // 32
        obuf = torch.full((words + 20,), -1, dtype=torch.int64, device="cuda")
        plain = cn.n_to_bits_dev(view, strict_lut=strict, tail_lut=tail).cpu().numpy()
        seeded = int(rng.integers(0, 1000))
        acc = None
        if not spread:
            acc = torch.full((1,), seeded, dtype=torch.int64, device="cuda")
        got, acc = cn.n_to_bits_checked_dev(view, out=obuf[off_out:], acc=acc, strict_lut=strict, tail_lut=tail, spread=spread)
        tag = (seed, n_len, off_in, off_out, strict, tail, spread)
        assert np.array_equal(got.cpu().numpy(), plain), tag
        if strict:
            assert np.array_equal(plain.view(np.uint64), oracle.n_to_bits_lut(n)), tag
        assert int(acc.sum().item()) - (0 if spread else seeded) == oracle.validate(n), tag
        o = obuf.cpu().numpy()
        assert (o[:off_out] == -1).all() and (o[off_out + words :] == -1).all(), tag
        if n_len == 0:
            continue
        # fused: packed words + decoded letters + the count, all three pointers at their own phases
        bo = int(rng.choice([0, 0, 128, 1, 16, 77]))
        pbuf = torch.full((words + off_out + 8,), -1, dtype=torch.int64, device="cuda")
        bbuf = torch.full((n_len + bo + 64,), 0x2A, dtype=torch.uint8, device="cuda")
        bits, back, acc2 = cn.round_trip_checked_dev(view, out_bits=pbuf[off_out : off_out + words], out_n=bbuf[bo : bo + n_len],
                                                     strict_lut=strict, tail_lut=tail, spread=spread)
        assert np.array_equal(bits.cpu().numpy(), plain), tag
        assert np.array_equal(back.cpu().numpy(), oracle.bits_to_n_lut(plain.view(np.uint64), n_len)), tag
        assert int(acc2.sum().item()) == oracle.validate(n), tag
        pb, bb = pbuf.cpu().numpy(), bbuf.cpu().numpy()
        assert (pb[:off_out] == -1).all() and (pb[off_out + words :] == -1).all() and (bb[:bo] == 0x2A).all() and (bb[bo + n_len :] == 0x2A).all(), tag
        # 5-letter alphabet
        n5 = _dirty(rng, alpha5, n_len)
        if tail:
            n5 = n5 & 0x7F
        view.copy_(torch.from_numpy(n5))
        plain5 = n2.n_to_bits2_dev(view, strict_lut=strict, tail_lut=tail).cpu().numpy()
        got5, acc5 = n2.n_to_bits2_checked_dev(view, strict_lut=strict, tail_lut=tail, spread=spread)
        assert np.array_equal(got5.cpu().numpy(), plain5), tag
        assert int(acc5.sum().item()) == oracle.validate(n5, allow_n=True), tag
    # host slices: small path, pipeline, several chunks
    for n_len in [int(rng.choice([1, 27, 4096, 40000, 1 << 20, (1 << 20) + 1, 3 << 20])) + int(rng.integers(0, 40)) for _ in range(4)]:
        strict = bool(rng.integers(0, 2))
        n = _dirty(rng, alpha, n_len)
        w, bad = cn.n_to_bits_hip_checked(n, strict_lut=strict)
        assert np.array_equal(w, cn.n_to_bits_hip(n, strict_lut=strict)) and bad == oracle.validate(n), (seed, n_len, strict)
        n5 = _dirty(rng, alpha5, n_len)
        w5, bad5 = n2.n_to_bits2_hip_checked(n5, strict_lut=strict)
        assert np.array_equal(w5, n2.n_to_bits2_hip(n5, strict_lut=strict)) and bad5 == oracle.validate(n5, allow_n=True), (seed, n_len, strict)
