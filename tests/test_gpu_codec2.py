"""GPU parity tests for the 2-bit codec: the HIP path (through the C ABI) against the CPU
oracle on the same seeded inputs, the reference's known-answer vectors, every kernel variant,
every edge case the reference tests or leaves undefined, and -- at BASELINE.json's full
sizes -- size-independent properties (round trip, checksum of checksums).  Bit-exact: this
is integer/byte work."""

import ctypes

import numpy as np
import pytest

from conftest import need_free_hbm

pytestmark = pytest.mark.gpu

VALID = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
SIZES = [1, 3, 4, 31, 32, 33, 63, 64, 65, 255, 256, 4095, 4096, 4097, 16383, 16384, 16385,
         32768 + 5, 65536, 100003, (1 << 20), (1 << 20) + 13, (1 << 22) + 16384 + 31]

N_ENC_VARIANTS, N_DEC_VARIANTS = 28, 46  # kEncodeVariants / kDecodeVariants in hip/codec2_launch.hpp (checked below)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "the gpu tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def cn():
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib

    _lib.lib()  # fail loudly here if the HIP library is missing
    return cn


@pytest.fixture()
def tuning(lab_build):
    from cute_nucleotides_amd import devutil

    saved = {k: devutil.get_tuning(k) for k in ("encode", "decode", "small_nt", "launch_tiles", "decode_cache_log2")}
    yield devutil
    for k, v in saved.items():
        devutil.set_tuning(k, v)


def _timed_ms(torch, fn):
    """device milliseconds of fn() between two events on torch's current stream (the stream the C ABI gets)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    e1.synchronize()
    return out, e0.elapsed_time(e1)


def _words(hexes):
    return np.array([int(h, 16) for h in hexes], dtype=np.uint64)


def _rand_valid(n_len, seed):
    return VALID[np.random.default_rng(seed).integers(0, VALID.size, n_len)]


# ---- the reference's own vectors -------------------------------------------------------
def test_reference_kats_host_and_device_tier(cn, torch_cuda, kats):
    torch = torch_cuda
    for k in kats["kats"]:
        five = "bits2" in k["fn"] or "_n2_" in k["fn"]
        if k["kind"] == "encode":
            n = k["input_ascii"].encode()
            want = _words(k["expected_words_hex"])
            host = (cn.n_to_bits2_hip if five else cn.n_to_bits_hip)(n)
            assert host.tolist() == want.tolist(), k
            d = torch.frombuffer(bytearray(n), dtype=torch.uint8).cuda()
            dev = (cn.n_to_bits2_dev if five else cn.n_to_bits_dev)(d).cpu().numpy().view(np.uint64)
            assert dev.tolist() == want.tolist(), k
        else:
            bits = _words(k["input_words_hex"])
            want = k["expected_ascii"].encode()
            host = (cn.bits_to_n2_hip if five else cn.bits_to_n_hip)(bits, k["len"])
            assert bytes(host) == want, k
            d = torch.from_numpy(bits.view(np.int64)).cuda()
            dev = (cn.bits_to_n2_dev if five else cn.bits_to_n_dev)(d, k["len"]).cpu().numpy()
            assert bytes(dev) == want, k


def test_bench_generator_input(cn, kats):
    g = kats["bench_inputs"][0]
    n = (g["unit"] * g["repeat"]).encode()
    bits = cn.n_to_bits_hip(n)
    assert bits.size == 1250 and (bits == np.uint64(0xD8D8D8D8D8D8D8D8)).all()
    assert bytes(cn.bits_to_n_hip(bits, 40000)) == n


# ---- encode vs oracle ------------------------------------------------------------------
@pytest.mark.parametrize("n_len", SIZES)
def test_encode_matches_lut_oracle_host_tier(cn, oracle, n_len):
    n = _rand_valid(n_len, n_len)
    assert np.array_equal(cn.n_to_bits_hip(n), oracle.n_to_bits_lut(n))
    assert np.array_equal(cn.n_to_bits_hip(n, strict_lut=True), oracle.n_to_bits_lut(n))


def test_variant_tables(tuning):
    assert tuning.get_tuning("encode_variants") == N_ENC_VARIANTS and tuning.get_tuning("decode_variants") == N_DEC_VARIANTS
    assert tuning.get_tuning("encode") == 0 and tuning.get_tuning("decode") == 0  # 0 = shipped default
    assert len({name for _, name in tuning.variants("encode")}) == N_ENC_VARIANTS


@pytest.mark.parametrize("variant", range(N_ENC_VARIANTS))
def test_encode_every_variant_device_tier(cn, oracle, torch_cuda, tuning, variant):
    torch = torch_cuda
    tuning.set_tuning("encode", variant)
    tuning.set_tuning("small_nt", 0)
    for n_len in (2048, 16384, 65536 + 31, (1 << 21) + 4097, 3 * (1 << 20), 2048 * 16 * 5 + 2048 * 3 + 17):
        n = _rand_valid(n_len, 7 * n_len + variant)
        want = oracle.n_to_bits_lut(n)
        d = torch.from_numpy(n).cuda()
        for strict in (False, True):
            got = cn.n_to_bits_dev(d, strict_lut=strict).cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), (variant, n_len, strict)


def test_encode_arbitrary_bytes_both_semantics(cn, oracle, torch_cuda):
    """default == the reference SIMD variants' (byte>>1)&3; strict == n_to_bits_lut, on all
    256 byte values (bytes >= 0x80 are defined as 0 where the reference is UB)."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    for n_len in (256, 32 * 1000, 1 << 20):
        n = rng.integers(0, 256, n_len, dtype=np.uint8)
        n[:256] = np.arange(256, dtype=np.uint8)
        d = torch.from_numpy(n).cuda()
        fast = cn.n_to_bits_dev(d).cpu().numpy().view(np.uint64)
        strict = cn.n_to_bits_dev(d, strict_lut=True).cpu().numpy().view(np.uint64)
        assert np.array_equal(fast, oracle.n_to_bits_bitextract(n))  # n_len % 32 == 0: no LUT tail
        assert np.array_equal(fast, oracle.n_to_bits_movemask(n))
        assert np.array_equal(strict, oracle.n_to_bits_lut(n))
    # ragged tail with arbitrary bytes: strict still equals the LUT oracle; the default applies
    # (byte>>1)&3 to the tail too (documented; the reference's SIMD fns switch to the LUT there)
    n = rng.integers(0, 256, 16384 + 77, dtype=np.uint8)
    d = torch.from_numpy(n).cuda()
    assert np.array_equal(cn.n_to_bits_dev(d, strict_lut=True).cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(n))
    fast = cn.n_to_bits_dev(d).cpu().numpy().view(np.uint64)
    codes = ((n >> 1) & 3).astype(np.uint64)
    pad = np.zeros((-n.size) % 32, dtype=np.uint64)
    want = np.bitwise_or.reduce(np.concatenate([codes, pad]).reshape(-1, 32) << (np.arange(32, dtype=np.uint64) * np.uint64(2)), axis=1)
    assert np.array_equal(fast, want)


def test_encode_unaligned_device_pointers(cn, oracle, torch_cuda):
    torch = torch_cuda
    n = _rand_valid(70001, 3)
    buf = torch.zeros(n.size + 64, dtype=torch.uint8, device="cuda")
    for off in (1, 3, 8, 15):
        view = buf[off : off + n.size]
        view.copy_(torch.from_numpy(n))
        got = cn.n_to_bits_dev(view).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, oracle.n_to_bits_lut(n)), off


def test_encode_last_word_zero_padded_and_no_overrun(cn, oracle, torch_cuda):
    torch = torch_cuda
    for n_len in (5, 16384 + 5, (1 << 20) + 1):
        n = _rand_valid(n_len, n_len)
        words = (n_len + 31) // 32
        out = torch.full((words + 4,), -1, dtype=torch.int64, device="cuda")
        cn.n_to_bits_dev(torch.from_numpy(n).cuda(), out=out)
        got = out.cpu().numpy()
        assert (got[words:] == -1).all(), "wrote past ceil(n/32) words"
        assert np.array_equal(got[:words].view(np.uint64), oracle.n_to_bits_lut(n))
        assert int(got[:words].view(np.uint64)[-1]) >> (2 * (n_len & 31)) == 0


# ---- decode vs oracle ------------------------------------------------------------------
@pytest.mark.parametrize("n_len", SIZES)
def test_decode_matches_lut_oracle_host_tier(cn, oracle, n_len):
    rng = np.random.default_rng(n_len + 11)
    bits = rng.integers(0, 2**64, (n_len + 31) // 32, dtype=np.uint64)  # garbage beyond len is ignored
    assert np.array_equal(cn.bits_to_n_hip(bits, n_len), oracle.bits_to_n_lut(bits, n_len))


@pytest.mark.parametrize("variant", range(N_DEC_VARIANTS))
def test_decode_every_variant_device_tier(cn, oracle, torch_cuda, tuning, variant):
    torch = torch_cuda
    tuning.set_tuning("decode", variant)
    tuning.set_tuning("small_nt", 0)
    rng = np.random.default_rng(variant)
    for n_len in (2048, 16384, 65536 + 31, (1 << 21) + 4097, 3 * (1 << 20), 2048 * 16 * 5 + 2048 * 3 + 17):
        bits = rng.integers(0, 2**64, (n_len + 31) // 32, dtype=np.uint64)
        d = torch.from_numpy(bits.view(np.int64)).cuda()
        got = cn.bits_to_n_dev(d, n_len).cpu().numpy()
        assert np.array_equal(got, oracle.bits_to_n_lut(bits, n_len)), (variant, n_len)


def test_decode_len_smaller_than_capacity_and_no_overrun(cn, oracle, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(99)
    bits = rng.integers(0, 2**64, 4096, dtype=np.uint64)
    d = torch.from_numpy(bits.view(np.int64)).cuda()
    for length in (0, 1, 31, 33, 16384, 16385, 4096 * 32 - 1, 4096 * 32):
        out = torch.full((4096 * 32 + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        cn.bits_to_n_dev(d, length, out=out)
        got = out.cpu().numpy()
        assert (got[length:] == 0x5A).all(), "decode wrote past len"
        assert np.array_equal(got[:length], oracle.bits_to_n_lut(bits, length))
    with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
        cn.bits_to_n_dev(d, 4096 * 32 + 1)


def test_decode_unaligned_output(cn, oracle, torch_cuda):
    torch = torch_cuda
    bits = np.random.default_rng(4).integers(0, 2**64, 3000, dtype=np.uint64)
    d = torch.from_numpy(bits.view(np.int64)).cuda()
    buf = torch.zeros(3000 * 32 + 64, dtype=torch.uint8, device="cuda")
    for off in (1, 7, 16):
        cn.bits_to_n_dev(d, 3000 * 32 - 3, out=buf[off:])
        assert np.array_equal(buf[off : off + 3000 * 32 - 3].cpu().numpy(), oracle.bits_to_n_lut(bits, 3000 * 32 - 3))


def test_round_trip_case_and_u_fold(cn):
    n = np.frombuffer(b"acgtuACGTU" * 1001, dtype=np.uint8)
    back = cn.bits_to_n_hip(cn.n_to_bits_hip(n), n.size)
    assert bytes(back) == bytes(n).upper().replace(b"U", b"T")


def test_host_tier_multi_chunk(cn, oracle):
    """> one 64 Mi-nt staging chunk, ragged: exercises the double-buffered H2D/kernel/D2H loop."""
    n_len = (64 << 20) * 2 + 12345
    n = oracle.fill_random_acgt(n_len, 42)
    bits = cn.n_to_bits_hip(n)
    assert np.array_equal(bits, oracle.n_to_bits_movemask(n))  # port == LUT oracle on valid input
    assert np.array_equal(bits[: 1 << 16], oracle.n_to_bits_lut(n[: 32 << 16]))
    back = cn.bits_to_n_hip(bits, n_len)
    assert np.array_equal(back, n)


def test_sharded_tier_single_device(cn, oracle):
    n = oracle.fill_random_acgt((1 << 22) + 77, 9)
    bits = cn.n_to_bits_hip_sharded(n, ndev=1)
    assert np.array_equal(bits, oracle.n_to_bits_lut(n))
    assert np.array_equal(cn.bits_to_n_hip_sharded(bits, n.size, ndev=1), n)
    bits_all = cn.n_to_bits_hip_sharded(n, ndev=0)  # all visible devices
    assert np.array_equal(bits_all, bits)
    from cute_nucleotides_amd._lib import CuteNtError

    import torch

    with pytest.raises(CuteNtError):
        cn.n_to_bits_hip_sharded(n, ndev=torch.cuda.device_count() + 1)
    # the per-device workers are persistent: repeated calls, a shutdown in between (which releases
    # the workers' streams and staging), concurrent callers (they queue on the pool)
    from cute_nucleotides_amd import _lib
    import threading

    for _ in range(3):
        assert np.array_equal(cn.n_to_bits_hip_sharded(n, ndev=1), bits)
    assert _lib.lib().cnt_shutdown() == 0
    assert np.array_equal(cn.n_to_bits_hip_sharded(n, ndev=1), bits)
    bad = []

    def worker(seed):
        m = oracle.fill_random_acgt(300000 + seed, seed)
        for _ in range(4):
            if not np.array_equal(cn.n_to_bits_hip_sharded(m, ndev=1), oracle.n_to_bits_lut(m)):
                bad.append(seed)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


# ---- device utilities used by the large-size checks --------------------------------------
def test_device_generator_and_checksum_match_oracle(cn, oracle, torch_cuda):
    from cute_nucleotides_amd import devutil

    torch = torch_cuda
    for n_len, first in ((1 << 20, 0), (100003, 32 * 12345), (31, 64)):
        d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
        devutil.fill_random_acgt(d, 0x5EED, first_nt=first)
        assert np.array_equal(d.cpu().numpy(), oracle.fill_random_acgt(n_len, 0x5EED, first_nt=first))
    d5 = torch.empty(27 * 5000 + 11, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgtn(d5, 77, first_nt=27 * 3)
    assert np.array_equal(d5.cpu().numpy(), oracle.fill_random_acgtn(d5.numel(), 77, first_nt=27 * 3))
    w = np.random.default_rng(1).integers(0, 2**64, 100001, dtype=np.uint64)
    dw = torch.from_numpy(w.view(np.int64)).cuda()
    assert devutil.checksum_words(dw, first_word=17) == oracle.checksum_words(w, first_word=17)
    a = torch.from_numpy(np.arange(100000, dtype=np.uint8)).cuda()
    b = a.clone()
    assert devutil.count_mismatch(a, b) == 0
    b[5] += 1
    b[99999] += 1
    b[31337] ^= 0x80
    assert devutil.count_mismatch(a, b) == 3
    assert devutil.count_mismatch(a[1:], b[1:]) == 3  # unaligned path


# ---- BASELINE.json configs[1], configs[2]: 1 GiB, bit-exact vs n_to_bits_lut + round trip ------
def test_config_1gib_encode_bit_exact_and_round_trip(cn, oracle, torch_cuda, fullsize):
    from cute_nucleotides_amd import devutil

    torch = torch_cuda
    n_len = 1 << 30
    need_free_hbm(4)
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 0x5EED)
    packed = torch.empty(n_len // 32, dtype=torch.int64, device="cuda")
    _, enc_ms = _timed_ms(torch, lambda: cn.n_to_bits_dev(d, out=packed))
    back = torch.empty(n_len, dtype=torch.uint8, device="cuda")  # allocated outside the timed call
    _, dec_ms = _timed_ms(torch, lambda: cn.bits_to_n_dev(packed, n_len, out=back))
    fullsize(30, enc_ms + dec_ms, encode_ms=round(enc_ms, 3), decode_ms=round(dec_ms, 3), config="configs[1]+[2]", first_call=True)
    assert devutil.count_mismatch(d, back) == 0  # configs[2]: encode -> decode round trip
    host_n = oracle.fill_random_acgt(n_len, 0x5EED)  # regenerate on the host: no PCIe copy of the input
    want = oracle.n_to_bits_lut(host_n)  # the scalar parity oracle, all 2^30 nt
    got = packed.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)  # configs[1]: bit-exact vs n_to_bits_lut
    assert devutil.checksum_words(packed) == oracle.checksum_words(want)


# ---- metric size: 16 GiB, verified through size-independent properties ----------------------
def test_metric_16gib_round_trip_and_checksum_of_checksums(cn, oracle, torch_cuda, fullsize):
    from cute_nucleotides_amd import devutil

    torch = torch_cuda
    n_len = 1 << 34
    need_free_hbm(40)  # 16 + 4 + 16 GiB resident
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 0xC0FFEE)
    packed = torch.empty(n_len // 32, dtype=torch.int64, device="cuda")
    _, enc_ms = _timed_ms(torch, lambda: cn.n_to_bits_dev(d, out=packed))
    back = torch.empty(n_len, dtype=torch.uint8, device="cuda")  # allocated outside the timed call
    _, dec_ms = _timed_ms(torch, lambda: cn.bits_to_n_dev(packed, n_len, out=back))
    fullsize(34, enc_ms + dec_ms, encode_ms=round(enc_ms, 3), decode_ms=round(dec_ms, 3), config="metric size", first_call=True)
    assert devutil.count_mismatch(d, back) == 0
    del back
    # checksum of checksums: per-64 MiB-chunk checksums of the packed words; sampled chunks are
    # reproduced by the CPU oracle from the seed alone, and the chunk sums add up to the whole.
    chunk_nt = 64 << 20
    chunk_w = chunk_nt // 32
    n_chunks = n_len // chunk_nt
    total = devutil.checksum_words(packed)
    sums = [devutil.checksum_words(packed[c * chunk_w : (c + 1) * chunk_w], first_word=c * chunk_w) for c in range(n_chunks)]
    assert sum(sums) % (1 << 64) == total
    for c in (0, 1, n_chunks // 2 + 3, n_chunks - 1):
        host_n = oracle.fill_random_acgt(chunk_nt, 0xC0FFEE, first_nt=c * chunk_nt)
        want = oracle.n_to_bits_lut(host_n)
        assert oracle.checksum_words(want, first_word=c * chunk_w) == sums[c], c
        got = packed[c * chunk_w : (c + 1) * chunk_w].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), c


# ---- BASELINE.json configs[3]: encode + decode over a 64 GiB buffer (144 GiB resident) ----------
def test_config_64gib_round_trip_multi_launch(cn, oracle, torch_cuda, fullsize):
    """2^36 nt needs 2^25 workgroups x 64 threads = 2^31 threads, one more than HIP allows in a
    launch, so this also covers the launcher's split into several launches."""
    from cute_nucleotides_amd import devutil

    torch = torch_cuda
    n_len = 1 << 36
    need_free_hbm(150)  # 64 + 16 + 64 GiB resident
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 0xBEEF)
    packed = torch.empty(n_len // 32, dtype=torch.int64, device="cuda")
    _, enc_ms = _timed_ms(torch, lambda: cn.n_to_bits_dev(d, out=packed))
    back = torch.empty(n_len, dtype=torch.uint8, device="cuda")  # allocated outside the timed call
    _, dec_ms = _timed_ms(torch, lambda: cn.bits_to_n_dev(packed, n_len, out=back))
    fullsize(36, enc_ms + dec_ms, encode_ms=round(enc_ms, 3), decode_ms=round(dec_ms, 3), config="configs[3] two passes", first_call=True)
    assert devutil.count_mismatch(d, back) == 0
    del back
    chunk_nt = 16 << 20
    chunk_w = chunk_nt // 32
    n_chunks = n_len // chunk_nt
    # chunks around the launch split point ((2^31-1) / 64 threads -> 33554368 tiles of 2 KiB, i.e. the
    # second launch covers only the last 64 tiles of the 2^36-nt buffer) and the ends
    split_nt = ((0x7FFFFFFF // 64) // 64) * 64 * 2048
    assert 0 < n_len - split_nt < chunk_nt
    picks = {0, 1, n_chunks // 2, n_chunks - 2, n_chunks - 1}
    for c in sorted(picks):
        host_n = oracle.fill_random_acgt(chunk_nt, 0xBEEF, first_nt=c * chunk_nt)
        want = oracle.n_to_bits_lut(host_n)
        got = packed[c * chunk_w : (c + 1) * chunk_w].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), c


# ---- any-alignment plan: head peel + funnel-shifted loads (hip/device_tier.inc encode_dev/decode_dev) ----
ALIGN_IN_OFFS = [0, 1, 2, 3, 4, 5, 7, 8, 12, 13, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 127]
ALIGN_OUT_WORD_OFFS = [0, 1, 2, 3, 7, 8, 15]


@pytest.mark.parametrize("strict", [False, True])
def test_encode_alignment_matrix(cn, oracle, torch_cuda, tuning, strict):
    """Every input byte phase x output word phase, sizes on both sides of the peel/tile/guard
    boundaries, guard values around the output: bit-exact and nothing written outside."""
    torch = torch_cuda
    tuning.set_tuning("small_nt", 0)  # small ragged inputs would otherwise take the generic kernel alone
    if True:
        sizes = [2048 * 3 + 144 + 5, 512 + 2048 + 143, 512 + 2048 + 144, 512 + 2048 + 145, 40000, 100003]
        big = _rand_valid(max(sizes), 77) if not strict else np.random.default_rng(78).integers(0, 256, max(sizes), dtype=np.uint8)
        ibuf = torch.zeros(big.size + 256, dtype=torch.uint8, device="cuda")
        obuf = torch.empty(big.size // 32 + 64, dtype=torch.int64, device="cuda")
        for n_len in sizes:
            n = big[:n_len]
            want = oracle.n_to_bits_lut(n)  # valid alphabet (fast mode) or any bytes under CNT_STRICT_LUT
            words = (n_len + 31) // 32
            for io in ALIGN_IN_OFFS:
                view = ibuf[io : io + n_len]
                view.copy_(torch.from_numpy(n))
                for oo in ALIGN_OUT_WORD_OFFS:
                    obuf.fill_(-1)
                    out = obuf[8 + oo : 8 + oo + words]
                    cn.n_to_bits_dev(view, out=out, strict_lut=strict)
                    got = obuf.cpu().numpy()
                    assert (got[: 8 + oo] == -1).all() and (got[8 + oo + words :] == -1).all(), (n_len, io, oo)
                    assert np.array_equal(got[8 + oo : 8 + oo + words].view(np.uint64), want), (n_len, io, oo)


def test_decode_alignment_matrix(cn, oracle, torch_cuda, tuning):
    """Every output byte phase (mod 128) x packed-word phase, lengths around the boundaries."""
    torch = torch_cuda
    tuning.set_tuning("small_nt", 0)
    bits = np.random.default_rng(5).integers(0, 2**64, 4000, dtype=np.uint64)
    dbuf = torch.zeros(bits.size + 16, dtype=torch.int64, device="cuda")
    obuf = torch.empty(bits.size * 32 + 512, dtype=torch.uint8, device="cuda")
    lens = [4096 + 127, 4096 + 128, 4096 + 129, 3 * 4096 + 77, 100003, bits.size * 32]
    want_full = oracle.bits_to_n_lut(bits, bits.size * 32)
    for wo in (0, 1, 3):
        d = dbuf[wo : wo + bits.size]
        d.copy_(torch.from_numpy(bits.view(np.int64)))
        for oo in list(range(0, 36)) + [47, 48, 63, 64, 65, 96, 111, 127]:
            for length in lens:
                obuf.fill_(0x2A)
                cn.bits_to_n_dev(d, length, out=obuf[128 + oo : 128 + oo + length])
                got = obuf.cpu().numpy()
                assert (got[: 128 + oo] == 0x2A).all() and (got[128 + oo + length :] == 0x2A).all(), (wo, oo, length)
                assert np.array_equal(got[128 + oo : 128 + oo + length], want_full[:length]), (wo, oo, length)



def test_alignment_matrix_on_the_product_build(cn, oracle, torch_cuda):
    """The two matrices above force the tile kernels onto small inputs through a lab-build knob.  The product library has
    no knob: here the same head peel / window / funnel-shift paths run as shipped, at sizes past its small-input path
    (2^17 nt), a reduced matrix of input byte phase x packed-word phase x output byte phase, guards around every output."""
    from cute_nucleotides_amd import _lib

    assert not _lib.is_lab_build()
    torch = torch_cuda
    sizes = [(1 << 17) + 2048 * 3 + 149, (1 << 17) + 4096 + 129, (1 << 18) + 100003]
    big = _rand_valid(max(sizes), 79)
    ibuf = torch.zeros(big.size + 256, dtype=torch.uint8, device="cuda")
    pbuf = torch.empty(big.size // 32 + 64, dtype=torch.int64, device="cuda")
    obuf = torch.empty(big.size + 512, dtype=torch.uint8, device="cuda")
    for n_len in sizes:
        n = big[:n_len]
        want = oracle.n_to_bits_lut(n)
        want_back = oracle.bits_to_n_lut(want, n_len)
        words = want.size
        for io in (0, 1, 5, 16, 17, 63, 64, 100, 127):
            view = ibuf[io : io + n_len]
            view.copy_(torch.from_numpy(n))
            for po in (0, 1, 7, 8, 15):
                pbuf.fill_(-1)
                out = pbuf[8 + po : 8 + po + words]
                cn.n_to_bits_dev(view, out=out)
                got = pbuf.cpu().numpy()
                assert (got[: 8 + po] == -1).all() and (got[8 + po + words :] == -1).all(), (n_len, io, po)
                assert np.array_equal(got[8 + po : 8 + po + words].view(np.uint64), want), (n_len, io, po)
                for oo in (0, 1, 15, 16, 33, 64, 127) if io in (0, 5) else (3,):
                    obuf.fill_(0x2A)
                    cn.bits_to_n_dev(out, n_len, out=obuf[128 + oo : 128 + oo + n_len])
                    b = obuf.cpu().numpy()
                    assert (b[: 128 + oo] == 0x2A).all() and (b[128 + oo + n_len :] == 0x2A).all(), (n_len, po, oo)
                    assert np.array_equal(b[128 + oo : 128 + oo + n_len], want_back), (n_len, po, oo)


def test_decode_large_buffer_4k_head(cn, oracle, torch_cuda):
    """>= 2^20 nt: the head is peeled up to a 4-KiB boundary of the output (range kernel), then the
    funnel-shifting tiles, then a ragged end that does not start on a word."""
    torch = torch_cuda
    words = 40000
    bits = np.random.default_rng(6).integers(0, 2**64, words, dtype=np.uint64)
    d = torch.from_numpy(bits.view(np.int64)).cuda()
    want = oracle.bits_to_n_lut(bits, words * 32)
    obuf = torch.empty(words * 32 + 3 * 4096, dtype=torch.uint8, device="cuda")
    base = (-obuf.data_ptr()) % 4096  # obuf[base] is 4-KiB aligned
    for oo in (0, 1, 16, 100, 2048, 2049, 4095):
        for length in (words * 32, words * 32 - 4097, (1 << 20) + 5):
            obuf.fill_(0x2A)
            cn.bits_to_n_dev(d, length, out=obuf[base + oo : base + oo + length])
            got = obuf.cpu().numpy()
            assert (got[: base + oo] == 0x2A).all() and (got[base + oo + length :] == 0x2A).all(), (oo, length)
            assert np.array_equal(got[base + oo : base + oo + length], want[:length]), (oo, length)


@pytest.mark.parametrize("small_nt", [0, 1 << 17])
def test_device_tier_is_graph_capturable(cn, oracle, torch_cuda, tuning, small_nt):
    """The device tier only enqueues (no sync, no allocation): an encode + decode pair recorded
    into a HIP graph replays correctly on new contents of the same buffers."""
    torch = torch_cuda
    tuning.set_tuning("small_nt", small_nt)
    n_len = 40000 + 13  # the reference's bench size plus a ragged tail: tile, head and tail kernels all in the graph
    d_in = torch.zeros(n_len + 3, dtype=torch.uint8, device="cuda")[3:]  # and a misaligned input
    d_pk = torch.zeros((n_len + 31) // 32, dtype=torch.int64, device="cuda")
    d_out = torch.zeros(n_len, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # warm-up outside capture (module load)
        cn.n_to_bits_dev(d_in, out=d_pk)
        cn.bits_to_n_dev(d_pk, n_len, out=d_out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cn.n_to_bits_dev(d_in, out=d_pk)
        cn.bits_to_n_dev(d_pk, n_len, out=d_out)
    for seed in (1, 2, 3):
        n = _rand_valid(n_len, seed)
        d_in.copy_(torch.from_numpy(n))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(d_pk.cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(n))
        assert bytes(d_out.cpu().numpy()) == bytes(n).upper().replace(b"U", b"T")


# ---- fused round trip (BASELINE.json configs[3]) ----------------------------------------------
@pytest.mark.parametrize("strict", [False, True])
def test_fused_round_trip_matches_two_calls(cn, oracle, torch_cuda, strict):
    """cnt_round_trip_dev == n_to_bits followed by bits_to_n, for aligned (round_trip_stream) and misaligned
    (round_trip_window / the edge body alone) pointers, ragged sizes, any bytes under CNT_STRICT_LUT; guard bytes around both outputs."""
    torch = torch_cuda
    rng = np.random.default_rng(41)
    for n_len in (1, 31, 2047, 2048, 2049, 40000, 2048 * 37 + 5, (1 << 20) + 13):
        n = rng.integers(0, 256, n_len, dtype=np.uint8) if strict else _rand_valid(n_len, n_len)
        want_bits = oracle.n_to_bits_lut(n)
        want_back = oracle.bits_to_n_lut(want_bits, n_len)
        words = want_bits.size
        for in_off, bits_off, back_off in ((0, 0, 0), (128, 16, 256), (1, 0, 0), (0, 1, 0), (0, 0, 3)):
            ibuf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
            view = ibuf[in_off : in_off + n_len]
            view.copy_(torch.from_numpy(n))
            pbuf = torch.full((words + 64,), -1, dtype=torch.int64, device="cuda")
            bbuf = torch.full((n_len + 512,), 0x2A, dtype=torch.uint8, device="cuda")
            bits, back = cn.round_trip_dev(view, out_bits=pbuf[bits_off : bits_off + words], out_n=bbuf[back_off : back_off + n_len],
                                           strict_lut=strict)
            p, b = pbuf.cpu().numpy(), bbuf.cpu().numpy()
            assert np.array_equal(p[bits_off : bits_off + words].view(np.uint64), want_bits), (n_len, in_off, bits_off, back_off)
            assert np.array_equal(b[back_off : back_off + n_len], want_back), (n_len, in_off, bits_off, back_off)
            assert (p[:bits_off] == -1).all() and (p[bits_off + words :] == -1).all()
            assert (b[:back_off] == 0x2A).all() and (b[back_off + n_len :] == 0x2A).all()


@pytest.mark.parametrize("strict", [False, True])
def test_fused_round_trip_alignment_matrix(cn, oracle, torch_cuda, strict):
    """cnt_round_trip_dev at EVERY combination of input byte phase, packed-word phase and decoded-output
    byte phase (the window kernel: tiles on the decoded stream's lines, both other phases resolved on the packed codes),
    sizes on both sides of the tile + slack boundaries and around 2^20 (4-KiB grain), guard values around both outputs:
    bit-exact against n_to_bits_lut / bits_to_n_lut and nothing written outside.  tail_lut on arbitrary bytes too."""
    torch = torch_cuda
    sizes = [4096 + 399, 4096 + 400, 4096 + 401, 4096 + 128 + 400 + 1, 3 * 4096 + 77, 40000, 100003, (1 << 20) + 4096 + 13]
    rng = np.random.default_rng(61)
    big = rng.integers(0, 256, max(sizes), dtype=np.uint8) if strict else _rand_valid(max(sizes), 62)
    ibuf = torch.zeros(big.size + 256, dtype=torch.uint8, device="cuda")
    pbuf = torch.empty(big.size // 32 + 64, dtype=torch.int64, device="cuda")
    bbuf = torch.empty(big.size + 8192 + 512, dtype=torch.uint8, device="cuda")
    in_offs = [0, 1, 5, 15, 16, 17, 63, 64, 100, 127]
    back_offs = [0, 1, 3, 15, 16, 33, 64, 127, 128, 2049, 4095]
    for n_len in sizes:
        n = big[:n_len]
        want_bits = oracle.n_to_bits_lut(n)
        want_back = oracle.bits_to_n_lut(want_bits, n_len)
        words = want_bits.size
        for io in in_offs if n_len < (1 << 20) else [0, 5, 64]:
            view = ibuf[io : io + n_len]
            view.copy_(torch.from_numpy(n))
            for po in (0, 1, 3, 8, 15):
                for bo in back_offs if n_len < 100003 else [0, 1, 33, 2049, 4095]:
                    pbuf.fill_(-1)
                    bbuf.fill_(0x2A)
                    cn.round_trip_dev(view, out_bits=pbuf[8 + po : 8 + po + words], out_n=bbuf[128 + bo : 128 + bo + n_len], strict_lut=strict)
                    p, b = pbuf.cpu().numpy(), bbuf.cpu().numpy()
                    assert (p[: 8 + po] == -1).all() and (p[8 + po + words :] == -1).all(), (n_len, io, po, bo)
                    assert (b[: 128 + bo] == 0x2A).all() and (b[128 + bo + n_len :] == 0x2A).all(), (n_len, io, po, bo)
                    assert np.array_equal(p[8 + po : 8 + po + words].view(np.uint64), want_bits), (n_len, io, po, bo)
                    assert np.array_equal(b[128 + bo : 128 + bo + n_len], want_back), (n_len, io, po, bo)
    if not strict:  # CNT_TAIL_LUT through the window kernel: the SIMD encoders to the letter on arbitrary bytes -- bit extraction
        # on whole 32-nt blocks, BYTE_LUT on the final partial word, whose first letters a tile's last packed dword must not
        # reach (every residue of the length mod 32, several phases of all three pointers)
        for n_len in [100003 + k for k in range(32)] + [4096 * 3 + 400 + k for k in (1, 17, 18, 30, 31)]:
            raw = rng.integers(0, 128, n_len, dtype=np.uint8)  # the reference indexes BYTE_LUT[128]: 7-bit input
            whole = raw.size // 32 * 32
            want = np.concatenate([oracle.n_to_bits_bitextract(raw[:whole]), oracle.n_to_bits_lut(raw[whole:])]) if raw.size % 32 else oracle.n_to_bits_bitextract(raw)
            for io, po, bo in ((7, 1, 3), (0, 0, 1), (100, 5, 2049), (16, 0, 16)):
                view = ibuf[io : io + raw.size]
                view.copy_(torch.from_numpy(raw))
                bits, back = cn.round_trip_dev(view, out_bits=pbuf[po : po + (raw.size + 31) // 32], out_n=bbuf[bo : bo + raw.size], tail_lut=True)
                assert np.array_equal(bits.cpu().numpy().view(np.uint64), want), (n_len, io, po, bo)
                assert np.array_equal(back.cpu().numpy(), oracle.bits_to_n_lut(want, raw.size)), (n_len, io, po, bo)


def test_fused_tail_lut_never_lets_a_tile_touch_the_final_partial_word(cn, oracle, torch_cuda):
    """CNT_TAIL_LUT through round_trip_window: the final partial word is BYTE_LUT's (n_to_bits.rs:109-111) in BOTH outputs --
    its packed codes and its decoded letters -- and everything in front of it is bit extraction.  Whether the last tile's
    letters could reach into that word depends on all three pointer phases and the length at once (a simulation of the
    launcher's arithmetic over every phase found 370 000 such cases before the launcher was told where the word starts):
    twelve of them, input byte phase / packed-word phase / output byte phase / length, each with the 31 lengths around it;
    arbitrary 7-bit bytes."""
    torch = torch_cuda
    rng = np.random.default_rng(63)
    big = rng.integers(0, 128, 5 * 4096 + 256, dtype=np.uint8)
    ib = torch.zeros(big.size + 512, dtype=torch.uint8, device="cuda")
    pb = torch.zeros(big.size // 32 + 16, dtype=torch.int64, device="cuda")
    bb = torch.zeros(big.size + 4096, dtype=torch.uint8, device="cuda")
    assert ib.data_ptr() % 128 == 0 and pb.data_ptr() % 64 == 0 and bb.data_ptr() % 128 == 0  # the phases below are absolute
    for io, po, bo, n0 in ((100, 7, 24, 12412), (5, 3, 54, 12507), (100, 7, 54, 12380), (100, 7, 58, 12380), (37, 3, 89, 12475), (1, 1, 27, 12543),
                           (100, 7, 62, 12380), (5, 3, 62, 12507), (5, 3, 55, 12507), (1, 3, 22, 12543), (37, 3, 62, 12507), (1, 3, 88, 12479)):
        ib[io : io + big.size].copy_(torch.from_numpy(big))
        for n_len in range(n0 - 15, n0 + 16):
            raw, whole = big[:n_len], n_len // 32 * 32
            want = np.concatenate([oracle.n_to_bits_bitextract(raw[:whole]), oracle.n_to_bits_lut(raw[whole:])]) if n_len % 32 else oracle.n_to_bits_bitextract(raw)
            bits, back = cn.round_trip_dev(ib[io : io + n_len], out_bits=pb[po : po + (n_len + 31) // 32], out_n=bb[bo : bo + n_len], tail_lut=True)
            assert np.array_equal(bits.cpu().numpy().view(np.uint64), want), (io, po, bo, n_len)
            assert np.array_equal(back.cpu().numpy(), oracle.bits_to_n_lut(want, n_len)), (io, po, bo, n_len)


def test_fused_round_trip_config3_64gib(cn, oracle, torch_cuda, fullsize):
    """configs[3]: fused encode+decode over a 64 GiB buffer (144 GiB resident, 160 GiB while the plain
    encoder's words are held beside the fused ones): the decoded copy equals the input (random upper-case
    ACGT is its own canonical form), the packed words equal the plain encoder's, and sampled chunks --
    incl. the ones around the launch split at 2^36 - 64 tiles -- equal the CPU oracle's encode of the
    host-regenerated input.  The size is never reduced: too little free HBM is a visible skip."""
    torch = torch_cuda
    from cute_nucleotides_amd import devutil

    need_free_hbm(216)  # 64 (in) + 16 (packed) + 64 (decoded) + 16 (reference words) GiB + slack; 64 + 64 more while the input moves to phase 5
    n_len = (1 << 36) + 2048 * 3 + 77
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 36)
    bits = torch.empty((n_len + 31) // 32, dtype=torch.int64, device="cuda")
    back = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    _, ms = _timed_ms(torch, lambda: cn.round_trip_dev(d, out_bits=bits, out_n=back))
    fullsize(36, ms, config="configs[3] fused", nt=n_len, first_call=True, gbs=round(2.25 * n_len / ms / 1e6, 1))
    assert devutil.count_mismatch(d, back) == 0
    del back
    ref = cn.n_to_bits_dev(d)
    assert devutil.checksum_words(bits) == devutil.checksum_words(ref)
    assert devutil.count_mismatch(bits.view(torch.uint8), ref.view(torch.uint8)) == 0
    del ref
    chunk_nt = 16 << 20
    chunk_w = chunk_nt // 32
    n_chunks = (1 << 36) // chunk_nt
    for c in (0, n_chunks // 3, n_chunks - 1):
        host_n = oracle.fill_random_acgt(chunk_nt, 36, first_nt=c * chunk_nt)
        got = bits[c * chunk_w : (c + 1) * chunk_w].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, oracle.n_to_bits_lut(host_n)), c
    tail = oracle.fill_random_acgt(2048 * 3 + 77, 36, first_nt=1 << 36)  # the ragged end behind the fused tiles
    assert np.array_equal(bits[(1 << 36) // 32 :].cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(tail))
    # the same 64 GiB through the ANY-ALIGNMENT kernel (round 4: round_trip_window, two launches at this size, head and end as
    # edge items of the second): input at byte phase 5, packed words at word phase 1, decoded output at byte phase 77 of
    # buffers that hold a few bytes more; same words as the aligned call above, decoded copy equal to the input
    want_sum = devutil.checksum_words(bits)
    del bits
    torch.cuda.empty_cache()
    pad_in = torch.empty(n_len + 128, dtype=torch.uint8, device="cuda")
    v_in = pad_in[5 : 5 + n_len]
    devutil.fill_random_acgt(pad_in[:n_len], 36)
    v_in.copy_(d)
    del d
    torch.cuda.empty_cache()
    pad_bits = torch.full(((n_len + 31) // 32 + 8,), -1, dtype=torch.int64, device="cuda")
    pad_back = torch.full((n_len + 256,), 0x2A, dtype=torch.uint8, device="cuda")
    v_bits, v_back = pad_bits[1 : 1 + (n_len + 31) // 32], pad_back[77 : 77 + n_len]
    _, ms2 = _timed_ms(torch, lambda: cn.round_trip_dev(v_in, out_bits=v_bits, out_n=v_back))
    fullsize(36, ms2, config="configs[3] fused, misaligned (in +5 B, words +8 B, out +77 B)", nt=n_len, gbs=round(2.25 * n_len / ms2 / 1e6, 1))
    assert devutil.checksum_words(v_bits) == want_sum
    assert bool((pad_bits[:1] == -1).all()) and bool((pad_bits[1 + (n_len + 31) // 32 :] == -1).all())
    assert bool((pad_back[:77] == 0x2A).all()) and bool((pad_back[77 + n_len :] == 0x2A).all())
    for lo in (0, (1 << 35) - 12345, n_len - (1 << 28)):  # decoded copy == input on three 256-MiB windows (torch compares, any alignment)
        assert torch.equal(v_back[lo : lo + (1 << 28)], v_in[lo : lo + (1 << 28)]), lo


@pytest.mark.parametrize("past_cache", [False, True])
def test_decode_turn_rotation_at_every_kind_of_packed_offset(cn, oracle, torch_cuda, request, past_cache):
    """Decodes of 2^20 + a ragged bit with the packed words at every kind of offset inside their page, output phases 0 / 16 / 5 /
    77 / 4095, guards around the output, bit-exact against bits_to_n_lut.  As SHIPPED (past_cache = False, the product library)
    a call of this size lies inside the Infinity Cache: the output is peeled to its 4-KiB page and the stream kernel (any dword
    phase) or the shifted kernel (a bit phase) takes it -- NO turn placement, no window kernel: those start above
    cnt_chip_cache_nt (2^30 nt on an SPX MI355X; round 5's docstring claimed them for >= 2^20).  past_cache = True
    runs the same matrix on the lab build with the gate forced to 1 nt (tuning key decode_cache_log2 = 0): now the launcher peels
    the 0-3 (+4) further pages that place the XCD turns (device_tier.inc decode_turn_pages) -- a head of up to 5 pages rides in
    the edge items of the same launch -- and bits_to_n_window takes every packed stream off its lines or dwords."""
    from cute_nucleotides_amd import _lib, devutil

    if past_cache:
        request.getfixturevalue("lab_build")
        devutil.set_tuning("decode_cache_log2", 0)
        request.addfinalizer(lambda: devutil.set_tuning("decode_cache_log2", -1))
    else:
        assert not _lib.is_lab_build()
    torch = torch_cuda
    n_len = (1 << 20) + 4096 * 5 + 1234
    n = _rand_valid(n_len, 81)
    want = oracle.n_to_bits_lut(n)
    want_back = oracle.bits_to_n_lut(want, n_len)
    pbuf = torch.zeros(want.size + 2048, dtype=torch.int64, device="cuda")
    obuf = torch.empty(n_len + 4 * 4096, dtype=torch.uint8, device="cuda")
    pb = ((-pbuf.data_ptr()) % 4096) // 8  # pbuf[pb] sits on a 4-KiB page ...
    ob = 4096 + (-obuf.data_ptr()) % 4096  # ... and so does obuf[ob], with a guard page in front of it
    for p_off in (0, 8, 264, 512, 1016, 1024, 1032, 1544, 2048, 2056, 2824, 3072, 3080, 3592, 4088):
        d = pbuf[pb + p_off // 8 : pb + p_off // 8 + want.size]
        d.copy_(torch.from_numpy(want.view(np.int64)))
        for oo in (0, 16, 5, 77, 4095):
            obuf.fill_(0x2A)
            cn.bits_to_n_dev(d, n_len, out=obuf[ob + oo : ob + oo + n_len])
            got = obuf.cpu().numpy()
            assert (got[: ob + oo] == 0x2A).all() and (got[ob + oo + n_len :] == 0x2A).all(), (p_off, oo)
            assert np.array_equal(got[ob + oo : ob + oo + n_len], want_back), (p_off, oo)
        # a shorter length from the same words (len < capacity), still >= 2^20
        m = (1 << 20) + 7
        obuf.fill_(0x2A)
        cn.bits_to_n_dev(d, m, out=obuf[3 : 3 + m])
        got = obuf.cpu().numpy()
        assert (got[:3] == 0x2A).all() and (got[3 + m :] == 0x2A).all() and np.array_equal(got[3 : 3 + m], want_back[:m]), p_off


def test_window_decoder_at_every_packed_phase_with_the_gate_forced_open(cn, oracle, torch_cuda, tuning):
    """bits_to_n_window is the shipped kernel for decodes past the Infinity Cache whose packed stream is off
    its lines or dwords, and its only GPU coverage needed 4 GiB and reached q in {1, 2, 29, 30}, sh in {0, 2, 6, 22}.  Lab build,
    gate forced to 1 nt (decode_cache_log2 = 0), at most 64 tiles per launch (the several-launch loop: edges in the last one
    only): ALL 32 dword phases q x ALL 16 bit phases sh -- the worst case q = 31 with sh != 0 reaches the far end of the slab's
    second row and the 144-byte slack behind the tile -- at 2^20 + a ragged bit, guard pages on both sides of the output,
    against a decode of the same words on the grid (itself checked against bits_to_n_lut)."""
    from cute_nucleotides_amd import _lib

    torch = torch_cuda
    tuning.set_tuning("decode_cache_log2", 0)
    tuning.set_tuning("launch_tiles", 64)
    L = _lib.lib()
    try:
        n_len = (1 << 20) + 4096 * 3 + 1234
        bits = np.random.default_rng(83).integers(0, 2**64, (n_len + 31) // 32 + 4, dtype=np.uint64)
        want = torch.from_numpy(oracle.bits_to_n_lut(bits, n_len)).cuda()
        pbuf = torch.zeros(bits.size + 1024, dtype=torch.int64, device="cuda")
        obuf = torch.empty(n_len + 4 * 4096, dtype=torch.uint8, device="cuda")
        pb = ((-pbuf.data_ptr()) % 4096) // 8
        ob = 4096 + (-obuf.data_ptr()) % 4096
        seen = set()
        plan = (ctypes.c_uint64 * 7)()
        for p_off in (0, 8, 1016):  # packed bytes off the page: dword phases come from here AND from the output's peel
            d = pbuf[pb + p_off // 8 : pb + p_off // 8 + bits.size]
            d.copy_(torch.from_numpy(bits.view(np.int64)))
            for a_off in range(0, 512) if p_off == 0 else range(3, 512, 37):
                out = obuf[ob + a_off : ob + a_off + n_len]
                assert L.cnt_test_decode_plan(d.data_ptr(), out.data_ptr(), n_len, 1, plan) == 0
                seen.add((int(plan[4]), int(plan[3]) // 2, int(plan[5])))
                obuf.fill_(0x2A)
                cn.bits_to_n_dev(d, n_len, out=out)
                assert torch.equal(out, want), (p_off, a_off, list(plan))
                assert bool((obuf[: ob + a_off] == 0x2A).all()) and bool((obuf[ob + a_off + n_len :] == 0x2A).all()), (p_off, a_off)
        assert {(q, sh) for q, sh, w in seen if w} >= {(q, sh) for q in range(32) for sh in range(16)} - {(0, 0)}  # every phase pair ran through the window kernel
        assert (0, 0, 0) in seen  # ... and the grid itself through the stream kernel
    finally:
        tuning.set_tuning("decode_cache_log2", -1)


def test_decode_past_the_infinity_cache_off_the_grid(cn, oracle, torch_cuda, fullsize):
    """Calls of more than 2^30 nt whose packed pointer is off its 128-B lines / dwords take bits_to_n_window, and their XCD
    turns are placed on the packed buffer's pages by peeling 0-3 (+4) further output pages (round 5, device_tier.inc
    decode_plan; tests/test_decode_plan.py walks the arithmetic).  2^30 + a ragged bit nucleotides of seeded random ACGT,
    encoded by the aligned call, then decoded from packed words at offsets that make every k, with dword phases (stream of
    words at byte offsets 8 ... 4088), bit phases (output offsets 5, 77) and a head of up to 5 pages riding in the edge
    items: the decoded text equals the input on the device, guards around it survive, and sampled chunks equal
    bits_to_n_lut of the host-regenerated input."""
    from conftest import need_free_hbm

    from cute_nucleotides_amd import _lib, devutil

    assert not _lib.is_lab_build()
    torch = torch_cuda
    need_free_hbm(4)
    n_len = (1 << 30) + 4096 * 7 + 1234
    d_in = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d_in, 41)
    words = (n_len + 31) // 32
    pbuf = torch.zeros(words + 1024, dtype=torch.int64, device="cuda")
    obuf = torch.empty(n_len + 3 * 4096, dtype=torch.uint8, device="cuda")
    pb = ((-pbuf.data_ptr()) % 4096) // 8
    ob = 4096 + (-obuf.data_ptr()) % 4096
    chunk = 1 << 20
    ms = None
    for p_off, o_off in ((8, 0), (264, 0), (1032, 16), (2056, 5), (3080, 77), (4088, 4095), (1024, 0), (520, 2048)):
        d_pk = pbuf[pb + p_off // 8 : pb + p_off // 8 + words]
        cn.n_to_bits_dev(d_in, out=d_pk)
        obuf.fill_(0x2A)
        view = obuf[ob + o_off : ob + o_off + n_len]
        _, ms = _timed_ms(torch, lambda: cn.bits_to_n_dev(d_pk, n_len, out=view))
        assert devutil.count_mismatch(d_in, view) == 0, (p_off, o_off)
        assert bool((obuf[: ob + o_off] == 0x2A).all()) and bool((obuf[ob + o_off + n_len :] == 0x2A).all()), (p_off, o_off)
        for lo in (0, (n_len // 2) // 32 * 32, n_len - chunk):  # head, middle, ragged end against the oracle
            host = oracle.fill_random_acgt(chunk + 32, 41, first_nt=lo // 32 * 32)[lo % 32 : lo % 32 + chunk]
            assert np.array_equal(view[lo : lo + chunk].cpu().numpy(), host), (p_off, o_off, lo)
    fullsize(30, ms, config="decode off the grid past the Infinity Cache (window kernel + turn placement)", nt=n_len)


# ---- boundary behaviour of the C ABI -------------------------------------------------------
def test_output_pointer_only_8_byte_aligned(cn, oracle, torch_cuda):
    """u64 outputs need 8-B alignment; 16-B is only needed for the fast path."""
    torch = torch_cuda
    n = _rand_valid(100000 + 7, 123)
    want = oracle.n_to_bits_lut(n)
    buf = torch.full((want.size + 3,), -1, dtype=torch.int64, device="cuda")
    out = buf[1:]  # data_ptr % 16 == 8
    assert out.data_ptr() % 16 == 8
    got = cn.n_to_bits_dev(torch.from_numpy(n).cuda(), out=out).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    assert int(buf[0].item()) == -1 and int(buf[want.size + 1].item()) == -1
    bits = torch.from_numpy(want.view(np.int64)).cuda()
    dbuf = torch.zeros(want.size + 1, dtype=torch.int64, device="cuda")
    dbuf[1:].copy_(bits)
    got = cn.bits_to_n_dev(dbuf[1:], n.size).cpu().numpy()  # input % 16 == 8
    assert np.array_equal(got, oracle.bits_to_n_lut(want, n.size))


def test_host_tier_is_thread_safe_and_survives_shutdown(cn, oracle):
    import threading

    from cute_nucleotides_amd import _lib

    inputs = [_rand_valid(3_000_000 + 17 * k, k) for k in range(6)]
    wants = [oracle.n_to_bits_lut(x) for x in inputs]
    results = [None] * len(inputs)

    def work(k):
        for _ in range(3):
            bits = cn.n_to_bits_hip(inputs[k])
            back = cn.bits_to_n_hip(bits, inputs[k].size)
            results[k] = (bits, back)
        _lib.lib().cnt_shutdown()  # frees this thread's streams / scratch / copy helpers

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(inputs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for k, (bits, back) in enumerate(results):
        assert np.array_equal(bits, wants[k])
        assert bytes(back) == bytes(inputs[k]).upper().replace(b"U", b"T")
    # the calling thread can shut down and keep going
    assert _lib.lib().cnt_shutdown() == 0
    assert np.array_equal(cn.n_to_bits_hip(inputs[0]), wants[0])


def test_large_ragged_sizes_64bit_indexing(cn, oracle, torch_cuda):
    """> 2^32 nt with a ragged tail: the whole-tile kernels, the tail kernel's 64-bit word offsets and
    the zero padding of the last word, checked at the very end of the buffer and by round trip."""
    from cute_nucleotides_amd import devutil

    torch = torch_cuda
    n_len = (1 << 33) + 3 * 16384 + 2048 + 77
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 77)
    packed = cn.n_to_bits_dev(d)
    back = cn.bits_to_n_dev(packed, n_len)
    assert devutil.count_mismatch(d, back) == 0
    tail_nt = 3 * 16384 + 2048 + 77  # starts on a 32-aligned position: 2^33 % 32 == 0
    host_tail = oracle.fill_random_acgt(tail_nt, 77, first_nt=1 << 33)
    want = oracle.n_to_bits_lut(host_tail)
    got = packed[(1 << 33) // 32 :].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    assert int(got[-1]) >> (2 * (n_len & 31)) == 0
    # strict mode takes the same path with the validity filter
    strict = cn.n_to_bits_dev(d, strict_lut=True)
    assert devutil.count_mismatch(strict, packed) == 0


def test_host_tier_small_call_path_staged_kernels(cn, oracle):
    """Calls up to 2^20 nt take the zero-copy path: the shim's pinned staging buffer, zero-padded to a whole word,
    read by n_to_bits_staged with 16-B loads (bits_to_n_staged writes whole words into the pinned result buffer),
    completion through the pinned flag.  Every length 1..200, the reference's bench size, the path's upper limit and
    one past it (pipeline), both encode semantics on arbitrary bytes, decode of prefixes, and many calls back to
    back (the flag's tick must never be confused with an earlier call's)."""
    rng = np.random.default_rng(77)
    sizes = list(range(1, 201)) + [4095, 4096, 4097, 40000, 65535, (1 << 20) - 1, 1 << 20, (1 << 20) + 1]
    for n_len in sizes:
        n = rng.integers(0, 256, n_len, dtype=np.uint8)
        assert np.array_equal(cn.n_to_bits_hip(n, strict_lut=True), oracle.n_to_bits_lut(n)), n_len
        v = _rand_valid(n_len, n_len)
        bits = cn.n_to_bits_hip(v)
        want = oracle.n_to_bits_lut(v)
        assert np.array_equal(bits, want), n_len
        if n_len <= 4097:  # default semantics: (byte>>1)&3 on EVERY byte incl. the ragged tail (include/cute_nt.h)
            pad = np.zeros(-(-n_len // 32) * 32, dtype=np.uint8)
            pad[:n_len] = n
            codes = ((pad >> 1) & 3).astype(np.uint64).reshape(-1, 32)
            assert np.array_equal(cn.n_to_bits_hip(n), (codes << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)), n_len
        assert np.array_equal(cn.bits_to_n_hip(bits, n_len), oracle.bits_to_n_lut(want, n_len)), n_len
        if n_len > 3:
            assert np.array_equal(cn.bits_to_n_hip(bits, n_len - 3), oracle.bits_to_n_lut(want, n_len - 3)), n_len
    # arbitrary words decode (bits beyond len ignored), guard bytes behind the caller's buffer untouched
    words = rng.integers(0, 2**63, 1250, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    import ctypes

    from cute_nucleotides_amd import _lib

    buf = np.full(40000 + 64, 0x2A, dtype=np.uint8)
    assert _lib.lib().cnt_bits_to_n(ctypes.c_void_p(words.ctypes.data), 1250, 39990, ctypes.c_void_p(buf.ctypes.data)) == 0
    assert np.array_equal(buf[:39990], oracle.bits_to_n_lut(words, 39990)) and (buf[39990:] == 0x2A).all()
    a, b = _rand_valid(40000, 1), _rand_valid(40000, 2)
    wa, wb = oracle.n_to_bits_lut(a), oracle.n_to_bits_lut(b)
    for k in range(2000):  # alternate inputs so a stale result would be caught
        got = cn.n_to_bits_hip(a if k & 1 else b)
        assert np.array_equal(got, wa if k & 1 else wb), k


def test_small_call_path_is_thread_safe(cn, oracle):
    """eight threads hammer the zero-copy small-call path (own pinned staging, own completion word per thread and
    device) with different inputs of different sizes; every result must be its own"""
    import threading

    bad = []

    def work(k):
        rng = np.random.default_rng(k)
        inputs = [_rand_valid(int(s), 1000 * k + i) for i, s in enumerate(rng.integers(1, 70000, 6))]
        wants = [oracle.n_to_bits_lut(x) for x in inputs]
        for it in range(400):
            j = it % len(inputs)
            bits = cn.n_to_bits_hip(inputs[j])
            if not np.array_equal(bits, wants[j]):
                bad.append((k, it, "enc"))
                return
            if it % 7 == 0 and not np.array_equal(cn.bits_to_n_hip(bits, inputs[j].size), oracle.bits_to_n_lut(wants[j], inputs[j].size)):
                bad.append((k, it, "dec"))
                return

    ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad[:5]


# ---- CNT_TAIL_LUT: the reference's SIMD encoders to the letter, on ARBITRARY bytes ---------------------------------
TAIL_SIZES = [1, 5, 31, 32, 33, 63, 65, 100, 2048 + 17, 4096 + 31, 40000 + 13, 100003, (1 << 20) + 7, (1 << 20) + 32, (1 << 22) + 16384 + 29]


def test_tail_lut_equals_the_simd_variants_on_arbitrary_bytes(cn, oracle, torch_cuda, tuning):
    """n_to_bits_{pext,shift,movemask,mul} extract bits 1..2 on whole 32-nt blocks and send the ragged end through
    BYTE_LUT (n_to_bits.rs:109-111,160-162,201-203,253-255): on bytes outside the alphabet neither the default mode
    ((b>>1)&3 everywhere) nor CNT_STRICT_LUT (table everywhere) equals them -- default | CNT_TAIL_LUT does, at any
    length, through the host tier, the device tier (tiles + edge workgroups, generic kernel) and the sharded tiers."""
    from cute_nucleotides_amd import sharding

    if not oracle.port_cpu_ok():
        pytest.skip("host CPU lacks AVX2/BMI2: the SIMD ports cannot run")
    torch = torch_cuda
    rng = np.random.default_rng(31)
    prev = sharding.alias_devices(True)
    try:
        for n_len in TAIL_SIZES:
            n = rng.integers(0, 128, n_len, dtype=np.uint8)  # any 7-bit byte: 88 of the 128 differ between table and bit extraction
            want = oracle.n_to_bits_movemask(n)
            for port in (oracle.n_to_bits_pext, oracle.n_to_bits_shift, oracle.n_to_bits_mul):
                assert np.array_equal(port(n), want), (n_len, port.__name__)  # the four reference variants agree with each other
            assert np.array_equal(cn.n_to_bits_hip(n, tail_lut=True), want), n_len  # host tier (staged small path / pipeline)
            d = torch.from_numpy(n).cuda()
            for small_nt in (0, 1 << 17):  # tiles + edge workgroups, and the generic kernel alone
                tuning.set_tuning("small_nt", small_nt)
                got = cn.n_to_bits_dev(d, tail_lut=True).cpu().numpy().view(np.uint64)
                assert np.array_equal(got, want), (n_len, small_nt)
            off = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")  # the window kernel's launch (input phase != 0)
            off[13 : 13 + n_len].copy_(d)
            assert np.array_equal(cn.n_to_bits_dev(off[13 : 13 + n_len], tail_lut=True).cpu().numpy().view(np.uint64), want), n_len
            f_bits, _ = cn.round_trip_dev(d, tail_lut=True)
            assert np.array_equal(f_bits.cpu().numpy().view(np.uint64), want), n_len
            for ndev in (1, 3, 8):
                assert np.array_equal(cn.n_to_bits_hip_sharded(n, ndev=ndev, tail_lut=True), want), (n_len, ndev)
            (s_bits,) = sharding.n_to_bits_sharded_dev([d], tail_lut=True)
            assert np.array_equal(s_bits.cpu().numpy().view(np.uint64), want), n_len
            assert np.array_equal(want, oracle.n_to_bits_bitextract(n)), n_len  # the oracle's scalar statement of the same rule
            # without the flag the default stays (b>>1)&3 everywhere; with CNT_STRICT_LUT the table everywhere
            codes = ((np.concatenate([n, np.zeros(-n_len % 32, dtype=np.uint8)]) >> 1) & 3).astype(np.uint64).reshape(-1, 32)
            everywhere = (codes << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
            assert np.array_equal(cn.n_to_bits_hip(n), everywhere), n_len
            assert np.array_equal(cn.n_to_bits_hip(n, strict_lut=True, tail_lut=True), oracle.n_to_bits_lut(n)), n_len
    finally:
        sharding.alias_devices(prev)


# ---- one launch per call: head and ragged end ride in the tile kernel's grid --------------------------------------
def _hip_runtime():
    """the HIP runtime already mapped in this process (torch's), for the three graph calls the test needs"""
    import ctypes

    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    assert paths, "no HIP runtime mapped"
    return ctypes.CDLL(paths[0])


def _kernel_nodes_of(torch, fn):
    """enqueue fn() under stream capture and count the nodes of the captured graph"""
    import ctypes

    hip = _hip_runtime()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()  # module load / lazy init outside the capture
        side.synchronize()
        s = ctypes.c_void_p(side.cuda_stream)
        graph = ctypes.c_void_p()
        assert hip.hipStreamBeginCapture(s, 2) == 0  # hipStreamCaptureModeRelaxed
        try:
            fn()
        finally:
            rc = hip.hipStreamEndCapture(s, ctypes.byref(graph))
        assert rc == 0
        n = ctypes.c_size_t(0)
        assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) == 0
        hip.hipGraphDestroy(graph)
    return n.value


def test_any_size_and_alignment_is_one_launch(cn, oracle, torch_cuda):
    """encode_dev / decode_dev used to enqueue up to three kernels (head peel, tiles, ragged end);
    each extra launch cost ~5 us behind a 0.2 ms kernel at BASELINE.json's 1 GiB size.  The head words and the ragged
    end are now extra workgroups of the tile kernel's own grid: ONE node in a captured graph for every size and
    pointer phase -- and the results are still the oracle's."""
    torch = torch_cuda  # the PRODUCT build as shipped: 2^21 - 19 nt is past the small-input path (2^17)
    n_len = (1 << 21) - 19
    host = _rand_valid(n_len, 4)
    want = oracle.n_to_bits_lut(host)
    want_back = oracle.bits_to_n_lut(want, n_len)  # the canonical spelling: upper case, U -> T
    ibuf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
    pbuf = torch.zeros(n_len // 32 + 64, dtype=torch.int64, device="cuda")
    obuf = torch.zeros(n_len + 8192, dtype=torch.uint8, device="cuda")
    words = (n_len + 31) // 32
    for io, po, oo in ((0, 0, 0), (0, 3, 0), (5, 0, 77), (64, 1, 4095), (127, 7, 1), (16, 0, 16)):
        view = ibuf[io : io + n_len]
        view.copy_(torch.from_numpy(host))
        packed = pbuf[po : po + words]
        out = obuf[oo : oo + n_len]
        assert _kernel_nodes_of(torch, lambda: cn.n_to_bits_dev(view, out=packed)) == 1, (io, po)
        assert np.array_equal(packed.cpu().numpy().view(np.uint64), want), (io, po)
        assert _kernel_nodes_of(torch, lambda: cn.bits_to_n_dev(packed, n_len, out=out)) == 1, (po, oo)
        assert np.array_equal(out.cpu().numpy(), want_back), (po, oo)
    # the fused round trip on aligned pointers: its ragged end (here 2^21 - 19 = 511 tiles of 4096 nt + 4077 nt) rides along too
    view, packed, out = ibuf[:n_len], pbuf[:words], obuf[:n_len]
    view.copy_(torch.from_numpy(host))
    packed.zero_()
    out.zero_()
    assert _kernel_nodes_of(torch, lambda: cn.round_trip_dev(view, out_bits=packed, out_n=out)) == 1
    assert np.array_equal(packed.cpu().numpy().view(np.uint64), want) and np.array_equal(out.cpu().numpy(), want_back)
    # ... and at ANY alignment of its three pointers (it used to degrade to two launches off the 128-B grid)
    for io, po, oo in ((5, 0, 77), (64, 1, 4095), (127, 7, 1), (0, 3, 0), (0, 0, 16), (1, 0, 0)):
        view, packed, out = ibuf[io : io + n_len], pbuf[po : po + words], obuf[oo : oo + n_len]
        view.copy_(torch.from_numpy(host))
        packed.zero_()
        out.zero_()
        assert _kernel_nodes_of(torch, lambda: cn.round_trip_dev(view, out_bits=packed, out_n=out)) == 1, (io, po, oo)
        assert np.array_equal(packed.cpu().numpy().view(np.uint64), want) and np.array_equal(out.cpu().numpy(), want_back), (io, po, oo)
    # shorter than one tile: the generic kernel alone, also one launch
    small = ibuf[3 : 3 + 1000]
    assert _kernel_nodes_of(torch, lambda: cn.n_to_bits_dev(small, out=pbuf[:32])) == 1
    assert _kernel_nodes_of(torch, lambda: cn.round_trip_dev(small, out_bits=pbuf[:32], out_n=obuf[9 : 9 + 1000])) == 1


# ---- BASELINE.json configs[4]: every rank's shard of the 256 GiB job, at its GLOBAL offset ---------------------------
@pytest.mark.parametrize("k", range(8))
def test_config4_rank_shard_at_its_global_offset(cn, oracle, torch_cuda, fullsize, k):
    """"256 GiB encode chunk-sharded across 8 x MI355X": rank k's exact workload -- the 2^35-nt chunk
    [k * 2^35, (k+1) * 2^35) of the global counter-based ACGT stream -- on this GPU: partition from the C library
    (cnt_shard_range), encode through the device-resident sharded entry point, sampled 16 Mi-nt chunks and the last
    words against n_to_bits_lut of the oracle REGENERATED AT THE GLOBAL OFFSETS, position-salted checksums at the
    global word index, and a device round trip.  Word w depends on nt [32w, 32w+32) only (n_to_bits.rs:38-43), so the
    eight shards' words concatenated ARE the 256 GiB buffer's words."""
    from cute_nucleotides_amd import devutil, sharding

    torch = torch_cuda
    world, shard_nt, seed = 8, 1 << 35, 0x5EED + 4
    n_global = world * shard_nt
    lo, hi = sharding.shard_range_c(n_global, world, k)
    assert (lo, hi) == (k * shard_nt, (k + 1) * shard_nt) == sharding.partition(n_global, world)[k]
    need_free_hbm(76)  # 32 + 8 + 32 GiB resident
    d = torch.empty(hi - lo, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, seed, first_nt=lo)
    packed = torch.empty((hi - lo) // 32, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    (bits,), (ms,) = sharding.n_to_bits_sharded_dev([d], outs=[packed], want_ms=True)
    fullsize(35, ms, config="configs[4] rank %d of 8" % k, first_nt=lo, gnts=round((hi - lo) / (ms * 1e-3) / 1e9, 1))
    chunk_nt = 16 << 20
    chunk_w = chunk_nt // 32
    n_chunks = (hi - lo) // chunk_nt
    rng = np.random.default_rng(100 + k)
    for c in sorted({0, n_chunks - 1, int(rng.integers(1, n_chunks - 1)), int(rng.integers(1, n_chunks - 1))}):
        host_n = oracle.fill_random_acgt(chunk_nt, seed, first_nt=lo + c * chunk_nt)  # GLOBAL offset
        assert np.array_equal(d[c * chunk_nt : c * chunk_nt + 4096].cpu().numpy(), host_n[:4096]), (k, c)  # the shard holds the global stream
        want = oracle.n_to_bits_lut(host_n)
        got = bits[c * chunk_w : (c + 1) * chunk_w].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), (k, c)
        first_word = lo // 32 + c * chunk_w  # global word index: the salt of the checksum
        assert devutil.checksum_words(bits[c * chunk_w : (c + 1) * chunk_w], first_word=first_word) == oracle.checksum_words(want, first_word=first_word), (k, c)
    back = torch.empty(hi - lo, dtype=torch.uint8, device="cuda")
    sharding.bits_to_n_sharded_dev([bits], [hi - lo], outs=[back])
    assert devutil.count_mismatch(d, back) == 0
    # the ragged end of the job: the last rank's shard minus 19 nt -> 13 nucleotides in the final word, zero-padded
    if k == world - 1:
        r = (hi - lo) - 19
        (rb,) = sharding.n_to_bits_sharded_dev([d[:r]], outs=[packed])
        tail_n = oracle.fill_random_acgt(chunk_nt, seed, first_nt=hi - chunk_nt)[: chunk_nt - 19]
        want = oracle.n_to_bits_lut(tail_n)
        got = rb[-(chunk_w) :].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want) and int(got[-1]) >> 26 == 0


@pytest.mark.parametrize("xs", [0, 1, 2, 3, 4, 5])
def test_tile_maps_are_bijections_for_any_xcd_count(cn, oracle, torch_cuda, lab_build, xs):
    """the XCD count is asked of the device and reaches the kernels as log2 X.  Partition modes
    (CPX / DPX / QPX: 1 / 2 / 4 XCDs per device) cannot be switched on here, so the tuning key "xcd_shift" walks every
    value a device could answer (and two it could not): the block -> tile maps of every kernel family -- encode pairs,
    decode quads, fused, window / shifted twins, the 5-letter quads -- must stay bijections, i.e. results stay the
    oracle's, including sizes whose last group of X*C tiles is ragged."""
    from cute_nucleotides_amd import devutil

    torch = torch_cuda
    saved_small = devutil.get_tuning("small_nt")
    devutil.set_tuning("small_nt", 0)
    devutil.set_tuning("xcd_shift", xs)
    try:
        assert devutil.get_tuning("xcd_shift") == xs
        for n_len in (2048 * 7, 4096 * 129 + 5, 2048 * (16 << xs) + 2048 * 3 + 17, (1 << 22) + 4096 * 5 + 31):
            host = _rand_valid(n_len, 40 + xs)
            want = oracle.n_to_bits_lut(host)
            back_want = oracle.bits_to_n_lut(want, n_len)
            for off in (0, 7):  # aligned: stream kernels; +7: window / shifted twins
                buf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
                d = buf[off : off + n_len]
                d.copy_(torch.from_numpy(host))
                bits = cn.n_to_bits_dev(d)
                assert np.array_equal(bits.cpu().numpy().view(np.uint64), want), (xs, n_len, off)
                obuf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
                cn.bits_to_n_dev(bits, n_len, out=obuf[off : off + n_len])
                assert np.array_equal(obuf[off : off + n_len].cpu().numpy(), back_want), (xs, n_len, off)
            f_bits, f_back = cn.round_trip_dev(buf[:n_len].copy_(torch.from_numpy(host)))
            assert np.array_equal(f_bits.cpu().numpy().view(np.uint64), want) and np.array_equal(f_back.cpu().numpy(), back_want), (xs, n_len)
        n5 = oracle.fill_random_acgtn(3456 * ((8 << xs) + 3) + 100, 60 + xs)
        w5 = oracle.n_to_bits2_lut(n5)
        d5 = torch.from_numpy(n5).cuda()
        b5 = cn.n_to_bits2_dev(d5)
        assert np.array_equal(b5.cpu().numpy().view(np.uint64), w5), xs
        assert np.array_equal(cn.bits_to_n2_dev(b5, n5.size).cpu().numpy(), oracle.bits_to_n2_lut(w5, n5.size)), xs
    finally:
        devutil.set_tuning("xcd_shift", -1)
        devutil.set_tuning("small_nt", saved_small)
    assert devutil.get_tuning("xcd_shift") in (0, 1, 2, 3)  # back to the device's own answer (8 XCDs -> 3 on an MI355X in SPX mode)


def test_host_tier_outputs_at_odd_offsets_in_fresh_and_warm_pages(cn, oracle):
    """The copy team cuts a copy-out into blocks on the DESTINATION's 1-MiB (warm) or 2-MiB (fresh pages) grid: outputs
    that start anywhere inside a page, in mappings whose pages do not exist yet (the first call) and in the same
    mappings once they do (the second call), with guard bytes on both sides."""
    import ctypes

    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    n_len = (48 << 20) + 12345  # three 16-Mi-nt chunks and a ragged end; decode's output is > 8 MiB: "fresh" is detected
    n = oracle.fill_random_acgt(n_len, 77)
    words = (n_len + 31) // 32
    want = oracle.n_to_bits_movemask(n)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    for off in (0, 8, 4096 + 24, (2 << 20) - 8, (1 << 20) + 40):
        obuf = np.empty(words * 8 + (4 << 20), dtype=np.uint8)  # a fresh mapping: nothing below touches it before the call
        dbuf = np.empty(n_len + (4 << 20), dtype=np.uint8)
        for attempt in ("fresh", "warm"):
            out = obuf[off : off + words * 8].view(np.uint64)
            assert L.cnt_n_to_bits(p(n), n_len, p(out), words) == 0
            assert np.array_equal(out, want), (off, attempt)
            back = dbuf[off + 3 : off + 3 + n_len]
            assert L.cnt_bits_to_n(p(out), words, n_len, p(back)) == 0
            assert np.array_equal(back, n), (off, attempt)
            if attempt == "fresh":  # guards go in once the pages exist; the warm pass must leave them alone
                obuf[:off] = 0xA5
                obuf[off + words * 8 :] = 0xA5
                dbuf[: off + 3] = 0x5A
                dbuf[off + 3 + n_len :] = 0x5A
        assert (obuf[:off] == 0xA5).all() and (obuf[off + words * 8 :] == 0xA5).all(), off
        assert (dbuf[: off + 3] == 0x5A).all() and (dbuf[off + 3 + n_len :] == 0x5A).all(), off
