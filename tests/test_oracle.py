"""Pins the CPU oracle to the reference: every known-answer vector the reference's own
unit tests hold (tests/golden/reference_kats.json, lifted from src/n_to_bits.rs:408-470
and src/n_to_bits2.rs:270-299), then cross-checks the three independent restatements
(scalar C, x86 SIMD ports, numpy) against each other on random and edge-case inputs.
CPU only."""
import numpy as np
import pytest

VALID = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)


def _words(hexes):
    return np.array([int(h, 16) for h in hexes], dtype=np.uint64)


def test_every_reference_kat(oracle, kats):
    assert len(kats["kats"]) == 20
    seen = set()
    for k in kats["kats"]:
        fn = getattr(oracle, k["fn"])
        seen.add(k["fn"])
        if k["kind"] == "encode":
            got = fn(k["input_ascii"].encode())
            assert got.tolist() == _words(k["expected_words_hex"]).tolist(), k
        else:
            got = fn(_words(k["input_words_hex"]), k["len"])
            assert bytes(got) == k["expected_ascii"].encode(), k
    # all 13 reference functions are covered by at least one vector
    assert seen == {
        "n_to_bits_lut", "bits_to_n_lut", "n_to_bits_pext", "n_to_bits_shift", "n_to_bits_movemask",
        "n_to_bits_mul", "bits_to_n_shuffle", "bits_to_n_pdep", "bits_to_n_clmul",
        "n_to_bits2_lut", "bits_to_n2_lut", "n_to_bits2_pext", "bits_to_n2_pdep",
    }


def test_bench_generator_inputs(oracle, kats):
    # benches/bench_n_to_bits.rs:68-78: "ATCG"*10000 -> 1250 words of 0xD8D8...
    g = {b["generator"]: b for b in kats["bench_inputs"]}
    n = (g["get_nucleotides"]["unit"] * g["get_nucleotides"]["repeat"]).encode()
    bits = oracle.n_to_bits_lut(n)
    assert bits.size == 1250 and (bits == np.uint64(0xD8D8D8D8D8D8D8D8)).all()
    assert bytes(oracle.bits_to_n_lut(bits, 40000)) == n
    n5 = (g["get_nucleotides_undetermined"]["unit"] * g["get_nucleotides_undetermined"]["repeat"]).encode()
    b5 = oracle.n_to_bits2_lut(n5)
    assert b5.size == (40000 + 26) // 27
    assert bytes(oracle.bits_to_n2_lut(b5, 40000)) == n5
    # period of "ATCGN"*k in 27-nt words is 5 words (135 nt); first word is the KAT
    assert int(b5[0]) == 0x36A45D1F46D48BA3


@pytest.mark.parametrize("n_len", [0, 1, 3, 4, 31, 32, 33, 63, 64, 65, 1000, 4096, 40000, 100003])
def test_encoders_agree_on_valid_alphabet(oracle, n_len):
    rng = np.random.default_rng(n_len)
    n = VALID[rng.integers(0, VALID.size, n_len)]
    want = oracle.np_n_to_bits_lut(n)
    assert want.size == (n_len + 31) // 32
    for name in ("n_to_bits_lut", "n_to_bits_bitextract", "n_to_bits_pext", "n_to_bits_shift",
                 "n_to_bits_movemask", "n_to_bits_mul"):
        got = getattr(oracle, name)(n)
        assert np.array_equal(got, want), name


@pytest.mark.parametrize("n_len", [0, 1, 5, 31, 32, 33, 64, 999, 40000])
def test_decoders_agree_and_round_trip(oracle, n_len):
    rng = np.random.default_rng(1000 + n_len)
    words = (n_len + 31) // 32
    bits = rng.integers(0, 2**64, words, dtype=np.uint64)
    want = oracle.np_bits_to_n_lut(bits, n_len)
    for name in ("bits_to_n_lut", "bits_to_n_shuffle", "bits_to_n_pdep", "bits_to_n_clmul"):
        got = getattr(oracle, name)(bits, n_len)
        assert np.array_equal(got, want), name
    # decode -> encode returns the packed words with the unused high bits cleared
    back = oracle.n_to_bits_lut(want)
    if n_len & 31 and words:
        bits = bits.copy()
        bits[-1] &= np.uint64((1 << (2 * (n_len & 31))) - 1)
    assert np.array_equal(back, bits)


def test_len_guard_matches_reference_panic(oracle):
    bits = np.zeros(2, dtype=np.uint64)
    for name in ("bits_to_n_lut", "bits_to_n_shuffle", "bits_to_n_pdep", "bits_to_n_clmul"):
        with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
            getattr(oracle, name)(bits, 65)
        assert getattr(oracle, name)(bits, 64).size == 64
    with pytest.raises(ValueError):
        oracle.bits_to_n2_lut(bits, 55)
    with pytest.raises(ValueError):
        oracle.bits_to_n2_pdep(bits, 55)


def test_invalid_bytes_documented_divergence(oracle):
    """Reference LUT -> 0 for non-ACGTU; reference SIMD variants -> (c>>1)&3 on full
    32-nt blocks (SURVEY 8a edge semantics).  The oracle reproduces both."""
    n = np.arange(256, dtype=np.uint8)
    lut = oracle.n_to_bits_lut(n)
    ext = oracle.n_to_bits_bitextract(n)
    for name in ("n_to_bits_pext", "n_to_bits_shift", "n_to_bits_movemask", "n_to_bits_mul"):
        assert np.array_equal(getattr(oracle, name)(n), ext), name
    codes_lut = [(int(lut[i >> 5]) >> (2 * (i & 31))) & 3 for i in range(256)]
    codes_ext = [(int(ext[i >> 5]) >> (2 * (i & 31))) & 3 for i in range(256)]
    valid = set(b"ACGTUacgtu")
    for c in range(256):
        assert codes_ext[c] == (c >> 1) & 3
        assert codes_lut[c] == (((c >> 1) & 3) if c in valid else 0)
    # 'N' is the classic: LUT says A(0), bit-extract says G(3)
    assert codes_lut[ord("N")] == 0 and codes_ext[ord("N")] == 3


@pytest.mark.parametrize("n_len", [0, 1, 2, 3, 4, 5, 26, 27, 28, 31, 32, 33, 53, 54, 55, 80, 81, 1000, 40000, 100003])
def test_five_letter_codec(oracle, n_len):
    rng = np.random.default_rng(77 + n_len)
    alpha = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
    n = alpha[rng.integers(0, alpha.size, n_len)]
    a = oracle.n_to_bits2_lut(n)
    b = oracle.n_to_bits2_pext(n)
    assert a.size == (n_len + 26) // 27
    assert np.array_equal(a, b)
    if a.size:
        assert int(a.max()) >> 63 == 0  # bit 63 never set
    up = bytes(n).upper().replace(b"U", b"T")
    assert bytes(oracle.bits_to_n2_lut(a, n_len)) == up
    assert bytes(oracle.bits_to_n2_pdep(a, n_len)) == up


def test_generator_is_uniform_and_chunkable(oracle):
    n = oracle.fill_random_acgt(1 << 20, 0x5EED)
    counts = np.bincount(n, minlength=128)
    assert set(np.nonzero(counts)[0].tolist()) == set(b"ACGT")
    for c in b"ACGT":
        assert abs(counts[c] / n.size - 0.25) < 0.005
    # any 32-aligned chunk can be regenerated independently
    part = oracle.fill_random_acgt(4099, 0x5EED, first_nt=64 * 1000)
    assert np.array_equal(part, n[64000 : 64000 + 4099])
    n5 = oracle.fill_random_acgtn(27 * 4096, 7)
    frac_n = (n5 == ord("N")).mean()
    assert 0.04 < frac_n < 0.085
    part5 = oracle.fill_random_acgtn(1000, 7, first_nt=27 * 100)
    assert np.array_equal(part5, n5[2700:3700])


def test_checksum_position_sensitive(oracle):
    w = np.arange(1000, dtype=np.uint64)
    c = oracle.checksum_words(w)
    assert c == (oracle.checksum_words(w[:300]) + oracle.checksum_words(w[300:], first_word=300)) % 2**64
    w2 = w.copy()
    w2[[10, 11]] = w2[[11, 10]]
    assert oracle.checksum_words(w2) != c


def test_oracle_selftest_under_asan_ubsan():
    """The C restatements and the SIMD ports on exactly-sized heap buffers under
    -fsanitize=address,undefined: no over-read / over-write, no UB (the reference has both)."""
    import os
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-C", here, "asan"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout) and "cannot find" in (r.stderr + r.stdout):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "oracle selftest ok" in r.stdout
