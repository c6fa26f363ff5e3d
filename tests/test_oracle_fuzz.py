"""Property-based (hypothesis) differential fuzz of the CPU oracle's three independent
restatements of the reference -- scalar C, x86 SIMD ports, numpy -- on arbitrary lengths and
byte contents.  CPU only; small sizes; the point is ragged lengths and odd bytes."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import cnt_oracle as orc

ALPHABET = b"ACGTUacgtu"
nuc = st.binary(min_size=0, max_size=300).map(lambda b: bytes(ALPHABET[x % 10] for x in b))
anybytes = st.binary(min_size=0, max_size=300)


@settings(max_examples=300, deadline=None)
@given(nuc)
def test_all_encoders_agree_on_the_alphabet(n):
    want = orc.np_n_to_bits_lut(n)
    for f in (orc.n_to_bits_lut, orc.n_to_bits_bitextract, orc.n_to_bits_pext, orc.n_to_bits_shift,
              orc.n_to_bits_movemask, orc.n_to_bits_mul):
        assert np.array_equal(f(n), want)
    back = orc.bits_to_n_lut(want, len(n))
    assert bytes(back) == n.upper().replace(b"U", b"T")


@settings(max_examples=300, deadline=None)
@given(anybytes)
def test_simd_ports_agree_with_each_other_on_any_bytes(n):
    ext = orc.n_to_bits_bitextract(n)
    for f in (orc.n_to_bits_pext, orc.n_to_bits_shift, orc.n_to_bits_movemask, orc.n_to_bits_mul):
        assert np.array_equal(f(n), ext)
    # the LUT oracle differs only where a byte is off the alphabet (documented divergence)
    lut = orc.n_to_bits_lut(n)
    full = len(n) // 32 * 32
    for i in range(full):
        c = n[i]
        lut_code = (int(lut[i >> 5]) >> (2 * (i & 31))) & 3
        ext_code = (int(ext[i >> 5]) >> (2 * (i & 31))) & 3
        assert ext_code == (c >> 1) & 3
        assert lut_code == (ext_code if c in ALPHABET else 0)


@settings(max_examples=300, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=2**64 - 1), min_size=0, max_size=12), st.data())
def test_all_decoders_agree(words, data):
    bits = np.array(words, dtype=np.uint64)
    length = data.draw(st.integers(min_value=0, max_value=32 * len(words)))
    want = orc.np_bits_to_n_lut(bits, length)
    for f in (orc.bits_to_n_lut, orc.bits_to_n_shuffle, orc.bits_to_n_pdep, orc.bits_to_n_clmul):
        assert np.array_equal(f(bits, length), want)


@settings(max_examples=300, deadline=None)
@given(st.binary(min_size=0, max_size=200).map(lambda b: bytes(b"ACGTNacgtnUu"[x % 12] for x in b)))
def test_five_letter_round_trip_and_port_agreement(n):
    a = orc.n_to_bits2_lut(n)
    assert np.array_equal(a, orc.n_to_bits2_pext(n))
    up = n.upper().replace(b"U", b"T")
    assert bytes(orc.bits_to_n2_lut(a, len(n))) == up
    assert bytes(orc.bits_to_n2_pdep(a, len(n))) == up
