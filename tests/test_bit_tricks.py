"""Exhaustive CPU checks of the integer identities the HIP kernels rely on
(hip/codec2_kernels.hpp, codec5_kernels.hpp).  The formulas are
restated here in numpy uint32 arithmetic and compared with the oracle's definitions, so a
wrong shift/mask/magic constant is caught without a GPU."""
import numpy as np

U32 = np.uint32


def v_perm(src0, src1, sel):
    """v_perm_b32 for selector bytes 0..7: 0-3 pick src1 bytes, 4-7 pick src0 bytes."""
    src0, src1, sel = (np.asarray(a, dtype=np.uint64) for a in (src0, src1, sel))
    src0, src1, sel = np.broadcast_arrays(src0, src1, sel)
    pool = (src0 << np.uint64(32)) | src1
    out = np.zeros(pool.shape, dtype=np.uint64)
    for i in range(4):
        s = (sel >> np.uint64(8 * i)) & np.uint64(0xFF)
        assert (s < 8).all()
        out = out | ((pool >> (s * np.uint64(8))) & np.uint64(0xFF)) << np.uint64(8 * i)
    return out.astype(U32)


def all_dwords_of_bytes(byte_values):
    """every dword whose 4 bytes are drawn from byte_values (cartesian), capped for size"""
    b = np.array(byte_values, dtype=np.uint64)
    g = np.stack(np.meshgrid(b, b, b, b, indexing="ij"), -1).reshape(-1, 4)
    return (g[:, 0] | g[:, 1] << np.uint64(8) | g[:, 2] << np.uint64(16) | g[:, 3] << np.uint64(24)).astype(U32)


def enc_gather(y):
    u = ((y << U32(6)) | y).astype(U32)
    return ((u << U32(12)) | u).astype(U32)


def strict_filter(x):
    expect = v_perm(U32(0x47FF5554), U32(0x43FF41FF), x & U32(0x07070707))
    z = (x & U32(0xDFDFDFDF)) ^ expect
    nz = (((z & U32(0x7F7F7F7F)) + U32(0x7F7F7F7F)) | z) & U32(0x80808080)
    kill = (nz >> U32(5)) | (nz >> U32(6))
    return x & ~kill


def test_encode_gather_all_code_combinations():
    # every combination of the 4 two-bit codes, with arbitrary garbage in the other bits
    rng = np.random.default_rng(0)
    codes = np.arange(256, dtype=U32)
    for _ in range(8):
        garbage = rng.integers(0, 2**32, 256, dtype=np.uint64).astype(U32) & ~U32(0x06060606)
        x = garbage.copy()
        for k in range(4):
            x |= ((codes >> U32(2 * k)) & U32(3)) << U32(8 * k + 1)
        got = (enc_gather(x & U32(0x06060606)) >> U32(19)) & U32(0xFF)
        assert np.array_equal(got, codes)


def test_strict_filter_matches_byte_lut():
    valid = set(b"ACGTUacgtu")
    # all 256 byte values in every byte position (other positions random)
    rng = np.random.default_rng(1)
    for pos in range(4):
        base = rng.integers(0, 2**32, 256, dtype=np.uint64).astype(U32) & ~U32(0xFF << (8 * pos))
        x = base | (np.arange(256, dtype=U32) << U32(8 * pos))
        f = strict_filter(x)
        for c in range(256):
            got = (int(f[c]) >> (8 * pos + 1)) & 3
            want = ((c >> 1) & 3) if c in valid else 0
            assert got == want, (pos, c)
    # cartesian product over a mixed alphabet: each byte independent
    alpha = list(b"ACGTUacgtuNn@`\x00\x7f\x80\xc1\xff")
    x = all_dwords_of_bytes(alpha)
    f = strict_filter(x)
    for k in range(4):
        byte = (x >> U32(8 * k)) & U32(0xFF)
        want = np.where(np.isin(byte, list(valid)), (byte >> U32(1)) & U32(3), 0)
        assert np.array_equal((f >> U32(8 * k + 1)) & U32(3), want)


def test_decode_spread_and_lut():
    b = np.arange(256, dtype=U32)
    t = (b << U32(6)) | b
    sel = ((t << U32(12)) | t) & U32(0x03030303)
    out = v_perm(U32(0), U32(0x47544341), sel)
    lut = b"ACTG"  # n_to_bits.rs:23-30
    for v in range(256):
        want = bytes(lut[(v >> (2 * k)) & 3] for k in range(4))
        assert int(out[v]).to_bytes(4, "little") == want


def test_code5_tables():
    lut_lo, lut_hi = U32(0x01000000), U32(0x03040202)
    want_fast = {}
    for c in range(256):
        # reference SIMD: pshufb table on the low 3 bits, zero for bytes >= 0x80 (n_to_bits2.rs:127-136,151)
        table = {1: 0, 3: 1, 4: 2, 5: 2, 6: 4, 7: 3}
        want_fast[c] = 0 if c >= 0x80 else table.get(c & 7, 0)
    x = np.arange(256, dtype=U32) * U32(0x01010101)
    c = v_perm(lut_hi, lut_lo, x & U32(0x07070707))
    hi = x & U32(0x80808080)
    fast = c & ~((hi >> U32(5)) | (hi >> U32(6)) | (hi >> U32(7)))
    strict_want = {ord(ch): v for ch, v in zip("ACTUGNactugn", [0, 1, 2, 2, 3, 4] * 2)}
    expect = v_perm(U32(0x474E5554), U32(0x43FF41FF), x & U32(0x07070707))
    z = (x & U32(0xDFDFDFDF)) ^ expect
    nz = (((z & U32(0x7F7F7F7F)) + U32(0x7F7F7F7F)) | z) & U32(0x80808080)
    strict = c & ~((nz >> U32(5)) | (nz >> U32(6)) | (nz >> U32(7)))
    for v in range(256):
        assert int(fast[v]) & 0xFF == want_fast[v], v
        assert int(strict[v]) & 0xFF == strict_want.get(v, 0), v
        assert int(fast[v]) == want_fast[v] * 0x01010101


def test_digits3_magic_division():
    v = np.arange(128, dtype=U32)
    c = (v * U32(41)) >> U32(10)
    r = v - c * U32(25)
    b = (r * U32(13)) >> U32(6)
    a = r - b * U32(5)
    assert np.array_equal(c, v // 25)
    assert np.array_equal(b, (v // 5) % 5)
    assert np.array_equal(a, v % 5)
    letters = v_perm(U32(0x4E4E4E4E), U32(0x47544341), np.arange(5, dtype=U32))
    assert bytes(int(l) & 0xFF for l in letters) == b"ACTGN"  # n_to_bits2.rs:25-33


def test_generator_letter_constant():
    assert (0x54474341).to_bytes(4, "little") == b"ACGT"
    assert (0x47544341).to_bytes(4, "little") == b"ACTG"
