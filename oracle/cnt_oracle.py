"""ctypes loader for oracle/libcnt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (see oracle/cnt_oracle.h).  The product package
(cute_nucleotides_amd) never imports this module.

Also holds `np_*` numpy restatements of the two scalar 2-bit functions (same
reference lines) used as a third, independent opinion in tests/test_oracle.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcnt_oracle.so")

ELEN_MESSAGE = "The length is greater than the number of nucleotides!"  # n_to_bits.rs:53


def build(force=False):
    """Compile oracle/libcnt_oracle.so with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("cnt_oracle.c", "cnt_simd_port.c", "cnt_oracle.h")]
    if (not force) and os.path.exists(_LIB_PATH) and all(
        os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs
    ):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcnt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, u64p, sz = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t
        for name in (
            "cnt_oracle_n_to_bits_lut",
            "cnt_oracle_n_to_bits_bitextract",
            "cnt_oracle_n_to_bits2_lut",
            "cnt_port_n_to_bits_pext",
            "cnt_port_n_to_bits_shift",
            "cnt_port_n_to_bits_movemask",
            "cnt_port_n_to_bits_mul",
            "cnt_port_n_to_bits2_pext",
        ):
            f = getattr(L, name)
            f.argtypes = [u8p, sz, u64p, sz]
            f.restype = ctypes.c_int
        for name in (
            "cnt_oracle_bits_to_n_lut",
            "cnt_oracle_bits_to_n2_lut",
            "cnt_port_bits_to_n_shuffle",
            "cnt_port_bits_to_n_pdep",
            "cnt_port_bits_to_n_clmul",
            "cnt_port_bits_to_n2_pdep",
        ):
            f = getattr(L, name)
            f.argtypes = [u64p, sz, sz, u8p]
            f.restype = ctypes.c_int
        L.cnt_oracle_words_for.argtypes = [sz]
        L.cnt_oracle_words_for.restype = sz
        L.cnt_oracle_words2_for.argtypes = [sz]
        L.cnt_oracle_words2_for.restype = sz
        L.cnt_port_cpu_ok.restype = ctypes.c_int
        L.cnt_oracle_hamming.argtypes = [u64p, u64p, sz]
        L.cnt_oracle_hamming.restype = ctypes.c_uint64
        L.cnt_oracle_complement.argtypes = [u64p, sz, u64p]
        L.cnt_oracle_complement.restype = None
        L.cnt_oracle_reverse_complement.argtypes = [u64p, sz, u64p]
        L.cnt_oracle_reverse_complement.restype = None
        L.cnt_oracle_validate.argtypes = [u8p, sz, ctypes.c_int]
        L.cnt_oracle_validate.restype = ctypes.c_uint64
        L.cnt_port_time_alloc_inclusive.argtypes = [ctypes.c_int, ctypes.c_void_p, sz, ctypes.c_int]
        L.cnt_port_time_alloc_inclusive.restype = ctypes.c_double
        L.cnt_oracle_fill_random_acgt.argtypes = [u8p, sz, sz, ctypes.c_uint64]
        L.cnt_oracle_fill_random_acgt.restype = None
        L.cnt_oracle_fill_random_acgtn.argtypes = [u8p, sz, sz, ctypes.c_uint64]
        L.cnt_oracle_fill_random_acgtn.restype = None
        L.cnt_oracle_checksum_words.argtypes = [u64p, sz, sz]
        L.cnt_oracle_checksum_words.restype = ctypes.c_uint64
        _lib = L
    return _lib


def _as_u8(n):
    if isinstance(n, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(n), dtype=np.uint8)
    a = np.ascontiguousarray(n, dtype=np.uint8)
    return a


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a.size else ctypes.c_void_p(0)


def _check(rc):
    if rc == 1:
        raise ValueError(ELEN_MESSAGE)
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)


def _encode(fname, n, five=False):
    n = _as_u8(n)
    L = lib()
    words = (L.cnt_oracle_words2_for if five else L.cnt_oracle_words_for)(n.size)
    out = np.empty(words, dtype=np.uint64)
    _check(getattr(L, fname)(_ptr(n), n.size, _ptr(out), words))
    return out


def _aligned_u8(nbytes, align=32):
    raw = np.empty(nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off : off + nbytes]


def _decode(fname, bits, length, cap_bytes):
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    out = _aligned_u8(max(cap_bytes, 1))
    _check(getattr(lib(), fname)(_ptr(bits), bits.size, length, _ptr(out)))
    return out[:length].copy()


# ---- the parity oracle (scalar) -------------------------------------------------
def n_to_bits_lut(n):
    return _encode("cnt_oracle_n_to_bits_lut", n)


def bits_to_n_lut(bits, length):
    return _decode("cnt_oracle_bits_to_n_lut", bits, length, length)


def n_to_bits_bitextract(n):
    return _encode("cnt_oracle_n_to_bits_bitextract", n)


def n_to_bits2_lut(n):
    return _encode("cnt_oracle_n_to_bits2_lut", n, five=True)


def bits_to_n2_lut(bits, length):
    return _decode("cnt_oracle_bits_to_n2_lut", bits, length, length)


# ---- SIMD ports -------------------------------------------------------------------
def port_cpu_ok():
    return bool(lib().cnt_port_cpu_ok())


def n_to_bits_pext(n):
    return _encode("cnt_port_n_to_bits_pext", n)


def n_to_bits_shift(n):
    return _encode("cnt_port_n_to_bits_shift", n)


def n_to_bits_movemask(n):
    return _encode("cnt_port_n_to_bits_movemask", n)


def n_to_bits_mul(n):
    return _encode("cnt_port_n_to_bits_mul", n)


def n_to_bits2_pext(n):
    return _encode("cnt_port_n_to_bits2_pext", n, five=True)


def bits_to_n_shuffle(bits, length):
    return _decode("cnt_port_bits_to_n_shuffle", bits, length, len(bits) * 32)


def bits_to_n_pdep(bits, length):
    return _decode("cnt_port_bits_to_n_pdep", bits, length, len(bits) * 32)


def bits_to_n_clmul(bits, length):
    return _decode("cnt_port_bits_to_n_clmul", bits, length, len(bits) * 32)


def bits_to_n2_pdep(bits, length):
    return _decode("cnt_port_bits_to_n2_pdep", bits, length, len(bits) * 27 + 5)


# ---- packed-domain operations (not in the reference; parity unpinned) --------------------
def hamming(a, b, length):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return int(lib().cnt_oracle_hamming(_ptr(a), _ptr(b), length))


def complement(bits, length):
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    out = np.empty((length + 31) // 32, dtype=np.uint64)
    lib().cnt_oracle_complement(_ptr(bits), length, _ptr(out))
    return out


def reverse_complement(bits, length):
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    out = np.empty((length + 31) // 32, dtype=np.uint64)
    lib().cnt_oracle_reverse_complement(_ptr(bits), length, _ptr(out))
    return out


def validate(n, allow_n=False):
    n = _as_u8(n)
    return int(lib().cnt_oracle_validate(_ptr(n), n.size, 1 if allow_n else 0))


# ---- generator + checksum ---------------------------------------------------------
def fill_random_acgt(n_len, seed, first_nt=0):
    out = np.empty(n_len, dtype=np.uint8)
    lib().cnt_oracle_fill_random_acgt(_ptr(out), first_nt, n_len, seed & 0xFFFFFFFFFFFFFFFF)
    return out


def fill_random_acgtn(n_len, seed, first_nt=0):
    out = np.empty(n_len, dtype=np.uint8)
    lib().cnt_oracle_fill_random_acgtn(_ptr(out), first_nt, n_len, seed & 0xFFFFFFFFFFFFFFFF)
    return out


def checksum_words(words, first_word=0):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    return int(lib().cnt_oracle_checksum_words(_ptr(w), first_word, w.size))


# ---- numpy restatement (independent third opinion; small inputs) ------------------
_NP_BYTE_LUT = np.zeros(256, dtype=np.uint64)
for _c, _v in ((b"aA", 0), (b"cC", 1), (b"tTuU", 2), (b"gG", 3)):  # n_to_bits.rs:8-21
    for _b in _c:
        _NP_BYTE_LUT[_b] = _v
_NP_BITS_LUT = np.frombuffer(b"ACTG", dtype=np.uint8)  # n_to_bits.rs:23-30


def np_n_to_bits_lut(n):
    """n_to_bits.rs:34-47 in numpy: OR of code << 2*(i&31) into word i>>5."""
    n = _as_u8(n)
    words = (n.size + 31) // 32
    codes = np.zeros(words * 32, dtype=np.uint64)
    codes[: n.size] = _NP_BYTE_LUT[n]
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    return np.bitwise_or.reduce(codes.reshape(words, 32) << shifts, axis=1) if words else np.empty(0, np.uint64)


def np_bits_to_n_lut(bits, length):
    """n_to_bits.rs:51-69 in numpy."""
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    if length > bits.size * 32:
        raise ValueError(ELEN_MESSAGE)
    i = np.arange(length, dtype=np.uint64)
    codes = (bits[(i >> np.uint64(5)).astype(np.int64)] >> ((i & np.uint64(31)) << np.uint64(1))) & np.uint64(3)
    return _NP_BITS_LUT[codes.astype(np.int64)]
