"""Host-side mirror of the reference's `cute_nucleotides::n_to_bits2` module (src/n_to_bits2.rs):
the 5-letter {A,C,G,T/U,N} codec, 3 nt -> 7 bits, 27 nt per u64.

    n_to_bits2_hip(n)          <-> n_to_bits2_{lut,pext}(n: &[u8]) -> Vec<u64>          (:37, :118)
    bits_to_n2_hip(bits, len)  <-> bits_to_n2_{lut,pdep}(bits: &[u64], len) -> Vec<u8>  (:78, :196)
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib
from .n_to_bits import _checked_flags, _counter, _dev_guard, _enqueue, _out_bytes, _out_words, _own_out, _p, _u8, _u64, encode_flags


def n_to_bits2_hip(n, strict_lut=False, tail_lut=False):
    """default: the low-3-bit table of n_to_bits2_pext (n_to_bits2.rs:127-136) on every byte;
    tail_lut=True: n_to_bits2_pext to the letter -- that table up to word (len-5)/27, BYTE_LUT from
    there on (n_to_bits2.rs:120,179-185); strict_lut=True: n_to_bits2_lut (BYTE_LUT everywhere)."""
    n = _u8(n)
    out = np.empty(lib().cnt_words2_for(n.size), dtype=np.uint64)
    check(lib().cnt_n_to_bits2_ex(_p(n), n.size, _p(out), out.size, encode_flags(strict_lut, tail_lut)))
    return out


def n_to_bits2_hip_checked(n, strict_lut=False, tail_lut=False):
    """n_to_bits2_hip and, from the same pass, the number of bytes outside ACGTUNacgtun (BYTE_LUT's alphabet, n_to_bits2.rs:8-23):
    (words, invalid)"""
    n = _u8(n)
    out = np.empty(lib().cnt_words2_for(n.size), dtype=np.uint64)
    bad = ctypes.c_uint64(0)
    check(lib().cnt_n_to_bits2_checked(_p(n), n.size, _p(out), out.size, encode_flags(strict_lut, tail_lut), ctypes.byref(bad)))
    return out, bad.value


def bits_to_n2_hip(bits, length):
    bits = _u64(bits)
    if length > bits.size * 27:
        check(_lib.CNT_ELEN)
    out = np.empty(length, dtype=np.uint8)
    check(lib().cnt_bits_to_n2(_p(bits), bits.size, length, _p(out)))
    return out


def n_to_bits2_hip_into(n, out, strict_lut=False, tail_lut=False):
    """n_to_bits2_hip into a caller-owned uint64 array (>= ceil(len/27) elements); returns the view out[:words]."""
    n = _u8(n)
    words = lib().cnt_words2_for(n.size)
    _own_out(out, np.uint64, words, "n_to_bits2_hip_into")
    check(lib().cnt_n_to_bits2_ex(_p(n), n.size, _p(out), out.size, encode_flags(strict_lut, tail_lut)))
    return out[:words]


def bits_to_n2_hip_into(bits, length, out):
    """bits_to_n2_hip into a caller-owned uint8 array (>= length elements); returns the view out[:length]."""
    bits = _u64(bits)
    if length > bits.size * 27:
        check(_lib.CNT_ELEN)
    _own_out(out, np.uint8, length, "bits_to_n2_hip_into")
    check(lib().cnt_bits_to_n2(_p(bits), bits.size, length, _p(out)))
    return out[:length]


def n_to_bits2_hip_sharded(n, ndev=0, strict_lut=False, tail_lut=False):
    """n_to_bits2_hip with the buffer cut into contiguous chunks over `ndev` GPUs (0 = all)."""
    n = _u8(n)
    out = np.empty(lib().cnt_words2_for(n.size), dtype=np.uint64)
    check(lib().cnt_n_to_bits2_sharded_ex(_p(n), n.size, _p(out), out.size, ndev, encode_flags(strict_lut, tail_lut)))
    return out


def bits_to_n2_hip_sharded(bits, length, ndev=0):
    bits = _u64(bits)
    if length > bits.size * 27:
        check(_lib.CNT_ELEN)
    out = np.empty(length, dtype=np.uint8)
    check(lib().cnt_bits_to_n2_sharded(_p(bits), bits.size, length, _p(out), ndev))
    return out


def words2_for(n_len):
    return lib().cnt_words2_for(n_len)


def n_to_bits2_dev(n, out=None, strict_lut=False, tail_lut=False):
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words2_for(n.numel())
    out = _out_words(torch, out, words, n)
    _enqueue(n, lib().cnt_n_to_bits2_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out.data_ptr()),
             out.numel(), encode_flags(strict_lut, tail_lut))
    return out[:words]


def n_to_bits2_checked_dev(n, out=None, acc=None, strict_lut=False, tail_lut=False, spread=False):
    """n_to_bits2_dev + the number of bytes outside ACGTUNacgtun added to the device scalar `acc`, one pass: (words, acc)"""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words2_for(n.numel())
    out = _out_words(torch, out, words, n)
    acc = _counter(torch, acc, n, _lib.CNT_COUNT_SLOTS if spread else 1)
    _enqueue(n, lib().cnt_n_to_bits2_checked_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out.data_ptr()),
             out.numel(), _checked_flags(strict_lut, tail_lut, spread), ctypes.c_void_p(acc.data_ptr()))
    return out[:words], acc


def bits_to_n2_dev(bits, length, out=None):
    torch = _dev_guard(bits)
    if bits.dtype != torch.int64:
        raise TypeError("packed words must be an int64 tensor (u64 bit pattern)")
    if length > bits.numel() * 27:
        check(_lib.CNT_ELEN)
    out = _out_bytes(torch, out, length, bits)
    _enqueue(bits, lib().cnt_bits_to_n2_dev, ctypes.c_void_p(bits.data_ptr()), bits.numel(), length, ctypes.c_void_p(out.data_ptr()), 0)
    return out[:length]
