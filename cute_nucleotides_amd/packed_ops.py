"""Operations on 2-bit packed nucleotides without decoding: Hamming distance,
complement, reverse complement, and alphabet validation of ASCII buffers.  The reference does
not implement these (its README.md:20-25,45 only points at them); semantics are defined in
include/cute_nt.h and restated by the oracle.  Host tier: numpy; device tier: torch tensors on
torch's current stream."""
import ctypes

import numpy as np

from ._lib import check, lib
from .n_to_bits import _counter, _dev_guard, _enqueue, _out_words, _p, _u8, _u64

CNT_ALLOW_N = 0x2


def _need(bits, length):
    if length > bits.size * 32:
        raise ValueError("The length is greater than the number of nucleotides!")


# ---- host tier --------------------------------------------------------------------------
def hamming_hip(a, b, length):
    a, b = _u64(a), _u64(b)
    _need(a, length)
    _need(b, length)
    out = ctypes.c_uint64(0)
    check(lib().cnt_hamming(_p(a), _p(b), length, ctypes.byref(out)))
    return out.value


def complement_hip(bits, length):
    bits = _u64(bits)
    _need(bits, length)
    out = np.empty(lib().cnt_words_for(length), dtype=np.uint64)
    check(lib().cnt_complement(_p(bits), length, _p(out)))
    return out


def reverse_complement_hip(bits, length):
    bits = _u64(bits)
    _need(bits, length)
    out = np.empty(lib().cnt_words_for(length), dtype=np.uint64)
    check(lib().cnt_reverse_complement(_p(bits), length, _p(out)))
    return out


def validate_hip(n, allow_n=False):
    """Number of bytes that are not nucleotides (0 = the buffer is a valid sequence)."""
    n = _u8(n)
    out = ctypes.c_uint64(0)
    check(lib().cnt_validate(_p(n), n.size, CNT_ALLOW_N if allow_n else 0, ctypes.byref(out)))
    return out.value


# ---- device tier --------------------------------------------------------------------------
def hamming_dev(a, b, length, acc=None):
    torch = _dev_guard(a)
    _dev_guard(b)
    if a.dtype != torch.int64 or b.dtype != torch.int64:
        raise TypeError("packed words must be int64 tensors")
    if a.device != b.device:
        raise ValueError("both sequences must live on the same device")
    if length > min(a.numel(), b.numel()) * 32:
        raise ValueError("The length is greater than the number of nucleotides!")
    acc = _counter(torch, acc, a)
    _enqueue(a, lib().cnt_hamming_dev, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), length, ctypes.c_void_p(acc.data_ptr()))
    return acc  # device scalar: .item() syncs


def _unary_dev(fn, bits, length, out):
    torch = _dev_guard(bits)
    if bits.dtype != torch.int64:
        raise TypeError("packed words must be an int64 tensor")
    if length > bits.numel() * 32:
        raise ValueError("The length is greater than the number of nucleotides!")
    words = lib().cnt_words_for(length)
    out = _out_words(torch, out, words, bits)
    _enqueue(bits, fn, ctypes.c_void_p(bits.data_ptr()), length, ctypes.c_void_p(out.data_ptr()))
    return out[:words]


def complement_dev(bits, length, out=None):
    return _unary_dev(lib().cnt_complement_dev, bits, length, out)


def reverse_complement_dev(bits, length, out=None):
    return _unary_dev(lib().cnt_reverse_complement_dev, bits, length, out)


def validate_dev(n, allow_n=False, acc=None):
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    acc = _counter(torch, acc, n)
    _enqueue(n, lib().cnt_validate_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), CNT_ALLOW_N if allow_n else 0, ctypes.c_void_p(acc.data_ptr()))
    return acc
