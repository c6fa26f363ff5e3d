"""Host-side mirror of the reference's `cute_nucleotides::n_to_bits` module (src/n_to_bits.rs).

Same names, argument meaning and error behaviour as the Rust functions, with `_hip`
appended (the name the Rust binding in INTEGRATION.md uses):

    n_to_bits_hip(n)            <-> n_to_bits_{lut,pext,shift,movemask,mul}(n: &[u8]) -> Vec<u64>
    bits_to_n_hip(bits, len)    <-> bits_to_n_{lut,shuffle,pdep,clmul}(bits: &[u64], len) -> Vec<u8>

Host tier: numpy in / numpy out through the C ABI's host-pointer entry points (H2D, kernel,
D2H inside).  Device tier (`*_dev`): torch CUDA tensors in / out, enqueue-only on torch's
current stream -- this is what bench.py times.  Every call goes to libcute_nt_hip.so; there
is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import CNT_STRICT_LUT, check, lib


def _u8(n):
    if isinstance(n, (bytes, bytearray, memoryview)):
        return np.frombuffer(n, dtype=np.uint8)
    a = np.asarray(n)
    if a.dtype != np.uint8:
        raise TypeError("nucleotides must be bytes or a uint8 array")
    return np.ascontiguousarray(a)


def _u64(bits):
    a = np.asarray(bits)
    if a.dtype != np.uint64:
        raise TypeError("packed words must be a uint64 array")
    return np.ascontiguousarray(a)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data if a.size else 0)


# ---- host tier ------------------------------------------------------------------------
def n_to_bits_hip(n, strict_lut=False):
    """Encode {A,T/U,C,G} -> {00,10,01,11}, 32 nt per u64, LSB first (n_to_bits.rs:34-47).

    Returns ceil(len/32) words; the unused high bits of the last word are zero.
    strict_lut=True gives n_to_bits_lut's table semantics on bytes outside the alphabet
    (they encode as 0); the default is the SIMD variants' (byte>>1)&3."""
    n = _u8(n)
    out = np.empty(lib().cnt_words_for(n.size), dtype=np.uint64)
    check(lib().cnt_n_to_bits_ex(_p(n), n.size, _p(out), out.size, CNT_STRICT_LUT if strict_lut else 0))
    return out


def bits_to_n_hip(bits, length):
    """Decode `length` nucleotides from packed words (n_to_bits.rs:51-69).

    Raises ValueError("The length is greater than the number of nucleotides!") when
    length > 32*len(bits), the reference's panic (n_to_bits.rs:52-54)."""
    bits = _u64(bits)
    if length > bits.size * 32:
        check(_lib.CNT_ELEN)
    out = np.empty(length, dtype=np.uint8)
    check(lib().cnt_bits_to_n(_p(bits), bits.size, length, _p(out)))
    return out


def n_to_bits_hip_sharded(n, ndev=0):
    """n_to_bits_hip with the buffer cut into contiguous chunks over `ndev` GPUs (0 = all)."""
    n = _u8(n)
    out = np.empty(lib().cnt_words_for(n.size), dtype=np.uint64)
    check(lib().cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, ndev))
    return out


def bits_to_n_hip_sharded(bits, length, ndev=0):
    bits = _u64(bits)
    if length > bits.size * 32:
        check(_lib.CNT_ELEN)
    out = np.empty(length, dtype=np.uint8)
    check(lib().cnt_bits_to_n_sharded(_p(bits), bits.size, length, _p(out), ndev))
    return out


# ---- device tier (torch tensors; torch is plumbing for device memory + streams) -----------
def _stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_guard(t):
    import torch

    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("device tier needs a contiguous CUDA tensor")
    cur = ctypes.c_int(-1)
    check(lib().cnt_get_device(ctypes.byref(cur)))
    if cur.value != t.device.index:
        check(lib().cnt_set_device(t.device.index))
    return torch


def words_for(n_len):
    return lib().cnt_words_for(n_len)


def n_to_bits_dev(n, out=None, strict_lut=False):
    """Device-resident encode: uint8 CUDA tensor [N] -> int64 CUDA tensor [ceil(N/32)]
    (bit pattern of the u64 words).  Enqueues on torch's current stream and returns."""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words_for(n.numel())
    if out is None:
        out = torch.empty(words, dtype=torch.int64, device=n.device)
    elif out.dtype != torch.int64 or not out.is_cuda or not out.is_contiguous() or out.device != n.device:
        raise ValueError("out must be a contiguous int64 CUDA tensor on the input's device")
    check(lib().cnt_n_to_bits_dev(ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out.data_ptr()),
                                  out.numel(), CNT_STRICT_LUT if strict_lut else 0, _stream_ptr()))
    return out[:words]


def round_trip_dev(n, out_bits=None, out_n=None, strict_lut=False):
    """Fused device-resident encode + decode: returns (packed words, canonical ASCII) of `n` in one
    pass over it (cnt_round_trip_dev)."""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words_for(n.numel())
    if out_bits is None:
        out_bits = torch.empty(words, dtype=torch.int64, device=n.device)
    if out_n is None:
        out_n = torch.empty(n.numel(), dtype=torch.uint8, device=n.device)
    if out_bits.dtype != torch.int64 or out_n.dtype != torch.uint8 or out_n.numel() < n.numel() or not (
            out_bits.is_cuda and out_n.is_cuda and out_bits.is_contiguous() and out_n.is_contiguous()):
        raise ValueError("out_bits: contiguous int64 CUDA tensor; out_n: contiguous uint8 CUDA tensor with >= len(n) elements")
    check(lib().cnt_round_trip_dev(ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out_bits.data_ptr()), out_bits.numel(),
                                   ctypes.c_void_p(out_n.data_ptr()), CNT_STRICT_LUT if strict_lut else 0, _stream_ptr()))
    return out_bits[:words], out_n[: n.numel()]


def bits_to_n_dev(bits, length, out=None):
    """Device-resident decode: int64 CUDA tensor of packed words -> uint8 CUDA tensor [length]."""
    torch = _dev_guard(bits)
    if bits.dtype != torch.int64:
        raise TypeError("packed words must be an int64 tensor (u64 bit pattern)")
    if length > bits.numel() * 32:
        check(_lib.CNT_ELEN)
    if out is None:
        out = torch.empty(length, dtype=torch.uint8, device=bits.device)
    elif out.dtype != torch.uint8 or out.numel() < length or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must be a contiguous uint8 CUDA tensor with >= length elements")
    check(lib().cnt_bits_to_n_dev(ctypes.c_void_p(bits.data_ptr()), bits.numel(), length,
                                  ctypes.c_void_p(out.data_ptr()), 0, _stream_ptr()))
    return out[:length]
