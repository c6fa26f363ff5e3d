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
from ._lib import CNT_STRICT_LUT, CNT_TAIL_LUT, check, lib


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


def encode_flags(strict_lut=False, tail_lut=False):
    """The C ABI's encode flags (include/cute_nt.h).  Which reference function each setting reproduces
    on ARBITRARY bytes (on the alphabet ACGTUacgtu all of them agree):
        default              (byte>>1)&3 everywhere
        tail_lut=True        n_to_bits_{pext,shift,movemask,mul} exactly: bit extraction on whole
                             32-nt blocks, BYTE_LUT on the final partial word (n_to_bits.rs:109-111)
        strict_lut=True      n_to_bits_lut exactly: BYTE_LUT everywhere (n_to_bits.rs:8-21,34-47)"""
    return (CNT_STRICT_LUT if strict_lut else 0) | (CNT_TAIL_LUT if tail_lut else 0)


# ---- host tier ------------------------------------------------------------------------
def n_to_bits_hip(n, strict_lut=False, tail_lut=False):
    """Encode {A,T/U,C,G} -> {00,10,01,11}, 32 nt per u64, LSB first (n_to_bits.rs:34-47).

    Returns ceil(len/32) words; the unused high bits of the last word are zero.
    strict_lut=True gives n_to_bits_lut's table semantics on bytes outside the alphabet
    (they encode as 0); the default is the SIMD variants' (byte>>1)&3; tail_lut=True is the SIMD
    variants to the letter (their ragged end goes through the table), see encode_flags."""
    n = _u8(n)
    out = np.empty(lib().cnt_words_for(n.size), dtype=np.uint64)
    check(lib().cnt_n_to_bits_ex(_p(n), n.size, _p(out), out.size, encode_flags(strict_lut, tail_lut)))
    return out


def n_to_bits_hip_checked(n, strict_lut=False, tail_lut=False):
    """n_to_bits_hip and, from the same pass over the data, the number of bytes outside ACGTUacgtu -- what the reference's
    BYTE_LUT turns into code 0 without a word (n_to_bits.rs:8-21,42; README.md:23 points at a separate check).  Returns
    (words, invalid): the words are n_to_bits_hip's whatever `invalid` says."""
    n = _u8(n)
    out = np.empty(lib().cnt_words_for(n.size), dtype=np.uint64)
    bad = ctypes.c_uint64(0)
    check(lib().cnt_n_to_bits_checked(_p(n), n.size, _p(out), out.size, encode_flags(strict_lut, tail_lut), ctypes.byref(bad)))
    return out, bad.value


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


class _PinnedBlock:
    """owner of one cnt_host_alloc allocation; freed when the last array built on it is gone"""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        check(lib().cnt_host_alloc(ctypes.byref(self.ptr), max(1, nbytes)))

    def __del__(self):
        if getattr(self, "ptr", None) and self.ptr.value:
            try:
                lib().cnt_host_free(self.ptr)
            except Exception:  # interpreter shutdown: the library (or ctypes itself) may already be gone; the OS reclaims the pages
                pass
            self.ptr = ctypes.c_void_p()


def pinned_empty(shape, dtype=np.uint8):
    """np.empty in PINNED host memory (cnt_host_alloc): a host-slice call -- n_to_bits_hip_into / bits_to_n_hip_into and the
    returning forms' inputs -- whose input and / or output is such an array skips the staging copy on that side, the copy
    engines use the array in place (include/cute_nt.h "pinned caller memory").  For buffers that live as long as the pipeline:
    pinned memory cannot be swapped.  A torch tensor made with pin_memory=True (`.numpy()`) is the same thing."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    block = _PinnedBlock(count * dtype.itemsize)
    raw = (ctypes.c_uint8 * (count * dtype.itemsize)).from_address(block.ptr.value)
    raw._cnt_block = block  # the view chain keeps `raw` alive, `raw` keeps the allocation
    return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)


def is_pinned(a):
    """would the host tier use this array in place? (cnt_host_is_pinned)"""
    a = np.asarray(a)
    return bool(a.flags.c_contiguous and a.nbytes and lib().cnt_host_is_pinned(_p(a), a.nbytes))


class host_registered:
    """`with host_registered(a):` pins an EXISTING contiguous array in place for the block (cnt_host_register / _unregister:
    tens of milliseconds per GiB, once) -- for a buffer the caller cannot allocate through pinned_empty"""

    def __init__(self, a):
        if not isinstance(a, np.ndarray) or not a.flags.c_contiguous or not a.nbytes:
            raise ValueError("host_registered: a non-empty contiguous numpy array")
        self.a = a

    def __enter__(self):
        check(lib().cnt_host_register(_p(self.a), self.a.nbytes))
        return self.a

    def __exit__(self, *exc):
        check(lib().cnt_host_unregister(_p(self.a)))
        return False


def _own_out(out, dtype, need, what):
    if not isinstance(out, np.ndarray) or out.dtype != dtype or not out.flags.c_contiguous or not out.flags.writeable or out.size < need:
        raise ValueError("%s: out must be a writable contiguous %s array with >= %d elements" % (what, np.dtype(dtype).name, need))


def n_to_bits_hip_into(n, out, strict_lut=False, tail_lut=False):
    """n_to_bits_hip into an array the CALLER owns (uint64, >= ceil(len/32) elements): the `_into` form of the Rust / C++
    mirrors.  A loop that times like the reference's harness (result allocated and dropped inside the timed call,
    benches/bench_n_to_bits.rs:6-7) pays neither fresh-page faults nor munmap this way.  Returns the view out[:words]."""
    n = _u8(n)
    words = lib().cnt_words_for(n.size)
    _own_out(out, np.uint64, words, "n_to_bits_hip_into")
    check(lib().cnt_n_to_bits_ex(_p(n), n.size, _p(out), out.size, encode_flags(strict_lut, tail_lut)))
    return out[:words]


def bits_to_n_hip_into(bits, length, out):
    """bits_to_n_hip into a caller-owned uint8 array (>= length elements); returns the view out[:length]."""
    bits = _u64(bits)
    if length > bits.size * 32:
        check(_lib.CNT_ELEN)
    _own_out(out, np.uint8, length, "bits_to_n_hip_into")
    check(lib().cnt_bits_to_n(_p(bits), bits.size, length, _p(out)))
    return out[:length]


def n_to_bits_hip_sharded(n, ndev=0, strict_lut=False, tail_lut=False):
    """n_to_bits_hip with the buffer cut into contiguous chunks over `ndev` GPUs (0 = all)."""
    n = _u8(n)
    out = np.empty(lib().cnt_words_for(n.size), dtype=np.uint64)
    check(lib().cnt_n_to_bits_sharded_ex(_p(n), n.size, _p(out), out.size, ndev, encode_flags(strict_lut, tail_lut)))
    return out


def bits_to_n_hip_sharded(bits, length, ndev=0):
    bits = _u64(bits)
    if length > bits.size * 32:
        check(_lib.CNT_ELEN)
    out = np.empty(length, dtype=np.uint8)
    check(lib().cnt_bits_to_n_sharded(_p(bits), bits.size, length, _p(out), ndev))
    return out


# ---- device tier (torch tensors; torch is plumbing for device memory + streams) -----------
def _dev_guard(t):
    import torch

    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("device tier needs a contiguous CUDA tensor")
    return torch


def _enqueue(t, fn, *args):
    """Call a device-tier entry point for tensor `t`: on t's device, on torch's current stream OF THAT
    DEVICE (appended as the last argument), and with the calling thread's HIP device left exactly as it
    was found -- a codec call on a cuda:1 tensor must not move the process to cuda:1."""
    import torch

    L = lib()
    cur = ctypes.c_int(-1)
    check(L.cnt_get_device(ctypes.byref(cur)))
    stream = ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    if cur.value == t.device.index:
        check(fn(*args, stream))
        return
    check(L.cnt_set_device(t.device.index))
    try:
        rc = fn(*args, stream)
    finally:
        L.cnt_set_device(cur.value)
    check(rc)


def _out_words(torch, out, words, like):
    """a caller-supplied word buffer: int64, contiguous, on the input's device, >= `words` elements -- the C ABI
    counts capacity in WORDS, so anything narrower than 8 bytes per element would be written out of bounds"""
    if out is None:
        return torch.empty(words, dtype=torch.int64, device=like.device)
    if out.dtype != torch.int64 or not out.is_cuda or not out.is_contiguous() or out.device != like.device or out.numel() < words:
        raise ValueError("out must be a contiguous int64 CUDA tensor on the input's device with >= %d elements" % words)
    return out


def _out_bytes(torch, out, length, like):
    """a caller-supplied ASCII buffer: uint8, contiguous, on the input's device, >= `length` elements (the decode
    entry points of the C ABI take no output capacity: this check is the only one there is)"""
    if out is None:
        return torch.empty(length, dtype=torch.uint8, device=like.device)
    if out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous() or out.device != like.device or out.numel() < length:
        raise ValueError("out must be a contiguous uint8 CUDA tensor on the input's device with >= %d elements" % length)
    return out


def words_for(n_len):
    return lib().cnt_words_for(n_len)


def n_to_bits_dev(n, out=None, strict_lut=False, tail_lut=False):
    """Device-resident encode: uint8 CUDA tensor [N] -> int64 CUDA tensor [ceil(N/32)]
    (bit pattern of the u64 words).  Enqueues on torch's current stream and returns."""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words_for(n.numel())
    out = _out_words(torch, out, words, n)
    _enqueue(n, lib().cnt_n_to_bits_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out.data_ptr()),
             out.numel(), encode_flags(strict_lut, tail_lut))
    return out[:words]


def _counter(torch, acc, like, slots=1):
    """the device u64 a counting entry point ADDS to: a fresh zeroed scalar, or the caller's (which the CALLER zeroes, so that
    several calls can accumulate into one counter without a kernel in between); `slots` of them for the spread form"""
    if acc is None:
        return torch.zeros(slots, dtype=torch.int64, device=like.device)
    if acc.dtype != torch.int64 or not acc.is_cuda or acc.device != like.device or acc.numel() < slots or not acc.is_contiguous():
        raise ValueError("acc must be a contiguous int64 CUDA tensor on the input's device with >= %d elements" % slots)
    return acc


def _checked_flags(strict_lut, tail_lut, spread):
    return encode_flags(strict_lut, tail_lut) | (_lib.CNT_SPREAD_COUNT if spread else 0)


def n_to_bits_checked_dev(n, out=None, acc=None, strict_lut=False, tail_lut=False, spread=False):
    """Device-resident encode + validity count in ONE pass (cnt_n_to_bits_checked_dev): returns (words, acc) where acc is a
    device int64 scalar the call ADDED the number of bytes outside ACGTUacgtu to (a fresh zeroed one unless the caller passes
    its own).  1.25 B/nt of HBM traffic, where validate_dev + n_to_bits_dev is 2.25.  acc.item() synchronises.
    spread=True (CNT_SPREAD_COUNT): acc has CNT_COUNT_SLOTS = 2048 elements and the count is acc.sum() -- for data in which most
    tiles hold strays (one atomic per dirty tile on ONE counter is 32 x the clean time at 2^34 nt; over 2048 counters (128 cache lines) it is noise)."""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words_for(n.numel())
    out = _out_words(torch, out, words, n)
    acc = _counter(torch, acc, n, _lib.CNT_COUNT_SLOTS if spread else 1)
    _enqueue(n, lib().cnt_n_to_bits_checked_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out.data_ptr()),
             out.numel(), _checked_flags(strict_lut, tail_lut, spread), ctypes.c_void_p(acc.data_ptr()))
    return out[:words], acc


def round_trip_checked_dev(n, out_bits=None, out_n=None, acc=None, strict_lut=False, tail_lut=False, spread=False):
    """round_trip_dev + the validity count of n_to_bits_checked_dev, still one pass: (words, canonical ASCII, acc)"""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words_for(n.numel())
    out_bits = _out_words(torch, out_bits, words, n)
    out_n = _out_bytes(torch, out_n, n.numel(), n)
    acc = _counter(torch, acc, n, _lib.CNT_COUNT_SLOTS if spread else 1)
    _enqueue(n, lib().cnt_round_trip_checked_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out_bits.data_ptr()),
             out_bits.numel(), ctypes.c_void_p(out_n.data_ptr()), _checked_flags(strict_lut, tail_lut, spread), ctypes.c_void_p(acc.data_ptr()))
    return out_bits[:words], out_n[: n.numel()], acc


def round_trip_dev(n, out_bits=None, out_n=None, strict_lut=False, tail_lut=False):
    """Fused device-resident encode + decode: returns (packed words, canonical ASCII) of `n` in one
    pass over it (cnt_round_trip_dev)."""
    torch = _dev_guard(n)
    if n.dtype != torch.uint8:
        raise TypeError("nucleotides must be a uint8 tensor")
    words = lib().cnt_words_for(n.numel())
    out_bits = _out_words(torch, out_bits, words, n)
    out_n = _out_bytes(torch, out_n, n.numel(), n)
    _enqueue(n, lib().cnt_round_trip_dev, ctypes.c_void_p(n.data_ptr()), n.numel(), ctypes.c_void_p(out_bits.data_ptr()),
             out_bits.numel(), ctypes.c_void_p(out_n.data_ptr()), encode_flags(strict_lut, tail_lut))
    return out_bits[:words], out_n[: n.numel()]


def bits_to_n_dev(bits, length, out=None):
    """Device-resident decode: int64 CUDA tensor of packed words -> uint8 CUDA tensor [length]."""
    torch = _dev_guard(bits)
    if bits.dtype != torch.int64:
        raise TypeError("packed words must be an int64 tensor (u64 bit pattern)")
    if length > bits.numel() * 32:
        check(_lib.CNT_ELEN)
    out = _out_bytes(torch, out, length, bits)
    _enqueue(bits, lib().cnt_bits_to_n_dev, ctypes.c_void_p(bits.data_ptr()), bits.numel(), length, ctypes.c_void_p(out.data_ptr()), 0)
    return out[:length]
