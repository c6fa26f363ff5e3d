"""Device-side helpers for benches and large-size verification (generator, checksum,
mismatch count) over the C ABI's utility entry points.  torch is only the allocator."""
import ctypes

from ._lib import check, lib
from .n_to_bits import _dev_guard, _enqueue


def fill_random_acgt(out, seed, first_nt=0):
    """Fill a uint8 CUDA tensor with the counter-based uniform ACGT stream (same stream the
    oracle generates on the host, so any chunk can be regenerated there)."""
    _dev_guard(out)
    _enqueue(out, lib().cnt_fill_random_acgt_dev, ctypes.c_void_p(out.data_ptr()), first_nt, out.numel(), seed & 0xFFFFFFFFFFFFFFFF)
    return out


def fill_random_acgtn(out, seed, first_nt=0):
    _dev_guard(out)
    _enqueue(out, lib().cnt_fill_random_acgtn_dev, ctypes.c_void_p(out.data_ptr()), first_nt, out.numel(), seed & 0xFFFFFFFFFFFFFFFF)
    return out


def checksum_words(words, first_word=0):
    """Position-salted 64-bit checksum of an int64 CUDA tensor of packed words (syncs)."""
    torch = _dev_guard(words)
    acc = torch.zeros(1, dtype=torch.int64, device=words.device)
    _enqueue(words, lib().cnt_checksum_words_dev, ctypes.c_void_p(words.data_ptr()), first_word, words.numel(), ctypes.c_void_p(acc.data_ptr()))
    return int(acc.item()) & 0xFFFFFFFFFFFFFFFF


def count_mismatch(a, b):
    """Number of differing bytes between two equally sized CUDA tensors (syncs)."""
    torch = _dev_guard(a)
    nbytes = a.numel() * a.element_size()
    if nbytes != b.numel() * b.element_size():
        raise ValueError("size mismatch")
    acc = torch.zeros(1, dtype=torch.int64, device=a.device)
    _enqueue(a, lib().cnt_count_mismatch_dev, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), nbytes, ctypes.c_void_p(acc.data_ptr()))
    return int(acc.item())


def set_tuning(key, value):
    """LAB BUILD ONLY (cute_nucleotides_amd._lib.use_lab_build()): the product library has nothing to select and answers
    CNT_EINVAL for every key."""
    from . import _lib

    rc = lib().cnt_set_tuning(key.encode(), int(value))
    if rc == _lib.CNT_EINVAL and not _lib.is_lab_build():
        raise _lib.CuteNtError(rc, "cnt_set_tuning(%r): the product library has no run-time kernel selection; call "
                                   "cute_nucleotides_amd._lib.use_lab_build() first (bench/libcute_nt_hip_lab.so)" % key)
    check(rc)


def get_tuning(key):
    v = ctypes.c_int(0)
    check(lib().cnt_get_tuning(key.encode(), ctypes.byref(v)))
    return v.value


def variants(key):
    """[(index, description)] of the selectable kernel variants for key 'encode' | 'decode'."""
    n = get_tuning(key + "_variants")
    return [(i, lib().cnt_tuning_name(key.encode(), i).decode()) for i in range(n)]


def device_identity(index):
    """{device_index, pci_bus_id, numa_node, visible_devices} of a visible device, from the C ABI
    (cnt_device_pci_bus_id / cnt_device_numa_node / cnt_device_count)."""
    L = lib()
    buf = ctypes.create_string_buffer(64)
    check(L.cnt_device_pci_bus_id(index, buf, 64))
    node, count = ctypes.c_int(-1), ctypes.c_int(0)
    check(L.cnt_device_numa_node(index, ctypes.byref(node)))
    check(L.cnt_device_count(ctypes.byref(count)))
    return {"device_index": index, "pci_bus_id": buf.value.decode(), "numa_node": node.value, "visible_devices": count.value}
