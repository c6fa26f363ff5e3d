"""ctypes binding of libcute_nt_hip.so -- every symbol include/cute_nt.h declares.

There is NO fallback: if the HIP library is missing or fails to load, importing the codec
functions raises.  Nothing here (or anywhere in this package) touches oracle/.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcute_nt_hip.so")
# The LAB build of the same sources (-DCNT_LAB_VARIANTS: every measured kernel variant + cnt_set_tuning), under bench/ and
# never loaded unless a bench script or a test asks for it with use_lab_build(): the product library has no run-time kernel
# selection (cnt_set_tuning answers CNT_EINVAL there).
LAB_LIB_PATH = os.path.join(os.path.dirname(HERE), "bench", "libcute_nt_hip_lab.so")
# The TEST-HOOKS build (-DCNT_TEST_HOOKS: the product's kernels + the cnt_test_* hooks), under tests/: what the N > 1
# sharded tests load to fold shards onto a 1-GPU box.  The product library exports no hook and holds no switch.
HOOKS_LIB_PATH = os.path.join(os.path.dirname(HERE), "tests", "libcute_nt_hip_hooks.so")

CNT_OK, CNT_EINVAL, CNT_ECAP, CNT_ELEN, CNT_ENODEV, CNT_ERANGE = 0, 1, 2, 3, 4, 5
CNT_STRICT_LUT = 0x1
CNT_TAIL_LUT = 0x4
CNT_SPREAD_COUNT = 0x8  # the *_checked device entry points: the counter is CNT_COUNT_SLOTS u64, the count their sum
CNT_COUNT_SLOTS = 2048
CNT_QUEUE_TIMED = 0x1

_vp, _sz, _u64, _int, _uint = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint

# name -> (restype, argtypes): the whole ABI, also used by tests/test_abi.py
SIGNATURES = {
    "cnt_strerror": (ctypes.c_char_p, [_int]),
    "cnt_abi_version": (_int, []),
    "cnt_words_for": (_sz, [_sz]),
    "cnt_words2_for": (_sz, [_sz]),
    "cnt_device_count": (_int, [ctypes.POINTER(_int)]),
    "cnt_set_device": (_int, [_int]),
    "cnt_get_device": (_int, [ctypes.POINTER(_int)]),
    "cnt_device_pci_bus_id": (_int, [_int, ctypes.c_char_p, _sz]),
    "cnt_device_numa_node": (_int, [_int, ctypes.POINTER(_int)]),
    "cnt_shutdown": (_int, []),
    "cnt_host_tier_info": (_int, [ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "cnt_host_alloc": (_int, [ctypes.POINTER(ctypes.c_void_p), _sz]),
    "cnt_host_free": (_int, [_vp]),
    "cnt_host_register": (_int, [_vp, _sz]),
    "cnt_host_unregister": (_int, [_vp]),
    "cnt_host_is_pinned": (_int, [_vp, _sz]),
    "cnt_n_to_bits": (_int, [_vp, _sz, _vp, _sz]),
    "cnt_n_to_bits_ex": (_int, [_vp, _sz, _vp, _sz, _uint]),
    "cnt_n_to_bits_checked": (_int, [_vp, _sz, _vp, _sz, _uint, ctypes.POINTER(_u64)]),
    "cnt_n_to_bits2_checked": (_int, [_vp, _sz, _vp, _sz, _uint, ctypes.POINTER(_u64)]),
    "cnt_bits_to_n": (_int, [_vp, _sz, _sz, _vp]),
    "cnt_n_to_bits2": (_int, [_vp, _sz, _vp, _sz]),
    "cnt_n_to_bits2_ex": (_int, [_vp, _sz, _vp, _sz, _uint]),
    "cnt_bits_to_n2": (_int, [_vp, _sz, _sz, _vp]),
    "cnt_n_to_bits_sharded": (_int, [_vp, _sz, _vp, _sz, _int]),
    "cnt_n_to_bits_sharded_ex": (_int, [_vp, _sz, _vp, _sz, _int, _uint]),
    "cnt_n_to_bits2_sharded_ex": (_int, [_vp, _sz, _vp, _sz, _int, _uint]),
    "cnt_bits_to_n_sharded": (_int, [_vp, _sz, _sz, _vp, _int]),
    "cnt_n_to_bits2_sharded": (_int, [_vp, _sz, _vp, _sz, _int]),
    "cnt_bits_to_n2_sharded": (_int, [_vp, _sz, _sz, _vp, _int]),
    "cnt_shard_range": (_int, [_sz, _int, _int, _int, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "cnt_shard_worker_info": (_int, [_int, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "cnt_n_to_bits_sharded_dev": (_int, [_vp, _vp, _vp, _vp, _int, _uint, _vp]),
    "cnt_bits_to_n_sharded_dev": (_int, [_vp, _vp, _vp, _vp, _int, _uint, _vp]),
    "cnt_n_to_bits2_sharded_dev": (_int, [_vp, _vp, _vp, _vp, _int, _uint, _vp]),
    "cnt_bits_to_n2_sharded_dev": (_int, [_vp, _vp, _vp, _vp, _int, _uint, _vp]),
    "cnt_sharded_dev_open": (_int, [_int, _uint, ctypes.POINTER(_vp)]),
    "cnt_sharded_dev_open_on_streams": (_int, [_int, _vp, _uint, ctypes.POINTER(_vp)]),
    "cnt_sharded_dev_open_on_devices": (_int, [_int, ctypes.POINTER(_int), _uint, ctypes.POINTER(_vp)]),
    "cnt_sharded_dev_device": (_int, [_vp, _int, ctypes.POINTER(_int)]),
    "cnt_sharded_dev_wait_event": (_int, [_vp, _int, _vp]),
    "cnt_sharded_dev_record_event": (_int, [_vp, _int, _vp]),
    "cnt_sharded_dev_close": (_int, [_vp]),
    "cnt_sharded_dev_shards": (_int, [_vp, ctypes.POINTER(_int)]),
    "cnt_n_to_bits_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _uint]),
    "cnt_bits_to_n_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _uint]),
    "cnt_round_trip_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _uint]),
    "cnt_n_to_bits2_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _uint]),
    "cnt_bits_to_n2_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _uint]),
    "cnt_n_to_bits_checked_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _uint, _vp]),
    "cnt_n_to_bits2_checked_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _uint, _vp]),
    "cnt_round_trip_checked_sharded_dev_enqueue": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _uint, _vp]),
    "cnt_sharded_dev_wait": (_int, [_vp, _vp]),
    "cnt_sharded_dev_op_ms": (_int, [_vp, _sz, _vp]),
    "cnt_n_to_bits_dev": (_int, [_vp, _sz, _vp, _sz, _uint, _vp]),
    "cnt_bits_to_n_dev": (_int, [_vp, _sz, _sz, _vp, _uint, _vp]),
    "cnt_n_to_bits2_dev": (_int, [_vp, _sz, _vp, _sz, _uint, _vp]),
    "cnt_round_trip_dev": (_int, [_vp, _sz, _vp, _sz, _vp, _uint, _vp]),
    "cnt_bits_to_n2_dev": (_int, [_vp, _sz, _sz, _vp, _uint, _vp]),
    "cnt_n_to_bits_checked_dev": (_int, [_vp, _sz, _vp, _sz, _uint, _vp, _vp]),
    "cnt_n_to_bits2_checked_dev": (_int, [_vp, _sz, _vp, _sz, _uint, _vp, _vp]),
    "cnt_round_trip_checked_dev": (_int, [_vp, _sz, _vp, _sz, _vp, _uint, _vp, _vp]),
    "cnt_dev_alloc": (_int, [ctypes.POINTER(_vp), _sz]),
    "cnt_dev_free": (_int, [_vp]),
    "cnt_dev_upload": (_int, [_vp, _vp, _sz]),
    "cnt_dev_download": (_int, [_vp, _vp, _sz]),
    "cnt_dev_sync": (_int, [_vp]),
    "cnt_fill_random_acgt_dev": (_int, [_vp, _sz, _sz, _u64, _vp]),
    "cnt_fill_random_acgtn_dev": (_int, [_vp, _sz, _sz, _u64, _vp]),
    "cnt_checksum_words_dev": (_int, [_vp, _sz, _sz, _vp, _vp]),
    "cnt_count_mismatch_dev": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "cnt_hamming_dev": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "cnt_complement_dev": (_int, [_vp, _sz, _vp, _vp]),
    "cnt_reverse_complement_dev": (_int, [_vp, _sz, _vp, _vp]),
    "cnt_validate_dev": (_int, [_vp, _sz, _uint, _vp, _vp]),
    "cnt_hamming": (_int, [_vp, _vp, _sz, ctypes.POINTER(_u64)]),
    "cnt_complement": (_int, [_vp, _sz, _vp]),
    "cnt_reverse_complement": (_int, [_vp, _sz, _vp]),
    "cnt_validate": (_int, [_vp, _sz, _uint, ctypes.POINTER(_u64)]),
    "cnt_set_tuning": (_int, [ctypes.c_char_p, _int]),
    "cnt_get_tuning": (_int, [ctypes.c_char_p, ctypes.POINTER(_int)]),
    "cnt_tuning_name": (ctypes.c_char_p, [ctypes.c_char_p, _int]),
    "cnt_chip_info": (_int, [_int, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "cnt_chip_cache_nt": (_int, [_int, ctypes.POINTER(_u64)]),
    "cnt_check_device_range": (_int, [ctypes.c_void_p, ctypes.c_size_t, _int]),
}
# exported by the hooks build and the lab build ONLY (include/cute_nt.h, under #ifdef CNT_TEST_HOOKS)
TEST_HOOK_SIGNATURES = {
    "cnt_test_alias_devices": (_int, [_int]),
    "cnt_test_advise_output": (_int, [_vp, _sz]),
    "cnt_test_round_trip_plan": (_int, [_u64, _u64, _u64, _u64, _uint, ctypes.POINTER(_u64)]),
    "cnt_test_decode_plan": (_int, [_u64, _u64, _u64, _u64, ctypes.POINTER(_u64)]),
    "cnt_test_pipeline_pieces": (_int, [_u64, _uint, _uint, ctypes.POINTER(_u64), _int]),
    "cnt_test_host_trace": (_int, [ctypes.POINTER(_int), ctypes.POINTER(ctypes.c_double), _int]),
}
CNT_QUEUE_MAX_TIMED_OPS = 4096

_libs = {}          # "product" / "lab" / "hooks" -> loaded CDLL
_active = "product"
_PATHS = {"product": LIB_PATH, "lab": LAB_LIB_PATH, "hooks": HOOKS_LIB_PATH}


class CuteNtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libcute_nt_hip: %s (status %d)" % (message, status))
        self.status = status


def use_build(name):
    """TEST / BENCH SUPPORT: make every wrapper of this package call the named build -- "product" (the default),
    "hooks" (tests/libcute_nt_hip_hooks.so: product kernels + cnt_test_*) or "lab" (bench/libcute_nt_hip_lab.so) -- for this
    process, until switched back.  Returns the previous name.  The builds can be loaded side by side; each keeps its own
    streams and scratch."""
    global _active
    if name not in _PATHS:
        raise ValueError("unknown build %r" % (name,))
    prev, _active = _active, name
    return prev


def active_build():
    return _active


def has_test_hooks():
    return _active in ("lab", "hooks")


def use_lab_build(on=True):
    """BENCH / TEST SUPPORT: make every wrapper of this package call bench/libcute_nt_hip_lab.so (built on demand) instead of
    the product library, for this process, until switched back.  Returns the previous setting.  Both libraries can be
    loaded side by side; each keeps its own streams and scratch."""
    prev = _active == "lab"
    use_build("lab" if on else "product")
    return prev


def is_lab_build():
    return _active == "lab"


def lib():
    """Load the active HIP library (once per build).  Raises if it is not built -- no CPU fallback exists."""
    if _active not in _libs:
        path = _PATHS[_active]
        if not os.path.exists(path):
            # not built yet on this box: compile it (hipcc, gfx950) rather than give up -- this is
            # still the HIP library, never a CPU substitute; without hipcc the import fails below
            try:
                from . import build as _build

                {"lab": _build.build_lab, "hooks": _build.build_hooks, "product": _build.build}[_active]()
            except Exception as exc:  # noqa: BLE001
                raise ImportError("%s is missing and could not be built: %s" % (path, exc)) from exc
        if not os.path.exists(path):
            raise ImportError(
                "%s is missing: build it with `python -m cute_nucleotides_amd.build%s` "
                "(or __graft_entry__.build()); this package has no CPU fallback" % (path, {"lab": " --lab", "hooks": " --hooks", "product": ""}[_active])
            )
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (soname
        # libamdhip64.so.7) and loads it by file name, so it must be in the process BEFORE this
        # library's NEEDED libamdhip64.so.7 is resolved -- then both share that one runtime.
        # Loading in the other order maps two runtimes and the second one sees no device.
        import torch  # noqa: F401  (device memory + streams plumbing; must precede the CDLL)

        L = ctypes.CDLL(path)
        _assert_single_hip_runtime()
        sigs = dict(SIGNATURES, **TEST_HOOK_SIGNATURES) if _active != "product" else SIGNATURES
        for name, (res, args) in sigs.items():
            f = getattr(L, name)  # AttributeError here = ABI mismatch, fail loudly
            f.restype = res
            f.argtypes = args
        if _active == "product":
            for name in TEST_HOOK_SIGNATURES:  # the product exports no test hook
                if hasattr(L, name):
                    raise ImportError("%s exports the test hook %s: it is not a product build" % (path, name))
        if L.cnt_abi_version() != 1:
            raise ImportError("libcute_nt_hip ABI version mismatch")
        _libs[_active] = L
    return _libs[_active]


def _assert_single_hip_runtime():
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    real = {os.path.realpath(p) for p in paths}
    if len(real) > 1:
        raise ImportError("two HIP runtimes are mapped in this process (%s): import torch before "
                          "cute_nucleotides_amd" % ", ".join(sorted(real)))


def strerror(status):
    return lib().cnt_strerror(status).decode()


def check(status):
    """0 -> None; CNT_ELEN -> ValueError with the reference's panic text; else CuteNtError."""
    if status == CNT_OK:
        return
    msg = strerror(status)
    if status == CNT_ELEN:
        raise ValueError(msg)  # "The length is greater than the number of nucleotides!"
    raise CuteNtError(status, msg)
