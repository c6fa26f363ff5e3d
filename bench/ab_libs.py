#!/usr/bin/env python3
"""Same-process A/B of several BUILDS of the library (different source revisions / compiler flags), interleaved launch by
launch so that box-to-box and allocation-to-allocation variance cancels:

    python bench/ab_libs.py --libs cur=cute_nucleotides_amd/libcute_nt_hip.so r02=bench/libcute_nt_hip_r02.so \
        pre=bench/libcute_nt_hip_preload.so --log2-nt 34 --rounds 12

Every library is loaded with its own ctypes handle (RTLD_LOCAL: the identical extern "C" names do not clash, every
code object registers its own kernels); all of them work on the SAME device buffers.  Per round and library: one
encode, one decode, one fused round trip, each between two HIP events on torch's current stream.  Prints one JSON
line per (library, op) with mean / median / min ms and TB/s, plus the ratio to the first library.  Optional
--decode-variants / --encode-variants sweep cnt_set_tuning() values of the FIRST library in the same interleaving
(decode variants against each other and against the 1:4 probe in one process)."""
import argparse
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", default=["cur=bench/libcute_nt_hip_lab.so"],
                help="name=path ...; --decode-variants / --encode-variants need a LAB build first in the list (the product library has no cnt_set_tuning)")
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--minus", type=int, default=0, help="subtract this many nt (ragged sizes)")
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--decode-variants", default="", help="comma list: extra rows of the first library with cnt_set_tuning('decode', v)")
ap.add_argument("--encode-variants", default="")
ap.add_argument("--probes", action="store_true", help="also time bench/libcnt_probes.so's 4:1 and 1:4 no-arithmetic streams")
ap.add_argument("--arith", action="store_true", help="with --probes: the 4:1 / 1:4 streams with the codec's arithmetic applied 1, 2, 4, 8 times")
ap.add_argument("--packed-ops", action="store_true", help="also time hamming / complement / validate of every library")
ap.add_argument("--queue", type=int, default=1, help="launches per event pair (1 = isolated launches)")
a = ap.parse_args()

_vp, _sz, _u = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint


def load(path):
    L = ctypes.CDLL(os.path.join(ROOT, path))
    L.cnt_n_to_bits_dev.argtypes = [_vp, _sz, _vp, _sz, _u, _vp]
    L.cnt_bits_to_n_dev.argtypes = [_vp, _sz, _sz, _vp, _u, _vp]
    L.cnt_round_trip_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _u, _vp]
    L.cnt_fill_random_acgt_dev.argtypes = [_vp, _sz, _sz, ctypes.c_uint64, _vp]
    L.cnt_count_mismatch_dev.argtypes = [_vp, _vp, _sz, _vp, _vp]
    L.cnt_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.cnt_hamming_dev.argtypes = [_vp, _vp, _sz, _vp, _vp]
    L.cnt_complement_dev.argtypes = [_vp, _sz, _vp, _vp]
    L.cnt_reverse_complement_dev.argtypes = [_vp, _sz, _vp, _vp]
    L.cnt_validate_dev.argtypes = [_vp, _sz, _u, _vp, _vp]
    return L


libs = [(s.split("=", 1)[0], load(s.split("=", 1)[1])) for s in a.libs]
n_len = (1 << a.log2_nt) - a.minus
words = (n_len + 31) // 32
dev = torch.device("cuda", 0)
d_in = torch.empty(n_len, dtype=torch.uint8, device=dev)
d_pk = torch.empty(words, dtype=torch.int64, device=dev)
d_out = torch.empty(n_len, dtype=torch.uint8, device=dev)
stream = _vp(torch.cuda.current_stream().cuda_stream)
assert libs[0][1].cnt_fill_random_acgt_dev(d_in.data_ptr(), 0, n_len, 0x5EED, stream) == 0
torch.cuda.synchronize()

rows = []  # (label, op, callable)
for name, L in libs:
    rows.append((name, "encode", lambda L=L: L.cnt_n_to_bits_dev(d_in.data_ptr(), n_len, d_pk.data_ptr(), words, 0, stream)))
    rows.append((name, "decode", lambda L=L: L.cnt_bits_to_n_dev(d_pk.data_ptr(), words, n_len, d_out.data_ptr(), 0, stream)))
    rows.append((name, "fused", lambda L=L: L.cnt_round_trip_dev(d_in.data_ptr(), n_len, d_pk.data_ptr(), words, d_out.data_ptr(), 0, stream)))
if a.packed_ops:
    d_pk2 = torch.empty(words, dtype=torch.int64, device=dev)
    d_acc = torch.zeros(2, dtype=torch.int64, device=dev)
    assert libs[0][1].cnt_n_to_bits_dev(d_in.data_ptr(), n_len, d_pk.data_ptr(), words, 0, stream) == 0
    assert libs[0][1].cnt_fill_random_acgt_dev(d_out.data_ptr(), 0, n_len, 0xBEEF, stream) == 0
    assert libs[0][1].cnt_n_to_bits_dev(d_out.data_ptr(), n_len, d_pk2.data_ptr(), words, 0, stream) == 0
    torch.cuda.synchronize()
    for name, L in libs:  # these rows must not run interleaved with encode / decode rows of the same buffers: use --packed-ops alone
        rows.append((name, "hamming", lambda L=L: L.cnt_hamming_dev(d_pk.data_ptr(), d_pk2.data_ptr(), n_len, d_acc.data_ptr(), stream)))
        rows.append((name, "validate", lambda L=L: L.cnt_validate_dev(d_in.data_ptr(), n_len, 0, d_acc.data_ptr() + 8, stream)))
        rows.append((name, "complement", lambda L=L: L.cnt_complement_dev(d_pk.data_ptr(), n_len, d_out.data_ptr(), stream)))
        rows.append((name, "reverse_complement", lambda L=L: L.cnt_reverse_complement_dev(d_pk.data_ptr(), n_len, d_out.data_ptr(), stream)))
    rows = [r for r in rows if r[1] in ("hamming", "validate", "complement", "reverse_complement")]
L0 = libs[0][1]
for key, spec in (("decode", a.decode_variants), ("encode", a.encode_variants)):
    for v in [int(x) for x in spec.split(",") if x]:
        def call(key=key, v=v):
            L0.cnt_set_tuning(key.encode(), v)
            rc = (L0.cnt_bits_to_n_dev(d_pk.data_ptr(), words, n_len, d_out.data_ptr(), 0, stream) if key == "decode"
                  else L0.cnt_n_to_bits_dev(d_in.data_ptr(), n_len, d_pk.data_ptr(), words, 0, stream))
            L0.cnt_set_tuning(key.encode(), 0)
            return rc
        rows.append(("%s v%d" % (libs[0][0], v), key, call))
if a.probes and a.minus == 0:
    P = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
    P.probe_shipped.argtypes = [ctypes.c_int, _vp, _vp, _sz, _vp]
    rows.append(("probe 4:1", "encode", lambda: P.probe_shipped(2, d_in.data_ptr(), d_out.data_ptr(), n_len, stream)))
    rows.append(("probe 1:4", "decode", lambda: P.probe_shipped(3, d_in.data_ptr(), d_out.data_ptr(), n_len, stream)))
    if a.arith:  # the same streams with the codec's arithmetic applied R times in a dependent chain
        for r in (1, 2, 4, 8):
            rows.append(("probe 4:1 arith x%d" % r, "encode", lambda r=r: P.probe_shipped(20 + r, d_in.data_ptr(), d_out.data_ptr(), n_len, stream)))
            rows.append(("probe 1:4 arith x%d" % r, "decode", lambda r=r: P.probe_shipped(10 + r, d_in.data_ptr(), d_out.data_ptr(), n_len, stream)))

for _, _, fn in rows:  # warm-up: module load, stream creation
    assert fn() == 0
torch.cuda.synchronize()
ms = {(lab, op): [] for lab, op, _ in rows}
for r in range(a.rounds):
    order = rows if r % 2 == 0 else rows[::-1]  # alternate the order: no row always runs behind the same neighbour
    for lab, op, fn in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.queue):
            rc = fn()
        e1.record()
        e1.synchronize()
        assert rc == 0
        ms[(lab, op)].append(e0.elapsed_time(e1) / a.queue)

# the last writer of d_pk / d_out was some library's encode / decode of d_in: a final round trip check
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for name, L in ([] if a.packed_ops else libs):
    assert L.cnt_n_to_bits_dev(d_in.data_ptr(), n_len, d_pk.data_ptr(), words, 0, stream) == 0
    assert L.cnt_bits_to_n_dev(d_pk.data_ptr(), words, n_len, d_out.data_ptr(), 0, stream) == 0
    cnt.zero_()
    assert libs[0][1].cnt_count_mismatch_dev(d_in.data_ptr(), d_out.data_ptr(), n_len, cnt.data_ptr(), stream) == 0
    assert int(cnt.item()) == 0, "round trip failed for " + name

base = {}
for lab, op, _ in rows:
    v = ms[(lab, op)]
    bytes_moved = {"fused": 2.25, "hamming": 0.5, "validate": 1.0}.get(op, 1.25) * n_len
    med = statistics.median(v)
    base.setdefault(op, med)
    print(json.dumps({"lib": lab, "op": op, "nt": n_len, "queue": a.queue, "ms_mean": round(statistics.fmean(v), 4), "ms_median": round(med, 4), "ms_min": round(min(v), 4),
                      "TBs_median": round(bytes_moved / med / 1e9, 3), "frac_of_8TBs": round(bytes_moved / med / 1e9 / 8.0, 4),
                      "vs_first_row_of_op": round(med / base[op], 4)}), flush=True)
