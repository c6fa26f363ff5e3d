#!/usr/bin/env python3
"""Workload for bench/pmc_bound.sh: the shipped-shape no-arithmetic streams (bench/probes.hip probe_shipped:
read-only, write-only, 1:1 copy, 4:1 = encode's shape, 1:4 = decode's shape) and the three codec kernels
(encode, decode, fused round trip), each launched `reps` times over the same 2^34-nt buffers, so that one
rocprofv3 --pmc pass sees the L2<->fabric request counters of all of them side by side."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--what", choices=("codec2", "codec5"), default="codec2")
a = ap.parse_args()
if a.what == "codec5":
    # the 5-letter codec at the metric size (2^34 nt rounded to whole wave tiles): shipped encode / decode beside their
    # arithmetic-free twins with and without the LDS staging (bench/probes.hip probe_codec5)
    words = ((1 << a.log2_nt) // 27) // 128 * 128
    n5 = 27 * words
    d_n = torch.empty(n5 + 256, dtype=torch.uint8, device="cuda")[:n5]
    d_w = torch.empty(words, dtype=torch.int64, device="cuda")
    d_b = torch.empty(n5 + 256, dtype=torch.uint8, device="cuda")[:n5]
    devutil.fill_random_acgtn(d_n, 0x5EED)
    P = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
    P.probe_codec5.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scratch_w = torch.empty(words, dtype=torch.int64, device="cuda")
    for _ in range(a.reps):
        for kind in (1, 2):
            assert P.probe_codec5(kind, d_n.data_ptr(), scratch_w.data_ptr(), words, s) == 0
        cn.n_to_bits2_dev(d_n, out=d_w)
        for kind in (3, 4):
            assert P.probe_codec5(kind, d_w.data_ptr(), d_b.data_ptr(), words, s) == 0
        cn.bits_to_n2_dev(d_w, n5, out=d_b)
    torch.cuda.synchronize()
    assert devutil.count_mismatch(d_n, d_b) == 0
    print("pmc bound workload (codec5) ok: %d words, reps = %d" % (words, a.reps))
    sys.exit(0)
n = 1 << a.log2_nt
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_packed = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 0x5EED)
P = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
P.probe_shipped.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(a.reps):
    for kind in (0, 4, 1, 2, 3, 11, 12, 14, 21, 24):  # 1x: the 1:4 / 4:1 streams with the codec's arithmetic applied x times
        assert P.probe_shipped(kind, d_in.data_ptr(), d_out.data_ptr(), n, s) == 0
    cn.n_to_bits_dev(d_in, out=d_packed)
    cn.bits_to_n_dev(d_packed, n, out=d_out)
    cn.round_trip_dev(d_in, out_bits=d_packed, out_n=d_out)
torch.cuda.synchronize()
assert devutil.count_mismatch(d_in, d_out) == 0
print("pmc bound workload ok: n = 2^%d, reps = %d" % (a.log2_nt, a.reps))
