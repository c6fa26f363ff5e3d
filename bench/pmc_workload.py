#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes (bench/profile.sh): a known-byte-count calibration pair
(read-only and write-only probes over the same 16 GiB buffer) followed by the codec kernels, each
launched a few times.  FETCH_SIZE / WRITE_SIZE are mis-scaled on gfx950 (MI355X_MICROARCH.md
section HBM), so bench/parse_profiles.py turns the probes' counts into calibration factors and
applies them to the codec kernels' counts."""
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
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--codec-only", action="store_true", help="calibration probes + the two 2-bit codec kernels only (bench.py's live traffic leg)")
a = ap.parse_args()
n = 1 << a.log2_nt
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_packed = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 0x5EED)
P = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
P.probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                        ctypes.c_int, ctypes.c_void_p]
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(a.reps):
    assert P.probe_run(0, 4, 1, d_in.data_ptr(), d_out.data_ptr(), n, 0, s) == 0  # k_read: reads n bytes
    assert P.probe_run(4, 4, 0, d_in.data_ptr(), d_out.data_ptr(), n, 0, s) == 0  # k_write: writes n bytes to d_out
    assert P.probe_run(1, 4, 1, d_in.data_ptr(), d_out.data_ptr(), n, 0, s) == 0  # k_copy: n read + n written
    cn.n_to_bits_dev(d_in, out=d_packed)
    cn.bits_to_n_dev(d_packed, n, out=d_out)
torch.cuda.synchronize()
assert devutil.count_mismatch(d_in, d_out) == 0
if a.codec_only:
    print("pmc workload ok (codec only): n = 2^%d, reps = %d" % (a.log2_nt, a.reps))
    raise SystemExit(0)
# secondary kernels: 5-letter codec on a 27*2^28-nt prefix of the same buffers, packed-domain ops
from cute_nucleotides_amd import packed_ops as po  # noqa: E402

n5 = 27 * (1 << 28)
if n5 <= n:
    devutil.fill_random_acgtn(d_in[:n5], 7)
    for _ in range(a.reps):
        cn.n_to_bits2_dev(d_in[:n5], out=d_packed[: n5 // 27])
        cn.bits_to_n2_dev(d_packed[: n5 // 27], n5, out=d_out[:n5])
    torch.cuda.synchronize()
    assert devutil.count_mismatch(d_in[:n5], d_out[:n5]) == 0
devutil.fill_random_acgt(d_in, 0x5EED)
cn.n_to_bits_dev(d_in, out=d_packed)
other = torch.empty_like(d_packed)
acc = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(a.reps):
    po.complement_dev(d_packed, n, out=other)
    po.hamming_dev(d_packed, other, n)
    po.validate_dev(d_in)
    cn.n_to_bits_checked_dev(d_in, out=d_packed, acc=acc)  # the validated encode: its traffic must be the plain encode's (1.25 B/nt)
assert int(acc.item()) == 0
torch.cuda.synchronize()
print("pmc workload ok: n = 2^%d, reps = %d" % (a.log2_nt, a.reps))
