"""Does the relative placement of the buffers of one call matter (same-offset accesses of two
streams landing on the same HBM channels)?  Shifts the second buffer of hamming / encode / decode by
various byte offsets inside one allocation.   usage (GPU box): python bench/placement_lab.py"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil, packed_ops as po  # noqa: E402

n = 1 << 33
PAD = 64 << 20
OFFS = [0, 128, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 8192, 17 << 20, (32 << 20) + 4096 * 5]


def timed(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 1)
pk_a = torch.empty(n // 32, dtype=torch.int64, device="cuda")
pk_big = torch.empty(n // 32 + PAD // 8, dtype=torch.int64, device="cuda")
out_big = torch.empty(n + PAD, dtype=torch.uint8, device="cuda")
cn.n_to_bits_dev(d_in, out=pk_a)
print("base addresses: in %x  pk_a %x  pk_big %x  out_big %x" % (d_in.data_ptr(), pk_a.data_ptr(), pk_big.data_ptr(), out_big.data_ptr()))
for off in OFFS:
    b = pk_big[off // 8: off // 8 + n // 32]
    o = out_big[off: off + n]
    b.copy_(pk_a)
    rows = []
    for _ in range(3):
        th = timed(lambda: po.hamming_dev(pk_a, b, n))
        te = timed(lambda: cn.n_to_bits_dev(d_in, out=b))
        td = timed(lambda: cn.bits_to_n_dev(pk_a, n, out=o))
        rows.append((th, te, td))
    th, te, td = (statistics.median(r[i] for r in rows) for i in range(3))
    print("offset %9d B: hamming %7.1f GB/s   encode(out shifted) %7.1f GB/s   decode(out shifted) %7.1f GB/s" % (
        off, 0.5 * n / th / 1e6, 1.25 * n / te / 1e6, 1.25 * n / td / 1e6), flush=True)
