"""Does cache-blocking the encode->decode round trip (decode chunk i right after encoding it, while
its packed words may still sit in the 256 MB Infinity Cache) beat two full passes?
usage (GPU box): python bench/blocked_step_lab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402

n = 1 << 34
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_pk = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 0x5EED)


def step(chunk):
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        cn.n_to_bits_dev(d_in[lo:hi], out=d_pk[lo // 32:hi // 32])
        cn.bits_to_n_dev(d_pk[lo // 32:hi // 32], hi - lo, out=d_out[lo:hi])


for log2c in (34, 32, 30, 29, 28, 27, 26, 25, 24):
    c = 1 << log2c
    step(c)
    assert devutil.count_mismatch(d_in, d_out) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        step(c)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("chunk 2^%d nt (packed %6.1f MiB): step %.3f ms = %.1f Gnt/s (2N/t), %.1f GB/s algorithmic" % (
        log2c, c / 4 / 2**20, ms, 2 * n / ms / 1e6, 2.5 * n / ms / 1e6), flush=True)
