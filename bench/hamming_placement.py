#!/usr/bin/env python3
"""Does hamming's rate depend on where its two inputs lie relative to each other?  (packed_ops.hamming on the driver line
moved between 0.81 and 0.90 of 8 TB/s from box to box while every other kernel stayed within 2 %.)  One allocation, `a`
at its start, `b` at W + off for a list of offsets; persistent kernel and tiled kernel; 5 calls queued per event pair.

    python bench/hamming_placement.py [--log2-nt 34]"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cute_nucleotides_amd import devutil, packed_ops as po  # noqa: E402
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library


ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--queue", type=int, default=5)
a = ap.parse_args()
n = 1 << a.log2_nt
W = n // 32  # words per operand
OFFS = [0, 128, 256, 4096, 8192, 64 << 10, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 5 << 20, 16 << 20, (16 << 20) + 8192, 48 << 20]
buf = torch.empty(2 * W + (64 << 20) // 8, dtype=torch.int64, device="cuda")
buf.random_()
acc = torch.zeros(1, dtype=torch.int64, device="cuda")
x = buf[:W]


def timed(fn):
    ms = []
    fn()
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.queue):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / a.queue)
    return ms


for persistent in (1, 0):
    devutil.set_tuning("reduce_persistent", persistent)
    for rnd in range(2):
        for off in OFFS:
            y = buf[W + off // 8: 2 * W + off // 8]
            ms = timed(lambda: po.hamming_dev(x, y, n, acc=acc))
            med = statistics.median(ms)
            print(json.dumps({"persistent": persistent, "round": rnd, "b_minus_a_bytes": W * 8 + off, "extra_offset": off, "ms_median": round(med, 4), "ms_min": round(min(ms), 4),
                              "frac_of_8TBs": round(0.5 * n / med / 1e6 / 8000, 4)}), flush=True)
devutil.set_tuning("reduce_persistent", 1)

# the driver line's own allocation pattern (bench.py measure_packed_ops): a 2^34-byte ASCII buffer, then two separately
# allocated packed operands produced by the encoder
import cute_nucleotides_amd as cn  # noqa: E402

del x, buf
torch.cuda.empty_cache()
for trial in range(3):
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 11 + trial)
    x = cn.n_to_bits_dev(d)
    devutil.fill_random_acgt(d, 12 + trial)
    y = cn.n_to_bits_dev(d)
    out = torch.empty_like(x)
    for persistent in (1, 0):
        devutil.set_tuning("reduce_persistent", persistent)
        ms = timed(lambda: po.hamming_dev(x, y, n, acc=acc))
        med = statistics.median(ms)
        print(json.dumps({"bench_pattern_trial": trial, "persistent": persistent, "a": hex(x.data_ptr()), "b": hex(y.data_ptr()), "b_minus_a_bytes": y.data_ptr() - x.data_ptr(),
                          "ms_median": round(med, 4), "frac_of_8TBs": round(0.5 * n / med / 1e6 / 8000, 4)}), flush=True)
    devutil.set_tuning("reduce_persistent", 1)
    keep = torch.empty((trial + 1) * (3 << 20) + 4096, dtype=torch.uint8, device="cuda")  # perturb the next trial's placement
    del d, x, y, out

# ... and the same after what precedes it on the driver line: configs[3]'s 144 GiB were resident and have been freed
torch.cuda.empty_cache()
big = [torch.empty(1 << 36, dtype=torch.uint8, device="cuda"), torch.empty(1 << 36, dtype=torch.uint8, device="cuda"), torch.empty(1 << 34, dtype=torch.uint8, device="cuda")]
for t in big:
    t.zero_()
torch.cuda.synchronize()
del big, t
torch.cuda.empty_cache()
for trial in range(3):
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 21 + trial)
    x = cn.n_to_bits_dev(d)
    devutil.fill_random_acgt(d, 22 + trial)
    y = cn.n_to_bits_dev(d)
    for persistent in (1, 0):
        devutil.set_tuning("reduce_persistent", persistent)
        ms = timed(lambda: po.hamming_dev(x, y, n, acc=acc))
        med = statistics.median(ms)
        print(json.dumps({"after_144GiB_freed_trial": trial, "persistent": persistent, "a": hex(x.data_ptr()), "b": hex(y.data_ptr()),
                          "ms_median": round(med, 4), "frac_of_8TBs": round(0.5 * n / med / 1e6 / 8000, 4)}), flush=True)
    devutil.set_tuning("reduce_persistent", 1)
    del d, x, y
    torch.cuda.empty_cache()
