#!/usr/bin/env python3
"""Device-tier throughput of the packed-domain operations on one MI355X (2^34 nt by default).
Algorithmic bytes per nucleotide: hamming 0.5 read; complement / reverse complement 0.25 read +
0.25 written; validate 1 read."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil, packed_ops as po  # noqa: E402
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library


ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reduce-persistent", type=int, default=1, help="1 = one launch of persistent waves (shipped); 0 = round 1's tiles + scratch + second pass")
ap.add_argument("--reduce-xi", type=int, default=1, help="XCD-interleaved pages for the reductions (tuning key reduce_xi)")
a = ap.parse_args()
devutil.set_tuning("reduce_xi", a.reduce_xi)
devutil.set_tuning("reduce_persistent", a.reduce_persistent)
n = 1 << a.log2_nt
d = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d, 1)
x = cn.n_to_bits_dev(d)
devutil.fill_random_acgt(d, 2)
y = cn.n_to_bits_dev(d)
out = torch.empty_like(x)


def timed(fn, inner=5):
    """median over a.iters measurements of `inner` back-to-back calls between two events (per call): the Python
    wrapper needs 20-50 us before each launch, which a single-call measurement would count as GPU time"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    ts = []
    for _ in range(a.iters):
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


acc = torch.zeros(1, dtype=torch.int64, device="cuda")  # the reductions ADD to a caller-owned counter: no kernel but theirs in the timing
rows = [
    ("hamming", 0.5, lambda: po.hamming_dev(x, y, n, acc=acc)),
    ("complement", 0.5, lambda: po.complement_dev(x, n, out=out)),
    ("reverse_complement", 0.5, lambda: po.reverse_complement_dev(x, n, out=out)),
    ("validate", 1.0, lambda: po.validate_dev(d, acc=acc)),
]
dist = int(po.hamming_dev(x, y, n).item())
assert abs(dist / n - 0.75) < 0.001, dist  # two independent uniform sequences differ in 3/4 of the positions
for name, bpn, fn in rows:
    ms = timed(fn)
    print(json.dumps({"op": name, "nt": n, "ms": round(ms, 4), "gnts": round(n / ms / 1e6, 1), "bytes_per_nt": bpn,
                      "GBs": round(bpn * n / ms / 1e6, 1), "frac_of_8TBs": round(bpn * n / ms / 1e6 / 8000, 4)}))
