#!/usr/bin/env python3
"""Device-tier throughput of the 5-letter codec (n_to_bits2 / bits_to_n2) on one MI355X.
Algorithmic bytes per nucleotide: 1 + 8/27 = 1.2963 in each direction.
    python bench/bench_codec5.py [--log2-words 29] [--iters 10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library


ap = argparse.ArgumentParser()
ap.add_argument("--log2-words", type=int, default=29, help="packed words = 2^k (nt = 27 * 2^k; 29 -> 13.5 GiB of ASCII)")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--encode2", default="", help="comma list: only these encode variants, in interleaved rounds")
ap.add_argument("--decode2", default="", help="comma list: only these decode variants, in interleaved rounds")
a = ap.parse_args()
words = 1 << a.log2_words
n = 27 * words
d = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgtn(d, 0x5EED)
packed = torch.empty(words, dtype=torch.int64, device="cuda")
back = torch.empty(n, dtype=torch.uint8, device="cuda")
cn.n_to_bits2_dev(d, out=packed)
cn.bits_to_n2_dev(packed, n, out=back)
torch.cuda.synchronize()
assert devutil.count_mismatch(d, back) == 0


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(a.iters):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


bpn = 1.0 + 8.0 / 27.0
rows = []
for key, fn in (("encode2", lambda: cn.n_to_bits2_dev(d, out=packed)), ("decode2", lambda: cn.bits_to_n2_dev(packed, n, out=back))):
    only = [int(x) for x in getattr(a, key).split(",") if x]
    if only:  # interleaved: one launch of every chosen variant per round, median over the rounds
        names = dict(devutil.variants(key))
        ms = {v: [] for v in only}
        for rnd in range(a.iters + 1):
            for v in only:
                devutil.set_tuning(key, v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                if rnd:
                    ms[v].append(e0.elapsed_time(e1))
        for v in only:
            m = sorted(ms[v])[len(ms[v]) // 2]
            rows.append({"what": key, "variant": v, "kernel": names[v], "ms": round(m, 4), "min_ms": round(min(ms[v]), 4), "frac_of_8TBs": round(bpn * n / m / 1e6 / 8000, 4)})
        devutil.set_tuning(key, 0)
        continue
    for v, name in devutil.variants(key):
        devutil.set_tuning(key, v)
        fn()
        ms = timed(fn)
        rows.append({"what": key, "variant": v, "kernel": name, "ms": round(ms, 4), "gnts": round(n / ms / 1e6, 1),
                     "GBs": round(bpn * n / ms / 1e6, 1), "frac_of_8TBs": round(bpn * n / ms / 1e6 / 8000, 4)})
    devutil.set_tuning(key, 0)
assert devutil.count_mismatch(d, back) == 0
for r in rows:
    print(json.dumps(r))
