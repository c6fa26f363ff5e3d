#!/usr/bin/env python3
"""The page-tiled 5-letter decoder (bits_to_n2_page, lab variants decode2 50-58) against the shipped word-tiled kernel:
first bit-exactness against the shipped kernel's output at ragged lengths and every output alignment class, then isolated
launches over 2^34 nt, interleaved rounds, median / min per variant.  One JSON line per variant."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import _lib, devutil  # noqa: E402

_lib.use_lab_build()
variants = [0] + [v for v, _ in devutil.variants("decode2") if v >= (70 if "--k1" in sys.argv else 50)]
names = dict(devutil.variants("decode2"))

# parity against the shipped kernel (itself checked against the oracle by tests/test_gpu_codec5.py)
n_chk = 27 * 40000 + 13
d = torch.empty(n_chk, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgtn(d, 77)
pk = cn.n_to_bits2_dev(d)
bad = 0
for v in variants[1:]:
    for off in (0, 1, 5, 27, 64, 100, 127):
        for length in (n_chk, n_chk - 1, n_chk - 27, 4096 + off, 4095, 8192 + 128, 27 * 1000, 12345):
            buf = torch.full((n_chk + 256,), 0x7A, dtype=torch.uint8, device="cuda")
            base = (-buf.data_ptr()) % 128 + off
            out = buf[base: base + length]
            devutil.set_tuning("decode2", v)
            cn.bits_to_n2_dev(pk, length, out=out)
            devutil.set_tuning("decode2", 0)
            torch.cuda.synchronize()
            if not torch.equal(out, d[:length]) or int((buf[:base] != 0x7A).sum()) or int((buf[base + length:] != 0x7A).sum()):
                bad += 1
                print("MISMATCH", v, off, length, file=sys.stderr)
print(json.dumps({"parity_cases_failed": bad}), flush=True)
if bad:
    sys.exit(1)

words = ((1 << 34) // 27) // 128 * 128
n5 = 27 * words
d_n = torch.empty(n5, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgtn(d_n, 0x5EED)
d_w = cn.n_to_bits2_dev(d_n)
d_b = torch.empty(n5, dtype=torch.uint8, device="cuda")
ms = {v: [] for v in variants}
for rnd in range(8):
    for v in variants:
        devutil.set_tuning("decode2", v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cn.bits_to_n2_dev(d_w, n5, out=d_b)
        e1.record()
        e1.synchronize()
        if rnd:
            ms[v].append(e0.elapsed_time(e1))
        elif devutil.count_mismatch(d_n, d_b):
            print("MISMATCH at 2^34", v, file=sys.stderr)
            sys.exit(1)
devutil.set_tuning("decode2", 0)
for v in variants:
    med = statistics.median(ms[v])
    print(json.dumps({"variant": v, "name": names[v], "ms": round(med, 4), "min_ms": round(min(ms[v]), 4),
                      "frac": round((n5 + 8 * words) / med / 1e-3 / 8e12, 4)}), flush=True)
