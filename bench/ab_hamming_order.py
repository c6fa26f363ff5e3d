#!/usr/bin/env python3
"""Lab build: hamming_persist's three load orders (tuning key hamming_order: 0 interleaved = shipped, 1 blocks, 2 skewed)
alternating in one process on the same two operands -- does the order in which the two streams are requested change what
the operands' PHYSICAL placement costs (0.81-0.93 from box to box)?  One JSON line per order + the box's identity."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import _lib, devutil, packed_ops as po  # noqa: E402

_lib.use_lab_build()
n = 1 << 34
d = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d, 1)
x = cn.n_to_bits_dev(d)
devutil.fill_random_acgt(d, 2)
y = cn.n_to_bits_dev(d)
del d
acc = torch.zeros(1, dtype=torch.int64, device="cuda")
want = None
t = {0: [], 1: [], 2: []}
for rnd in range(8):
    for order in (0, 1, 2):
        devutil.set_tuning("hamming_order", order)
        if rnd == 0:
            got = int(po.hamming_dev(x, y, n).item())
            want = got if want is None else want
            assert got == want, (order, got, want)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            po.hamming_dev(x, y, n, acc=acc)
        e1.record()
        e1.synchronize()
        t[order].append(e0.elapsed_time(e1) / 5)
devutil.set_tuning("hamming_order", 0)
ident = devutil.device_identity(0)
for order in (0, 1, 2):
    ms = statistics.median(t[order])
    print(json.dumps({"hamming_order": order, "ms": round(ms, 4), "min_ms": round(min(t[order]), 4), "frac": round(0.5 * n / ms / 1e6 / 8000, 4),
                      "pci": ident["pci_bus_id"], "x_ptr": hex(x.data_ptr()), "y_ptr": hex(y.data_ptr())}))
