#!/usr/bin/env python3
"""cnt_bits_to_n_dev with the packed stream entered at every kind of offset inside a 4-KiB page (the output peeled to a page as
always): the shipped four-tiles-per-XCD map against plain dispatch order (lab variant 42), same buffers, one JSON line per offset.
Answers "why does decode off the grid cost between 0 and 12 %": the XCD map wants the packed stream's 4-KiB pieces on pages."""
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
n = 1 << 34
pad = 16384
b_in = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
b_out = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
b_pk = torch.empty(n // 32 + pad // 8, dtype=torch.int64, device="cuda")
devutil.fill_random_acgt(b_in[:n], 1)


def timed(fn, it=4):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / it)
    return statistics.median(ms)


cases = [(0, 0), (0, 8), (0, 1032), (0, 2056)] + [(16, p) for p in (0, 8, 264, 520, 776, 1032, 1288, 1544, 1800, 2056, 2312, 2568, 2824, 3080, 3088, 3336, 3592, 3848, 4088)]
for a_off, p_off in cases:
    d_out = b_out[a_off : a_off + n]
    d_pk = b_pk[p_off // 8 : p_off // 8 + n // 32]
    cn.n_to_bits_dev(b_in[:n], out=d_pk)
    head = (4096 - a_off) % 4096
    row = {"ascii_off": a_off, "packed_off": p_off, "first_tile_packed_byte_mod_4096": (p_off + 4 * (head >> 4)) % 4096}
    for v, name in ((0, "xcd_quads_GBs"), (42, "plain_order_GBs")):
        devutil.set_tuning("decode", v)
        row[name] = round(1.25 * n / timed(lambda: cn.bits_to_n_dev(d_pk, n, out=d_out)) / 1e6, 1)
    devutil.set_tuning("decode", 0)
    assert devutil.count_mismatch(b_in[:n].contiguous(), d_out.contiguous()) == 0
    print(json.dumps(row), flush=True)
