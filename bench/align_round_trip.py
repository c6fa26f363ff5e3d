#!/usr/bin/env python3
"""cnt_round_trip_dev at every kind of misalignment against the aligned call (one launch at any
alignment, within 3 % of the aligned speed).  Offsets in bytes: input ASCII, packed words (multiples of 8), decoded ASCII.

usage (GPU box): python bench/align_round_trip.py [--log2-nt 34]  -> one JSON line per offset triple"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--queue", type=int, default=4)
ap.add_argument("--window-map", type=int, default=-1, help="lab build: tile map of round_trip_window (0 plain, 1 XCD pairs, 2 XCD quads); -1 = product library")
ap.add_argument("--cap", type=int, default=-1, help="lab build: resident workgroups per CU")
ap.add_argument("--plan", type=int, default=-1, help="lab build: what the launch puts on 4-KiB pages (0 decoded tiles, 1 input windows, 2 neither)")
ap.add_argument("--dense", action="store_true", help="one pointer at a time through its phases + random triples, with the launch plan's phases in every row")
a = ap.parse_args()
if a.window_map >= 0 or a.cap >= 0 or a.plan >= 0:
    from cute_nucleotides_amd import _lib

    _lib.use_lab_build()
    if a.window_map >= 0:
        devutil.set_tuning("round_trip_window_map", a.window_map)
    if a.cap >= 0:
        devutil.set_tuning("round_trip_cap", a.cap)
    if a.plan >= 0:
        devutil.set_tuning("round_trip_plan", a.plan)
n = 1 << a.log2_nt
pad = 16384
b_in = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
b_pk = torch.empty(n // 32 + pad // 8, dtype=torch.int64, device="cuda")
b_out = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
ref = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(ref, 0x5EED)
ref_sum = None
base = None
cases = ((0, 0, 0), (0, 0, 0), (1, 0, 0), (0, 8, 0), (0, 0, 1), (0, 0, 16), (0, 0, 64), (5, 8, 77), (64, 24, 4095), (127, 56, 1), (16, 0, 16), (100, 8, 2049))
if a.dense:
    import random

    rnd = random.Random(7)
    cases = [(0, 0, 0), (0, 0, 0)] + [(i, 0, 0) for i in (1, 16, 32, 48, 64, 80, 96, 112, 127)] + [(0, p, 0) for p in range(8, 64, 8)]
    cases += [(0, 0, b) for b in (1, 16, 32, 64, 96, 127, 2048, 4095)] + [(rnd.randrange(128), 8 * rnd.randrange(8), rnd.randrange(4096)) for _ in range(40)]


def plan_of(d_in, d_pk, d_out):
    import ctypes

    from cute_nucleotides_amd import _lib

    out = (ctypes.c_uint64 * 8)()
    _lib.lib().cnt_test_round_trip_plan(d_in.data_ptr(), d_pk.data_ptr(), d_out.data_ptr(), n, 0, out)
    return {"fast": int(out[0]), "phase": int(out[5]), "phase2": int(out[6]), "packed_store_mod128": int((d_pk.data_ptr() + 4 * out[2]) % 128)}


for io, po, bo in cases:
    d_in, d_pk, d_out = b_in[io : io + n], b_pk[po // 8 : po // 8 + n // 32], b_out[bo : bo + n]
    d_in.copy_(ref)
    cn.round_trip_dev(d_in, out_bits=d_pk, out_n=d_out)
    torch.cuda.synchronize()
    s = devutil.checksum_words(d_pk.contiguous()) if po % 16 == 0 else None
    ok = bool((d_out == ref).all()) if n <= (1 << 32) else (devutil.count_mismatch(ref, d_out) == 0 if bo % 16 == 0 else bool((d_out[: 1 << 30] == ref[: 1 << 30]).all()))
    ms = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.queue):
            cn.round_trip_dev(d_in, out_bits=d_pk, out_n=d_out)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / a.queue)
    med = statistics.median(ms)
    if base is None and (io, po, bo) == (0, 0, 0) and ref_sum is not None:
        base = med
    if ref_sum is None:
        ref_sum = s
    row = {"in_off": io, "packed_off": po, "back_off": bo, "ms": round(med, 4), "min_ms": round(min(ms), 4), "GBs": round(2.25 * n / med / 1e6, 1),
           "frac": round(2.25 * n / med / 1e6 / 8000, 4), "vs_aligned": round(med / base, 4) if base else None, "verified": ok and (s is None or s == ref_sum)}
    if a.dense:
        row.update(plan_of(d_in, d_pk, d_out))
    print(json.dumps(row), flush=True)
