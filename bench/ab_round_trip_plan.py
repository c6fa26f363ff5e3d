#!/usr/bin/env python3
"""Same-process A/B of the any-alignment fused launch plans (lab build; tuning keys round_trip_plan / round_trip_window_map):
for every offset triple every configuration in turn on the SAME buffers, so that process-to-process placement (which moves
the aligned baseline by up to 3 %) cancels.  One JSON line per triple, a summary line at the end."""
import argparse
import json
import os
import random
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import _lib, devutil  # noqa: E402

_lib.use_lab_build()
ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--triples", type=int, default=24)
ap.add_argument("--configs", default="0:0,1:0,1:1,1:2,3:0", help="plan:window_map pairs")
a = ap.parse_args()
configs = [tuple(int(x) for x in c.split(":")) for c in a.configs.split(",")]
n = 1 << a.log2_nt
pad = 16384
b_in = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
b_pk = torch.empty(n // 32 + pad // 8, dtype=torch.int64, device="cuda")
b_out = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
ref = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(ref, 0x5EED)


def timed(d_in, d_pk, d_out, rounds=5, queue=4):
    ms = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(queue):
            cn.round_trip_dev(d_in, out_bits=d_pk, out_n=d_out)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / queue)
    return statistics.median(ms)


rnd = random.Random(11)
triples = [(1, 0, 0), (0, 8, 0), (0, 0, 1), (127, 56, 4095)] + [(rnd.randrange(128), 8 * rnd.randrange(8), rnd.randrange(4096)) for _ in range(a.triples)]
b_in[:n].copy_(ref)
cn.round_trip_dev(b_in[:n], out_bits=b_pk[: n // 32], out_n=b_out[:n])
ref_sum = devutil.checksum_words(b_pk[: n // 32])
ratios = {c: [] for c in configs}
for io, po, bo in triples:
    base = timed(b_in[:n], b_pk[: n // 32], b_out[:n])  # the aligned kernel, re-measured next to every triple
    b_in[:n].copy_(ref)  # the triple's shifted copy below overwrote the head of it
    d_in, d_pk, d_out = b_in[io : io + n], b_pk[po // 8 : po // 8 + n // 32], b_out[bo : bo + n]
    d_in.copy_(ref)
    row = {"in_off": io, "packed_off": po, "back_off": bo, "aligned_ms": round(base, 4)}
    for plan, wmap in configs:
        devutil.set_tuning("round_trip_plan", plan)
        devutil.set_tuning("round_trip_window_map", wmap)
        d_out.zero_()
        cn.round_trip_dev(d_in, out_bits=d_pk, out_n=d_out)
        ok = devutil.count_mismatch(ref, d_out) == 0 if bo % 16 == 0 else bool((d_out[: 1 << 28] == ref[: 1 << 28]).all() and (d_out[-(1 << 28):] == ref[-(1 << 28):]).all())
        if po % 16 == 0:
            ok = ok and devutil.checksum_words(d_pk.contiguous()) == ref_sum
        ms = timed(d_in, d_pk, d_out)
        row["plan%d_map%d" % (plan, wmap)] = [round(ms, 4), round(ms / base, 4), bool(ok)]
        ratios[(plan, wmap)].append(ms / base)
    devutil.set_tuning("round_trip_plan", 0)
    devutil.set_tuning("round_trip_window_map", 0)
    b_in[:n].copy_(ref)
    print(json.dumps(row), flush=True)
print(json.dumps({"summary": {"plan%d_map%d" % c: {"mean": round(statistics.mean(v), 4), "max": round(max(v), 4), "min": round(min(v), 4)} for c, v in ratios.items()}}))
