#!/usr/bin/env python3
"""Per-launch decode durations in three contexts (behind an encode of the same buffers = bench.py's step; five decodes back to
back; isolated between two synchronisations) for the shipped decode shape and rounds 1-2's (variant 35):
profiles/r03_decode_contexts.log."""
import sys, statistics
sys.path.insert(0, "/root/repo")
import torch
import cute_nucleotides_amd as cn
from cute_nucleotides_amd import devutil
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library

n = 1 << 34
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_p = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 1)
cn.n_to_bits_dev(d_in, out=d_p)
def dec_times(ctx, v, reps=40):
    devutil.set_tuning("decode", v)
    ts = []
    if ctx == "after_encode":
        for _ in range(reps // 5):
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(5)]
            torch.cuda.synchronize()
            for k in range(5):
                cn.n_to_bits_dev(d_in, out=d_p); ev[k][0].record(); cn.bits_to_n_dev(d_p, n, out=d_out); ev[k][1].record()
            torch.cuda.synchronize()
            ts += [e[0].elapsed_time(e[1]) for e in ev]
    elif ctx == "back_to_back_decodes":
        for _ in range(reps // 5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            torch.cuda.synchronize()
            ev[0].record()
            for k in range(5):
                cn.bits_to_n_dev(d_p, n, out=d_out); ev[k + 1].record()
            torch.cuda.synchronize()
            ts += [ev[k].elapsed_time(ev[k + 1]) for k in range(5)]
    else:
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); cn.bits_to_n_dev(d_p, n, out=d_out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts
for rnd in range(2):
    for v in (0, 35):
        for ctx in ("after_encode", "back_to_back_decodes", "isolated"):
            ts = dec_times(ctx, v)
            print("v%-2d %-22s mean %.4f  deciles %s" % (v, ctx, statistics.mean(ts), " ".join("%.3f" % ts[min(len(ts) - 1, len(ts) * q // 10)] for q in range(11))), flush=True)
devutil.set_tuning("decode", 0)
