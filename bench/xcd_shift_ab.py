#!/usr/bin/env python3
"""What the XCD-aware block -> tile maps are worth, and what a WRONG XCD count costs: encode, decode and the 5-letter
decode at 2^34 nt with the tuning key "xcd_shift" = log2 of the XCD count the maps assume (device: 3 on an MI355X in SPX
mode; 0 = identity map, i.e. no XCD awareness), interleaved in one process, encode -> decode steps."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library


n = 1 << 34
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_pk = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 0x5EED)
shifts = [-1, 0, 1, 2, 3, 4]
res = {x: {"enc": [], "dec": []} for x in shifts}
for r in range(8):
    for x in (shifts if r % 2 == 0 else shifts[::-1]):
        devutil.set_tuning("xcd_shift", x)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(4)]
        torch.cuda.synchronize()
        for k in range(4):
            ev[k][0].record()
            cn.n_to_bits_dev(d_in, out=d_pk)
            ev[k][1].record()
            cn.bits_to_n_dev(d_pk, n, out=d_out)
            ev[k][2].record()
        torch.cuda.synchronize()
        if r:
            res[x]["enc"].append(statistics.fmean(e[0].elapsed_time(e[1]) for e in ev))
            res[x]["dec"].append(statistics.fmean(e[1].elapsed_time(e[2]) for e in ev))
devutil.set_tuning("xcd_shift", -1)
assert devutil.count_mismatch(d_in, d_out) == 0
base = {k: statistics.median(res[-1][k]) for k in ("enc", "dec")}
for x in shifts:
    e, d = statistics.median(res[x]["enc"]), statistics.median(res[x]["dec"])
    print(json.dumps({"xcd_shift": "device (%d)" % devutil.get_tuning("xcd_shift") if x < 0 else x, "assumed_xcds": None if x < 0 else 1 << x,
                      "encode_ms": round(e, 4), "encode_frac": round(1.25 * n / e / 1e9 / 8000, 4), "encode_vs_device": round(e / base["enc"], 4),
                      "decode_ms": round(d, 4), "decode_frac": round(1.25 * n / d / 1e9 / 8000, 4), "decode_vs_device": round(d / base["dec"], 4)}))
