#!/usr/bin/env python3
"""cnt_bits_to_n_dev with the packed stream entered at every kind of offset inside a 4-KiB page (the output peeled to a page as
always).  Round 4 asked "why does decode off the grid cost between 0 and 12 %" and answered "the four-tiles-per-XCD map wants
the packed stream's 4-KiB pieces on pages".  Round 5 moved the turns and found the rest; for every offset
pair this script times, INTERLEAVED over several rounds in one process on the same buffers,

    k0..k3          round 4's kernels (stream at any dword phase, shifted for a bit phase) with k further output pages peeled into
                    the head -- every turn's packed piece moves by k KiB (device_tier.inc decode_turn_pages); k0 = round 4's launch
    after, nearest  ... with the k that starts the turns in [page, page + 1 KiB) / within 512 B of a page boundary of the packed buffer
    window_k0       bits_to_n_window (line-aligned 16-B loads + a funnel read of the wave's slab) for streams off their lines or
                    dwords, no turn placement
    window_nearest  ... with the rule
    shipped         the product's plan (tuning values -1 / 1: window + nearest past the Infinity Cache, round 4's launch inside it)
    plain           plain dispatch order (lab variant 42; no turns to misplace)

and prints one JSON line per pair with the GB/s (1.25 B/nt) of each -- median over rounds of the mean of `it` queued launches
-- plus the per-k packed residue r_k = (first tile's packed byte address) mod 4096.  profiles/r05_decode_off_grid.md reads
four boxes' sweeps and the size table (CNT_OFF_GRID_FEW_MODES=1 --quick --log2-nt 22..32).

    python bench/decode_off_grid.py [--log2-nt 34] [--rounds 6] [--it 4] [--quick]"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import _lib, devutil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--it", type=int, default=4)
ap.add_argument("--quick", action="store_true", help="eight offsets instead of the full sweep")
a = ap.parse_args()

_lib.use_lab_build()
n = 1 << a.log2_nt
pad = 32768
b_in = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
b_out = torch.empty(n + pad, dtype=torch.uint8, device="cuda")
b_pk = torch.empty(n // 32 + pad // 8, dtype=torch.int64, device="cuda")
devutil.fill_random_acgt(b_in[:n], 1)
base_out, base_pk = b_out.data_ptr(), b_pk.data_ptr()
assert base_out % 4096 == 0 and base_pk % 4096 == 0


def once(fn, it):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / it


# (ascii offset, packed offset): the ASCII offsets 0 / 16 keep the stream kernel (head a multiple of 16 nt), 5 / 77 take the
# shifted kernel (two dwords through v_alignbit); packed offsets walk the page in 256-B steps plus the neighbours of 0 and 1 KiB
packed = [0, 8, 16, 64, 128, 256, 264, 512, 520, 776, 1016, 1024, 1032, 1288, 1544, 1800, 2048, 2056, 2312, 2568, 2824, 3072, 3080, 3088, 3336, 3592, 3848, 4088]
cases = [(0, p) for p in packed] + [(16, p) for p in packed] + [(5, p) for p in (0, 8, 264, 1032, 2056, 3080)] + [(77, p) for p in (8, 1544, 3592)]
if a.quick:
    cases = [(0, 0), (0, 8), (0, 1032), (0, 2056), (16, 8), (16, 264), (16, 2312), (5, 264)]

# (name, decode variant, decode_rot, decode_window): round 4's kernels (stream at any dword phase, shifted for a bit phase) with
# k further pages / the two rules; round 5's window kernel (line-aligned 16-B loads + a funnel read of the wave's slab) with no
# turn placement, with the rule forced, and as shipped (the rule from 2^30 nt on); plain dispatch order
MODES = (("k0", 0, 0, 0), ("k1", 0, 1, 0), ("k2", 0, 2, 0), ("k3", 0, 3, 0), ("after", 0, 10, 0), ("nearest", 0, 11, 0),
         ("window_k0", 0, 0, 1), ("window_nearest", 0, 11, 1), ("shipped", 0, -1, 1), ("plain", 42, 0, 0))
if os.environ.get("CNT_OFF_GRID_FEW_MODES") == "1":  # size sweeps: the four that matter
    MODES = tuple(m for m in MODES if m[0] in ("k0", "nearest", "window_k0", "shipped"))
for a_off, p_off in cases:
    d_out = b_out[a_off : a_off + n]
    d_pk = b_pk[p_off // 8 : p_off // 8 + n // 32]
    cn.n_to_bits_dev(b_in[:n], out=d_pk)
    head = (4096 - (base_out + a_off) % 4096) % 4096
    r0 = (base_pk + p_off + 4 * (head >> 4)) % 4096
    row = {"ascii_off": a_off, "packed_off": p_off, "kernel": "stream" if head % 16 == 0 else "shifted", "r": r0,
           "r_k": [(r0 + 1024 * k) % 4096 for k in range(4)], "k_after": (4 - (r0 >> 10)) & 3, "k_nearest": (4 - ((r0 + 512) >> 10)) & 3}
    ms = {m[0]: [] for m in MODES}

    def call():
        cn.bits_to_n_dev(d_pk, n, out=d_out)

    def select(variant, rot, window):
        devutil.set_tuning("decode", variant)
        devutil.set_tuning("decode_rot", rot)
        devutil.set_tuning("decode_window", window)

    for name, variant, rot, window in MODES:  # warm every mode once
        select(variant, rot, window)
        call()
    torch.cuda.synchronize()
    for rnd in range(a.rounds):
        order = MODES if rnd % 2 == 0 else MODES[::-1]  # alternate the order: a drift does not favour one mode
        for name, variant, rot, window in order:
            select(variant, rot, window)
            ms[name].append(once(call, a.it))
    for m in MODES:
        row[m[0]] = round(1.25 * n / statistics.median(ms[m[0]]) / 1e6, 1)
        row[m[0] + "_best"] = round(1.25 * n / min(ms[m[0]]) / 1e6, 1)
    for variant, rot, window in ((0, 3, 1), (0, 3, 0), (0, -1, 1)):  # the largest heads through the edge items, both kernel families; then as shipped
        select(variant, rot, window)
        d_out.zero_()
        call()
        assert devutil.count_mismatch(b_in[:n].contiguous(), d_out.contiguous()) == 0, (a_off, p_off, rot, window)
    print(json.dumps(row), flush=True)
