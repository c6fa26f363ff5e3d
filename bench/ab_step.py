#!/usr/bin/env python3
"""A/B of kernel variants in bench.py's own context: full steps (encode, then decode, back to
back on one stream) for chosen (encode variant, decode variant) pairs, interleaved rounds,
per-kernel HIP-event times.  python bench/ab_step.py --enc 0,9,10 --dec 0,9,10,5 --rounds 7"""
import argparse
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


ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--enc", default="0,9")
ap.add_argument("--dec", default="0,9")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--five", action="store_true", help="the 5-letter codec (tuning keys encode2 / decode2)")
ap.add_argument("--dist", action="store_true", help="also print every decode launch's time, sorted (is a median hiding two modes?)")
a = ap.parse_args()
n = 1 << a.log2_nt
if a.five:
    n = n // 3456 * 3456
per, bpn = (27, 1 + 8 / 27) if a.five else (32, 1.25)
K_ENC, K_DEC = ("encode2", "decode2") if a.five else ("encode", "decode")
enc_fn = cn.n_to_bits2_dev if a.five else cn.n_to_bits_dev
dec_fn = cn.bits_to_n2_dev if a.five else cn.bits_to_n_dev
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_packed = torch.empty(n // per, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
(devutil.fill_random_acgtn if a.five else devutil.fill_random_acgt)(d_in, 0x5EED)
encs = [int(x) for x in a.enc.split(",")]
decs = [int(x) for x in a.dec.split(",")]
pairs = [(e, d) for e in encs for d in decs]
res = {p: {"enc": [], "dec": [], "step": [], "dec_each": [], "enc_each": []} for p in pairs}
names_e, names_d = dict(devutil.variants(K_ENC)), dict(devutil.variants(K_DEC))
for r in range(a.rounds + 1):
    for p in pairs:
        devutil.set_tuning(K_ENC, p[0])
        devutil.set_tuning(K_DEC, p[1])
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
        torch.cuda.synchronize()
        for k in range(a.steps):
            ev[k][0].record()
            enc_fn(d_in, out=d_packed)
            ev[k][1].record()
            dec_fn(d_packed, n, out=d_out)
            ev[k][2].record()
        torch.cuda.synchronize()
        if r == 0:
            assert devutil.count_mismatch(d_in, d_out) == 0, p
            continue
        res[p]["enc"].append(sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps)
        res[p]["dec"].append(sum(e[1].elapsed_time(e[2]) for e in ev) / a.steps)
        res[p]["step"].append(ev[0][0].elapsed_time(ev[-1][2]) / a.steps)
        res[p]["enc_each"] += [e[0].elapsed_time(e[1]) for e in ev]
        res[p]["dec_each"] += [e[1].elapsed_time(e[2]) for e in ev]
devutil.set_tuning(K_ENC, 0)
devutil.set_tuning(K_DEC, 0)
for p in sorted(pairs, key=lambda p: statistics.median(res[p]["step"])):
    e, d, s = (statistics.median(res[p][k]) for k in ("enc", "dec", "step"))
    print("enc v%-2d dec v%-2d  step %.4f ms  enc %.4f ms = %6.1f GB/s  dec %.4f ms = %6.1f GB/s   | %s | %s" % (
        p[0], p[1], s, e, bpn * n / e / 1e6, d, bpn * n / d / 1e6, names_e[p[0]], names_d[p[1]]))
    if a.dist:
        for k in ("enc_each", "dec_each"):
            v = sorted(res[p][k])
            print("    %s mean %.4f  deciles %s" % (k[:3], statistics.mean(v), " ".join("%.3f" % v[min(len(v) - 1, len(v) * q // 10)] for q in range(11))))
