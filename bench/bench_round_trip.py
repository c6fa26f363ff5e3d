"""Fused encode+decode (cnt_round_trip_dev, BASELINE.json configs[3]) against the two-pass step.
usage (GPU box): python bench/bench_round_trip.py [--log2-nt 34] [--caps 0,12,14,16,18,20,23]"""
import argparse
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


ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--caps", default="8")
ap.add_argument("--shapes", default="0", help="comma list of round_trip_shape values (0 = default, 1 = first shipped shape)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=4)
a = ap.parse_args()
n = 1 << a.log2_nt
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_pk = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 0x5EED)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / a.iters


def two_pass():
    cn.n_to_bits_dev(d_in, out=d_pk)
    cn.bits_to_n_dev(d_pk, n, out=d_out)


two_pass()
ref = devutil.checksum_words(d_pk)
cases = [("two passes (n_to_bits_dev + bits_to_n_dev)", None, 2.5)] + [
    ("fused round trip, shape %s, %s wg/CU" % (sh, c or "uncapped"), (int(sh), int(c)), 2.25) for sh in a.shapes.split(",") for c in a.caps.split(",")]
times = {c[0]: [] for c in cases}
for r in range(a.rounds + 1):
    for name, cap, _ in cases:
        if cap is None:
            t = timed(two_pass)
        else:
            devutil.set_tuning("round_trip_shape", cap[0])
            devutil.set_tuning("round_trip_cap", cap[1])
            d_pk.zero_()
            t = timed(lambda: cn.round_trip_dev(d_in, out_bits=d_pk, out_n=d_out))
            assert devutil.checksum_words(d_pk) == ref and devutil.count_mismatch(d_in, d_out) == 0
        if r:
            times[name].append(t)
for name, cap, bpn in cases:
    ms = statistics.median(times[name])
    print(json.dumps({"what": name, "nt": n, "ms": round(ms, 4), "gnts_round_trip": round(n / ms / 1e6, 1), "bytes_per_nt": bpn,
                      "GBs": round(bpn * n / ms / 1e6, 1), "frac_of_8TBs": round(bpn * n / ms / 1e6 / 8000, 4)}), flush=True)
