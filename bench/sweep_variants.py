#!/usr/bin/env python3
"""Within-process A/B sweep of the codec kernel variants and of the no-arithmetic bandwidth
probes (bench/probes.hip) on one MI355X.  Interleaved rounds, median + min per variant.

    python bench/sweep_variants.py --log2-nt 34 --rounds 7 --out gpurun_out/sweep.json
"""
import argparse
import ctypes
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



def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-nt", type=int, default=34)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--skip-probes", action="store_true")
    ap.add_argument("--offset", type=int, default=0, help="shift every buffer by this many bytes (multiple of 16)")
    args = ap.parse_args()

    n = 1 << args.log2_nt
    dev = torch.device("cuda", 0)
    off = args.offset
    assert off % 16 == 0
    d_in = torch.empty(n + off, dtype=torch.uint8, device=dev)[off:]
    d_packed = torch.empty(n // 32 + off // 8, dtype=torch.int64, device=dev)[off // 8:]
    d_out = torch.empty(n + off, dtype=torch.uint8, device=dev)[off:]
    devutil.fill_random_acgt(d_in, 0x5EED)
    cn.n_to_bits_dev(d_in, out=d_packed)
    torch.cuda.synchronize()
    ref_sum = devutil.checksum_words(d_packed)

    cases = [("encode", name, i, 0, 0) for i, name in devutil.variants("encode")]
    cases += [("decode", name, i, 0, 0) for i, name in devutil.variants("decode")]

    probes = None
    if not args.skip_probes:
        probes = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
        probes.probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        for kind in range(5):
            for u in (2, 4, 8) if kind in (0, 1, 4) else (1, 2):
                for nt in (0, 1):
                    for g in (0, 2048):
                        cases.append(("probe", kind, u, nt, g))

    times = {c: [] for c in cases}
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(c):
        what, kind, u, nt, g = c
        if what == "encode":
            devutil.set_tuning("encode", u)
            return timed(lambda: cn.n_to_bits_dev(d_in, out=d_packed), args.iters)
        if what == "decode":
            devutil.set_tuning("decode", u)
            return timed(lambda: cn.bits_to_n_dev(d_packed, n, out=d_out), args.iters)
        a, b = ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_out.data_ptr())

        def f():
            rc = probes.probe_run(kind, u, nt, a, b, n, g, stream())
            assert rc == 0, rc
        return timed(f, args.iters)

    for c in cases:  # warm-up + correctness of every codec variant
        run(c)
        if c[0] == "encode":
            assert devutil.checksum_words(d_packed) == ref_sum, c
        elif c[0] == "decode":
            assert devutil.count_mismatch(d_in, d_out) == 0, c
    devutil.set_tuning("encode", 0)
    devutil.set_tuning("decode", 0)
    devutil.fill_random_acgt(d_in, 0x5EED)  # probes scribbled on nothing we need, but be safe
    cn.n_to_bits_dev(d_in, out=d_packed)
    for _ in range(args.rounds):
        for c in cases:
            times[c].append(run(c))

    pname = {0: "read", 1: "copy", 2: "r4w1", 3: "r1w4", 4: "write"}
    pbytes = {0: 1.0, 1: 2.0, 2: 1.25, 3: 1.25, 4: 1.0}
    rows = []
    for c in cases:
        what, kind, u, nt, g = c
        med, mn = statistics.median(times[c]), min(times[c])
        bpn = 1.25 if what != "probe" else pbytes[kind]
        if what == "probe":
            label = "%s u=%d nt=%d grid=%d" % (pname[kind], u, nt, g)
        else:
            label = "v%-2d %s" % (u, kind)
        rows.append({
            "what": what, "kernel": label, "variant": u if what != "probe" else None,
            "ms_median": round(med, 4), "ms_min": round(mn, 4),
            "gnts_median": round(n / med / 1e6, 1), "total_GBs_median": round(bpn * n / med / 1e6, 1),
            "total_GBs_best": round(bpn * n / mn / 1e6, 1),
        })
    rows.sort(key=lambda r: (r["what"], r["ms_median"]))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"log2_nt": args.log2_nt, "rounds": args.rounds, "iters": args.iters, "rows": rows}, open(args.out, "w"), indent=1)
    for r in rows:
        print("%-7s %-66s %8.4f ms (min %8.4f)  %8.1f Gnt/s  %7.1f GB/s" % (
            r["what"], r["kernel"], r["ms_median"], r["ms_min"], r["gnts_median"], r["total_GBs_median"]))


if __name__ == "__main__":
    main()
