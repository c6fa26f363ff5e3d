#!/usr/bin/env python3
"""bench/codec5_page_lab.hip timed beside the 5-letter codec's shipped kernels and their word-tiled arithmetic-free twins
(bench/probes.hip probe_codec5 kinds 2 / 4): isolated launches over 2^34 nt, median of 7 each, one JSON line per row."""
import ctypes
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "bench", "libcodec5_page_lab.so")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", SO,
                           os.path.join(ROOT, "bench", "codec5_page_lab.hip")])


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil

    L = ctypes.CDLL(SO)
    L.page_probe.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    P = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
    P.probe_codec5.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    words = ((1 << 34) // 27) // 128 * 128
    n5 = 27 * words
    pages = n5 // 4096
    d_n = torch.empty(n5 + 4096, dtype=torch.uint8, device="cuda")
    d_w = torch.empty(words + 512, dtype=torch.int64, device="cuda")
    devutil.fill_random_acgtn(d_n[:n5], 0x5EED)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bytes_moved = n5 + 8 * words

    def timed(fn, reps=7):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        return statistics.median(ms), min(ms)

    def row(name, fn, **kw):
        med, lo = timed(fn)
        print(json.dumps(dict(row=name, ms=round(med, 4), min_ms=round(lo, 4), frac=round(bytes_moved / med / 1e-3 / 8e12, 4), **kw)), flush=True)

    def probe5(kind, a, b):
        assert P.probe_codec5(kind, a, b, words, s) == 0

    def page(d, m, g, cap, a, b):
        assert L.page_probe(d, m, g, cap, a, b, pages, s) == 0

    row("shipped encode2", lambda: cn.n_to_bits2_dev(d_n[:n5], out=d_w[:words]))
    row("word tiles, no arithmetic, no LDS (encode)", lambda: probe5(2, d_n.data_ptr(), d_w.data_ptr()))
    for m in (1, 4, 8):
        for g in (8, 64, 128):
            for cap in (15, 20, 23):
                row("page tiles (encode)", lambda: page(0, m, g, cap, d_n.data_ptr(), d_w.data_ptr()), map=m, grain=g, cap=cap)
    cn.n_to_bits2_dev(d_n[:n5], out=d_w[:words])
    d_b = torch.empty(n5 + 4096, dtype=torch.uint8, device="cuda")
    row("shipped decode2", lambda: cn.bits_to_n2_dev(d_w[:words], n5, out=d_b[:n5]))
    row("word tiles, no arithmetic, no LDS (decode)", lambda: probe5(4, d_w.data_ptr(), d_b.data_ptr()))
    for m in (1, 4, 8):
        for g in (8, 64, 128):
            for cap in (6, 8, 10, 12, 14, 16):
                row("page tiles (decode)", lambda: page(1, m, g, cap, d_w.data_ptr(), d_b.data_ptr()), map=m, grain=g, cap=cap)
