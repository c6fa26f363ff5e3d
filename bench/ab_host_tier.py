#!/usr/bin/env python3
"""Same-process A/B of the HOST tier of several builds of the library (cnt_n_to_bits / cnt_bits_to_n into reused outputs),
alternating call by call so that box, NUMA placement and page state are shared:
    python bench/ab_host_tier.py cur=cute_nucleotides_amd/libcute_nt_hip.so r03=bench/libcute_nt_hip_r03.so [--log2-nt 30]"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402,F401  (one HIP runtime per process, loaded first)

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--log2-nt", type=int, default=30)
ap.add_argument("--rounds", type=int, default=12)
a = ap.parse_args()
n_len = 1 << a.log2_nt
rng = np.random.default_rng(1)
n = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_len, dtype=np.uint8)]
libs = []
for spec in a.libs:
    name, path = spec.split("=")
    L = ctypes.CDLL(os.path.join(ROOT, path))
    L.cnt_n_to_bits.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    L.cnt_bits_to_n.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    libs.append((name, L))
bits = np.empty(n_len // 32, dtype=np.uint64)
back = np.empty(n_len, dtype=np.uint8)
p = lambda x: ctypes.c_void_p(x.ctypes.data)
for name, L in libs:
    assert L.cnt_n_to_bits(p(n), n_len, p(bits), bits.size) == 0 and L.cnt_bits_to_n(p(bits), bits.size, n_len, p(back)) == 0
assert np.array_equal(back, n)
t = {(name, op): [] for name, _ in libs for op in ("encode", "decode")}
for _ in range(a.rounds):
    for name, L in libs:
        t0 = time.perf_counter(); L.cnt_n_to_bits(p(n), n_len, p(bits), bits.size); t[(name, "encode")].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); L.cnt_bits_to_n(p(bits), bits.size, n_len, p(back)); t[(name, "decode")].append(time.perf_counter() - t0)
for (name, op), v in t.items():
    print(json.dumps({"lib": name, "op": op, "log2_nt": a.log2_nt, "ms_median": round(statistics.median(v) * 1e3, 3), "ms_min": round(min(v) * 1e3, 3), "ms_max": round(max(v) * 1e3, 3)}))
