"""Small-input device tier: direct launches vs one HIP-graph replay (encode + decode of 40 000 nt).
usage (GPU box): python bench/graph_lab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library


for small, n_len in ((0, 40000), (1 << 17, 40000), (0, 100000), (1 << 17, 100000), (0, 1 << 20)):
    devutil.set_tuning("small_nt", small)
    d_in = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d_in, 3)
    d_pk = torch.empty((n_len + 31) // 32, dtype=torch.int64, device="cuda")
    d_out = torch.empty(n_len, dtype=torch.uint8, device="cuda")

    def pair():
        cn.n_to_bits_dev(d_in, out=d_pk)
        cn.bits_to_n_dev(d_pk, n_len, out=d_out)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        pair()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            pair()
    for name, fn, per in (("direct launches", pair, 1), ("graph of 10 pairs", g.replay, 10)):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 2000 // per
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / (reps * per)
        print("%8d nt  small_nt=%-7d %-18s %.2f us per encode+decode pair" % (n_len, small, name, dt * 1e6), flush=True)
    assert devutil.count_mismatch(d_in, d_out) == 0
