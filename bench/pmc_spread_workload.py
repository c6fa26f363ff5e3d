#!/usr/bin/env python3
"""Workload for bench/pmc_spread.sh: 50 back-to-back bits_to_n decodes of the same 2^34-nt packed buffer, alternating
between two output buffers A / B (so that a duration that follows the BUFFER is told apart from one that follows TIME),
after one encode.  Run bare it prints every launch's HIP-event duration (events between the launches); under
`rocprofv3 --kernel-trace --pmc ...` the per-dispatch counters (GRBM_GUI_ACTIVE, TCC_CYCLE, credit stalls) are
what bench/parse_spread.py correlates with the per-dispatch durations."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2-nt", type=int, default=34)
ap.add_argument("--launches", type=int, default=50)
ap.add_argument("--events", action="store_true", help="time every launch with HIP events and print a JSON line")
ap.add_argument("--gap-us", type=float, default=0.0, help="host sleep between launches (0 = queued back to back)")
ap.add_argument("--kernel", default="decode", choices=("decode", "encode", "probe_r1w4", "probe_write", "probe_read", "probe_r4w1", "probe_copy"),
                help="what is launched 50 times: the codec kernels or bench/probes.hip's arithmetic-free streams of the same shapes")
a = ap.parse_args()
n = 1 << a.log2_nt
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_packed = torch.empty(n // 32, dtype=torch.int64, device="cuda")
outs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
devutil.fill_random_acgt(d_in, 0x5EED)
cn.n_to_bits_dev(d_in, out=d_packed)
for o in outs:  # warm-up: both outputs touched once
    cn.bits_to_n_dev(d_packed, n, out=o)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.launches + 1)] if a.events else None
import ctypes  # noqa: E402
import time  # noqa: E402

P = None
if a.kernel.startswith("probe"):
    P = ctypes.CDLL(os.path.join(ROOT, "bench", "libcnt_probes.so"))
    P.probe_shipped.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def launch(k):
    o = outs[k & 1]
    if a.kernel == "decode":
        cn.bits_to_n_dev(d_packed, n, out=o)
    elif a.kernel == "encode":
        cn.n_to_bits_dev(d_in, out=d_packed)
    else:  # kinds of probe_shipped: 0 read-only, 1 copy, 2 read4:write1, 3 read1:write4, 4 write-only; `bytes` = the wide side
        kind = {"probe_read": 0, "probe_copy": 1, "probe_r4w1": 2, "probe_r1w4": 3, "probe_write": 4}[a.kernel]
        src = d_packed if kind == 3 else d_in
        dst = d_packed if kind == 2 else o
        assert P.probe_shipped(kind, src.data_ptr(), dst.data_ptr(), n, stream) == 0


for k in range(a.launches):
    if ev:
        ev[k].record()
    launch(k)
    if a.gap_us:
        torch.cuda.synchronize()
        time.sleep(a.gap_us * 1e-6)
if ev:
    ev[a.launches].record()
torch.cuda.synchronize()
if a.kernel == "decode":
    assert devutil.count_mismatch(d_in, outs[0]) == 0 and devutil.count_mismatch(d_in, outs[1]) == 0
if ev:
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(a.launches)]
    print(json.dumps({"kernel": a.kernel, "what": "%s, 2^%d nt, %d launches, outputs alternate A/B, HIP events between launches" % (a.kernel, a.log2_nt, a.launches),
                      "gap_us": a.gap_us, "ms": [round(x, 4) for x in ms],
                      "ptr": {"A": hex(outs[0].data_ptr()), "B": hex(outs[1].data_ptr()), "packed": hex(d_packed.data_ptr())}}))
else:
    print("pmc spread workload ok")
