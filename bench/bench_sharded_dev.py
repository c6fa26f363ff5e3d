#!/usr/bin/env python3
"""Single-process multi-GPU encode/decode through the device-resident sharded tier (cnt_*_sharded_dev): shard k
of the global buffer lives on device k, the calling thread enqueues every shard on its device's stream and waits
for all of them -- no torchrun, no process group, no host staging, no collective.  This is BASELINE.json
configs[4] ("256 GiB encode chunk-sharded across 8 x MI355X, per-GPU and aggregate Gnt/s") from ONE process:

    python bench/bench_sharded_dev.py                       # all visible devices, 2^35 nt each (32 GiB)
    python bench/bench_sharded_dev.py --ndev 8 --log2-nt 35 # 8 x 32 GiB = 256 GiB
    python bench/bench_sharded_dev.py --ndev 4 --log2-nt 30 --alias   # 1-GPU box: code path only (cnt_test_alias_devices)

Prints one JSON line: per-shard device milliseconds (HIP events on each device's stream), per-GPU and aggregate
Gnt/s (aggregate = all nucleotides / wall time of the call), the partition (cnt_shard_range) and each device's
PCI address.  (bench.py --gpus N, one process per GPU, is the driver's contract; this is the same measurement
without the launcher.)"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402,F401
from cute_nucleotides_amd import _lib, devutil, sharding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ndev", type=int, default=0, help="shards (0 = all visible devices)")
ap.add_argument("--log2-nt", type=int, default=35, help="nucleotides per shard = 2^k")
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
ap.add_argument("--decode", action="store_true", help="also time the decode of the shards")
ap.add_argument("--alias", action="store_true", help="TEST SUPPORT: fold more shards than visible devices onto the devices that exist")
a = ap.parse_args()
count = torch.cuda.device_count()
ndev = a.ndev or count
alias = a.alias
if ndev > count and not alias:
    raise SystemExit("%d shards need %d devices (%d visible); --alias folds them for a code-path run" % (ndev, ndev, count))
if alias:  # the switch exists in the test-hooks build only (tests/libcute_nt_hip_hooks.so); the product runs shard k on device k
    _lib.use_build("hooks")
    sharding.alias_devices(True)
n_global = ndev << a.log2_nt
parts = [sharding.shard_range_c(n_global, ndev, k) for k in range(ndev)]
shards, outs = [], []
for k, (lo, hi) in enumerate(parts):
    dev = torch.device("cuda", k % count)
    t = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
    devutil.fill_random_acgt(t, a.seed, first_nt=lo)
    shards.append(t)
    outs.append(torch.empty((hi - lo + 31) // 32, dtype=torch.int64, device=dev))
sharding.n_to_bits_sharded_dev(shards, outs=outs)  # warm-up (creates the per-device streams)
walls, per = [], []
for _ in range(a.iters):
    t0 = time.perf_counter()
    _, ms = sharding.n_to_bits_sharded_dev(shards, outs=outs, want_ms=True)
    walls.append(time.perf_counter() - t0)
    per.append(ms)
wall = statistics.median(walls)
shard_ms = [statistics.median(p[k] for p in per) for k in range(ndev)]
line = {
    "what": "n_to_bits encode, device-resident shards, one process (cnt_n_to_bits_sharded_dev)", "ndev": ndev, "visible_devices": count,
    "alias_test_hook": alias, "nt_per_shard": 1 << a.log2_nt, "total_GiB": round(n_global / 2**30, 1),
    "partition": parts, "devices": [devutil.device_identity(k % count) for k in range(ndev)],
    "shard_ms": [round(x, 4) for x in shard_ms], "per_gpu_gnts": [round((hi - lo) / (m * 1e-3) / 1e9, 1) for (lo, hi), m in zip(parts, shard_ms)],
    "per_gpu_frac_of_8TBs": [round(1.25 * (hi - lo) / (m * 1e-3) / 8e12, 4) for (lo, hi), m in zip(parts, shard_ms)],
    "wall_ms": round(wall * 1e3, 4), "aggregate_gnts": round(n_global / wall / 1e9, 1), "data_path_collective": None,
}
# every shard round-trips, and the shards' words are the whole buffer's words (checksums are position-salted)
backs = sharding.bits_to_n_sharded_dev(outs, [hi - lo for lo, hi in parts])
line["verified"] = all(devutil.count_mismatch(s, b) == 0 for s, b in zip(shards, backs))
if a.decode:
    walls = []
    for _ in range(a.iters):
        t0 = time.perf_counter()
        sharding.bits_to_n_sharded_dev(outs, [hi - lo for lo, hi in parts], outs=backs)
        walls.append(time.perf_counter() - t0)
    line["decode_wall_ms"] = round(statistics.median(walls) * 1e3, 4)
    line["decode_aggregate_gnts"] = round(n_global / statistics.median(walls) / 1e9, 1)
print(json.dumps(line), flush=True)
