#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement on MI355X.

Metric (BASELINE.json): Gnt/s encode+decode on 16 GiB random ACGT; fraction of the HBM
roofline.  One "step" = one pass of the hot path over one batch: n_to_bits encode of the
device-resident 16 GiB ASCII buffer (-> 4 GiB packed) followed by bits_to_n decode of the
packed words (-> 16 GiB ASCII).  Inputs are generated on the device (counter-based ACGT
generator, seed in the JSON) and are resident in HBM before the timed region starts.

    python bench.py [--steps K --warmup W]                              (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (N>1, one rank per GPU: the driver's form)
    python bench.py --gpus N --steps K --warmup W                       (N>1 WITHOUT a launcher: ONE process drives the
        N devices through the enqueue-only sharded tier -- cnt_*_sharded_dev_enqueue queues all K steps on one library
        stream per shard, ONE cnt_sharded_dev_wait; no process group at all -- and prints the same JSON line plus
        `scaling_overhead_us`; what north_star describes; bench/bench_single_process.py)

N>1: the global buffer (N x per-GPU size) is cut into contiguous chunks on word boundaries,
rank r owns chunk r (cute_nucleotides_amd/sharding.py); there is no data-path collective --
the process group (gloo over loopback by default, RCCL with CNT_BENCH_BACKEND=nccl) is only used
for the barrier and the max-over-ranks of the timing.  Weak scaling: per-GPU work is fixed.

The per-GPU size stays 2^34 nt (the metric's 16 GiB) at every N, so the driver's N = 1, 2, 4, 8 values form a
weak-scaling curve; BASELINE.json configs[4] (256 GiB over 8 GPUs = 2^35 nt per GPU) is measured in the same run
as `configs4_shard` (encode of a 2^35-nt shard on every rank, per-GPU and aggregate Gnt/s).  `ranks` lists every
rank's device (index, PCI address, UUID, NUMA node, visible-device count) and its own kernel times, so an N-GPU
line shows N distinct devices and no data-path collective.

Rank 0 prints ONE JSON line.  `value` counts every nucleotide converted per second over all
ranks (N encoded + N decoded per step and rank), inputs already in HBM.  `roofline` is for
the n_to_bits encode kernel (the north-star target), `roofline_decode` for bits_to_n; both
use ALGORITHMIC bytes (1.25 B/nt: encode 1 read + 0.25 written, decode 0.25 read + 1
written) over the kernel's average duration measured live with HIP events on the
launch stream; `roofline.traffic` = HBM bytes per launch measured by this run itself (two child `rocprofv3 --pmc`
passes, FETCH_SIZE and WRITE_SIZE separately, calibrated on known-size probes; falls back to the committed
profiles/hbm_traffic.json, and says so in `traffic_source`).  `cpu_baseline` times the oracle's ports of the reference's fastest AVX2
paths (n_to_bits_movemask / bits_to_n_shuffle) on this box's host cores: rank 0, at EVERY N (north_star: "next to the
reference's own AVX2/BMI2 path timed on the GPU box's host cores in the same run", at 1, 2, 4 and 8 GPUs), after all GPU work of
the line -- the host threads never compete with a rank that is still enqueueing.  At N > 1 the reference-faithful
single-thread tables are left to the N = 1 line (`--cpu-seconds` bounds the leg either way).
"""
import argparse
import ctypes
import glob
import json
import os
import shutil
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "bench")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Everything that is not the contract itself lives in bench/ (VERDICT r03 next-8): bench_measure.py = statistics, HIP-event
# timing loops and the extra blocks of the N = 1 line; bench_single_process.py = `--gpus N` without a launcher.  This file
# keeps the arguments, the CPU-baseline leg (the only code allowed to touch oracle/), the timed region and the JSON line.
from bench_measure import (BYTES_PER_NT, HBM_PEAK_GBS, HOST_TIER_LOG2, crossover, device_row, first_contact_devices, gbs, gpu_numa_node,  # noqa: E402,F401
                           host_placement, load_probes, measure_ceilings, measure_codec5, host_tier_block, measure_config3_64gib,
                           measure_configs_1gib, measure_host_tier, measure_packed_ops, measure_pcie, measure_traffic_live, numpy_pack,
                           physical_cores, require_free_hbm, sockets, stats_ms, timed_calls, timed_queued, traffic_for_line)

def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--log2-nt", type=int, default=34, help="per-GPU nucleotides = 2^k (default 34 = 16 GiB, the metric size)")
    p.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    p.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU-baseline time budget (0 = skip)")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="headline only: skip the ceilings, the configs block and the fused pass")
    p.add_argument("--no-live-traffic", action="store_true", help="do not spawn the two rocprofv3 --pmc child runs; quote profiles/hbm_traffic.json")
    p.add_argument("--shard-log2-nt", type=int, default=35, help="per-GPU shard of BASELINE.json configs[4] (256 GiB over 8 GPUs = 2^35 nt each)")
    return p.parse_args()


def cpu_baseline(seconds, faithful_tables=True):
    """Time the oracle's x86 ports of the reference's best encoder/decoder on the host cores.
    faithful_tables=False (the N > 1 lines) keeps the all-core and one-thread figures and skips the reference-harness
    tables, which belong to -- and are on -- the N = 1 line.
    The oracle is used here only as the timed CPU baseline (kind 'port': the reference is Rust and
    cannot be built in this image).  Persistent threads: each owns one contiguous chunk and
    re-runs it until a common deadline (ctypes releases the GIL during the C call)."""
    import threading

    import numpy as np

    from oracle import cnt_oracle as orc

    orc.build()
    if not orc.port_cpu_ok():
        return {"value": None, "unit": "Gnt/s", "cores": 0, "kind": "port", "sample": "host CPU lacks AVX2/BMI2"}
    L = orc.lib()
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = len(cpus)  # logical CPUs = timed threads
    physical = physical_cores(cpus)
    per_thread = 16 << 20  # 16 Mi nt per thread: past any per-core share of L2/L3
    n_len = max(1 << 28, cores * per_thread)
    n = np.empty(n_len, dtype=np.uint8)
    bits = np.empty(n_len // 32, dtype=np.uint64)
    out = orc._aligned_u8(n_len)

    def spans_for(threads):
        per = (n_len // threads) // 32 * 32
        return [(k * per, n_len if k == threads - 1 else (k + 1) * per) for k in range(threads)]

    def parallel(fn, threads, budget):
        """every thread repeats fn(lo, hi) on its span until `budget` s have passed; returns nt/s"""
        spans = spans_for(threads)
        counts = [0] * threads
        gate = threading.Barrier(threads + 1)
        deadline = [0.0]

        def body(k):
            try:  # pin: first touch (the fill pass) and every later pass of span k stay on one CPU / NUMA node
                os.sched_setaffinity(0, {cpus[k % len(cpus)]})
            except (AttributeError, OSError):
                pass
            gate.wait()
            lo, hi = spans[k]
            while True:
                fn(lo, hi)
                counts[k] += 1
                if time.perf_counter() >= deadline[0]:
                    break

        ts = [threading.Thread(target=body, args=(k,)) for k in range(threads)]
        [t.start() for t in ts]
        t0 = time.perf_counter()
        deadline[0] = t0 + budget
        gate.wait()
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        return sum(c * (hi - lo) for c, (lo, hi) in zip(counts, spans)) / dt / 1e9

    def fill(lo, hi):
        L.cnt_oracle_fill_random_acgt(n.ctypes.data + lo, lo, hi - lo, 0x5EED)

    def enc(lo, hi):
        L.cnt_port_n_to_bits_movemask(n.ctypes.data + lo, hi - lo, bits.ctypes.data + (lo // 32) * 8, (hi - lo + 31) // 32)

    def dec(lo, hi):
        L.cnt_port_bits_to_n_shuffle(bits.ctypes.data + (lo // 32) * 8, (hi - lo + 31) // 32, hi - lo, out.ctypes.data + lo)

    parallel(fill, cores, 0.0)  # generate the sample (one pass, all cores); also first-touches n
    parallel(enc, cores, 0.0)   # untimed: first-touch bits / out
    parallel(dec, cores, 0.0)
    q = max(seconds / 5.0, 0.5)
    # the all-core figure moved +-16 % between rounds on the same CPU model (150 ... 208 Gnt/s) -- one window
    # of 256 threads is at the mercy of whatever else the host runs in it.  Three interleaved windows per direction, the MEDIAN on
    # the line and all three beside it.
    enc_runs, dec_runs = [], []
    for _ in range(3):
        enc_runs.append(parallel(enc, cores, q / 3.0))
        dec_runs.append(parallel(dec, cores, q / 3.0))
    encN, decN = sorted(enc_runs)[1], sorted(dec_runs)[1]
    enc1, dec1 = parallel(enc, 1, q), parallel(dec, 1, q)
    assert bytes(out[: 1 << 16]) == bytes(n[: 1 << 16])  # the timed decode really round-trips
    lut_nt = 1 << 24
    t0 = time.perf_counter()
    L.cnt_oracle_n_to_bits_lut(n.ctypes.data, lut_nt, bits.ctypes.data, lut_nt // 32)
    lut_enc = lut_nt / (time.perf_counter() - t0) / 1e9
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    both = lambda e, d: 2.0 / (1.0 / e + 1.0 / d)  # nt converted per second over an encode pass + a decode pass
    summary = {
        "value": round(both(encN, decN), 3), "unit": "Gnt/s", "cores": cores, "kind": "port",
        "cores_detail": {"threads_timed": cores, "logical_cpus": cores, "physical_cores": physical,
                         "sockets": sockets(), "note": "`cores` = timed threads = logical CPUs (SMT siblings included)"},
        "sample": "%d Mi random ACGT nt (%d pinned threads x >=16 Mi contiguous nt each, re-run until ~%.0f s total); "
                  "n_to_bits_movemask + bits_to_n_shuffle ports (oracle/cnt_simd_port.c), output preallocated"
                  % (n_len >> 20, cores, seconds),
        "encode_gnts": round(encN, 3), "decode_gnts": round(decN, 3),
        "all_core_windows": {"encode_gnts": [round(x, 1) for x in enc_runs], "decode_gnts": [round(x, 1) for x in dec_runs],
                             "note": "three interleaved timing windows per direction; `encode_gnts` / `decode_gnts` / `value` use the medians"},
        "one_thread": {"encode_gnts": round(enc1, 3), "decode_gnts": round(dec1, 3), "value": round(both(enc1, dec1), 3),
                       "note": "one thread over the whole sample"},
        "scalar_lut_encode_gnts_1thread": round(lut_enc, 3), "cpu_model": model,
        "reference_toolchain": {"cargo": shutil.which("cargo"), "rustc": shutil.which("rustc"),
                                "note": "kind is 'port' because the reference (Rust) cannot be built where no cargo/rustc exists"},
    }
    if not faithful_tables:
        summary["reference_faithful_tables"] = "on the N = 1 line (reference_faithful_40k_GiBs, reference_faithful_GiBs_1thread_alloc_inclusive)"
        return summary
    # reference-faithful rows: the reference's own bench input ("ATCG" x 10000, cache-resident), one
    # thread, output allocated inside every timed call (benches/bench_n_to_bits.rs:6-13) -- GiB/s of
    # nucleotides like criterion prints them, to sit beside README.md:344-366 (i9-9880H)
    n40 = np.frombuffer(b"ATCG" * 10000, dtype=np.uint8).copy()
    b40 = np.empty(1250, dtype=np.uint64)
    L.cnt_oracle_n_to_bits_lut(n40.ctypes.data, 40000, b40.ctypes.data, 1250)
    names = {0: "n_to_bits_lut", 1: "n_to_bits_pext", 2: "n_to_bits_shift", 3: "n_to_bits_movemask", 4: "n_to_bits_mul",
             5: "memcpy", 10: "bits_to_n_lut", 11: "bits_to_n_shuffle", 12: "bits_to_n_pdep", 13: "bits_to_n_clmul"}
    faithful = {}
    for fn, name in names.items():
        src = n40 if fn < 10 else b40
        L.cnt_port_time_alloc_inclusive(fn, src.ctypes.data, 40000, 2000)  # warm-up
        per = L.cnt_port_time_alloc_inclusive(fn, src.ctypes.data, 40000, 20000 if fn not in (0, 10) else 4000)
        faithful[name] = round(40000 / per / 2**30, 3)
    # the 5-letter groups of the same harness (bench_n_to_bits.rs:26-35,53-63: "ATCGN" x 8000), README.md:374-377
    n5 = np.frombuffer(b"ATCGN" * 8000, dtype=np.uint8).copy()
    b5 = np.empty(L.cnt_oracle_words2_for(40000), dtype=np.uint64)
    L.cnt_oracle_n_to_bits2_lut(n5.ctypes.data, 40000, b5.ctypes.data, b5.size)
    for fn, name in {20: "n_to_bits2_lut", 21: "n_to_bits2_pext", 30: "bits_to_n2_lut", 31: "bits_to_n2_pdep"}.items():
        src = n5 if fn < 30 else b5
        L.cnt_port_time_alloc_inclusive(fn, src.ctypes.data, 40000, 1000)
        per = L.cnt_port_time_alloc_inclusive(fn, src.ctypes.data, 40000, 10000 if fn in (21, 31) else 3000)
        faithful[name] = round(40000 / per / 2**30, 3)
    # the same allocation-inclusive single-thread calls at 1 MiB and 1 GiB of random ACGT
    # (DRAM-resident at 1 GiB; every call pays the page faults of its fresh output, like a fresh Vec)
    faithful_big = {}
    for log2 in HOST_TIER_LOG2:
        m = 1 << log2
        if m > n_len:
            continue
        L.cnt_port_n_to_bits_movemask(n.ctypes.data, m, bits.ctypes.data, m // 32)
        row = {}
        every = log2 in (20, 30)  # all ten functions at the two sizes SURVEY 8d names; the fastest pair + memcpy elsewhere
        for fn, name in names.items():
            if not every and fn not in (3, 5, 11):
                continue
            src = n if fn < 10 else bits
            est = m / (1.3e9 if fn in (0, 10) else 15e9)
            iters = max(1, min(20000, int(0.08 / est)))
            if log2 <= 24:
                L.cnt_port_time_alloc_inclusive(fn, src.ctypes.data, m, max(1, iters // 10))
            per = L.cnt_port_time_alloc_inclusive(fn, src.ctypes.data, m, iters)
            row[name] = round(m / per / 2**30, 3)
        faithful_big["2^%d" % log2] = row
    return dict({"reference_faithful_40k_GiBs": faithful, "reference_faithful_GiBs_1thread_alloc_inclusive": faithful_big}, **summary)


def main():
    args = parse_args()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            from bench_single_process import main_single_process

            return main_single_process(args, cpu_baseline)  # no launcher: one process drives the N devices (cnt_*_sharded_dev_enqueue)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Test hooks (never set by the driver): CNT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # CNT_BENCH_BACKEND=gloo swaps the process group's backend, so the N>1 code path can be
    # exercised on a 1-GPU box (RCCL refuses two ranks on one device).
    shared_gpu = os.environ.get("CNT_BENCH_SHARE_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    # The process group carries no data: it is the barrier + a MAX/MIN of a few scalars + one gather of the
    # per-rank report rows.  gloo over loopback is the default because it needs no GPU IPC and has been exercised
    # with 2 ranks on the GPU box (tests/test_gpu_bench.py); CNT_BENCH_BACKEND=nccl runs the same control plane over RCCL.
    backend = os.environ.get("CNT_BENCH_BACKEND", "gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the scalar reductions live

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on STDOUT from C++ ("[Gloo] Rank 0 is connected to ..."); the
        # contract is ONE JSON line there, so the process-level stdout points at stderr while the group
        # comes up (and for the whole run on ranks > 0, which print nothing themselves)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        def bring_up(name):
            if name == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
            else:
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                dist.init_process_group(backend=name)
            dist.barrier()  # first collective: lazy connection chatter happens here, not later

        try:
            try:
                bring_up(backend)
            except Exception as exc:  # the control plane must not be what stops a scaling run: try the other backend once
                other = "nccl" if backend != "nccl" else "gloo"
                print("bench.py: process group over %s failed (%s); trying %s" % (backend, str(exc)[:200], other), file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = other
                red_dev = dev if backend == "nccl" else torch.device("cpu")
                bring_up(backend)
        finally:
            sys.stdout.flush()
            if rank == 0:
                os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil, sharding

    # ---- who am I: the device this rank drives ------------------------------------------------------
    from cute_nucleotides_amd import _lib

    ident = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
    ident.update(device_row(torch, devutil, _lib.lib(), local_rank))  # index, PCI address, NUMA node, name, UUID, HBM, cnt_chip_info

    if world > 1:  # first contact, BEFORE anything is allocated: N ranks on N distinct devices (or labelled as folded), else stop
        early = [None] * world
        dist.all_gather_object(early, ident)
        if rank == 0:
            first_contact_devices(early, world, shared_gpu)

    n_per = 1 << args.log2_nt
    # first contact: enough free HBM on THIS rank's device before anything is allocated (ranks folded
    # onto one GPU by the test hook share it)
    ident["hbm_before"] = require_free_hbm(torch, local_rank, (world if shared_gpu else 1) * 2.25 * n_per + (1 << 30),
                                           "the timed buffers (2^%d nt in, packed, out%s)" % (args.log2_nt, " x %d folded ranks" % world if shared_gpu and world > 1 else ""))
    n_global = n_per * world
    lo, hi = sharding.partition(n_global, world)[rank]  # contiguous chunk on a word boundary
    assert (lo, hi) == sharding.shard_range_c(n_global, world, rank)  # the C library cuts at the same places
    n_len = hi - lo
    words = cn.n_to_bits.words_for(n_len)

    d_in = torch.empty(n_len, dtype=torch.uint8, device=dev)
    d_packed = torch.empty(words, dtype=torch.int64, device=dev)
    d_out = torch.empty(n_len, dtype=torch.uint8, device=dev)
    devutil.fill_random_acgt(d_in, args.seed, first_nt=lo)
    torch.cuda.synchronize()

    def step():
        cn.n_to_bits_dev(d_in, out=d_packed)
        cn.bits_to_n_dev(d_packed, n_len, out=d_out)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def allreduce(values, op):
        t = torch.tensor(values, dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=op)
        return [float(x) for x in t.tolist()]

    for _ in range(args.warmup):
        step()

    # per-kernel HIP events on the launch stream (torch's current stream == the stream the
    # C ABI is handed), recorded inside the timed region
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        cn.n_to_bits_dev(d_in, out=d_packed)
        ev[k][1].record()
        cn.bits_to_n_dev(d_packed, n_len, out=d_out)
        ev[k][2].record()
    barrier()
    elapsed = time.perf_counter() - t0

    elapsed = allreduce([elapsed], dist.ReduceOp.MAX)[0]
    enc_list = [e[0].elapsed_time(e[1]) for e in ev]
    dec_list = [e[1].elapsed_time(e[2]) for e in ev]
    enc_ms, dec_ms = statistics.fmean(enc_list), statistics.fmean(dec_list)
    # slowest rank's per-kernel averages (for the aggregate encode / decode rates); each rank's own times go
    # into its `ranks` row, rank 0's into `roofline*`, which describe one GPU
    enc_ms_max, dec_ms_max = allreduce([enc_ms, dec_ms], dist.ReduceOp.MAX)

    # ---- verification of the TIMED buffers, before anything else writes them --------------------------
    # Outside the timed region, and WITHOUT the oracle (bench.py may touch oracle/ only in its cpu_baseline
    # leg; bit-exact parity against the oracle is what tests/ -m gpu establish):
    #  (1) the timed buffers round-trip: decode(encode(x)) == x, compared on the device;
    #  (2) the first and last 2^20 nt of the timed packed buffer equal a numpy restatement of the layout;
    #  (3) the reference's periodic bench input (benches/bench_n_to_bits.rs:68-70, "ATCG" repeated) at 2^20 nt --
    #      a whole number of tiles and past the small-input path, so it runs the SAME stream kernels as the
    #      timed region -- encodes to 0xD8D8D8D8D8D8D8D8 words (the reference's unit-test vector,
    #      n_to_bits.rs:414-415) and decodes back.
    verified, packed_sum = None, None
    if not args.no_verify:
        ok = devutil.count_mismatch(d_in, d_out) == 0
        packed_sum = devutil.checksum_words(d_packed)
        m = min(1 << 20, n_len // 32 * 32)
        for off in (0, (n_len - m) // 32 * 32):
            host_n = d_in[off : off + m].cpu().numpy()
            got = d_packed[off // 32 : (off + m) // 32].cpu().numpy().view(np.uint64)
            ok = ok and bool(np.array_equal(got, numpy_pack(host_n)))
        kat = torch.from_numpy(np.frombuffer(b"ATCG" * (1 << 18), dtype=np.uint8).copy()).to(dev)
        assert kat.numel() % 4096 == 0 and kat.numel() > devutil.get_tuning("small_nt")
        kat_bits = cn.n_to_bits_dev(kat)
        ok = ok and bool((kat_bits.cpu().numpy().view(np.uint64) == np.uint64(0xD8D8D8D8D8D8D8D8)).all()) and kat_bits.numel() == 1 << 15
        ok = ok and bool(torch.equal(cn.bits_to_n_dev(kat_bits, kat.numel()), kat))
        del kat, kat_bits
        verified = ok

    extras = not args.no_extras
    # ---- BASELINE.json configs[3] in passing: the FUSED round trip over the same buffers -------------------
    fused = None
    if extras:
        fl = timed_calls(torch, lambda: cn.round_trip_dev(d_in, out_bits=d_packed, out_n=d_out), 5)
        fused = stats_ms(fl)
        if verified is not None:  # the fused kernel's own outputs: same packed words, same decoded text
            fused["verified"] = devutil.count_mismatch(d_in, d_out) == 0 and devutil.checksum_words(d_packed) == packed_sum
            verified = verified and fused["verified"]
        if n_len >= (1 << 22):  # the same call with all three pointers off the 128-B grid: one launch as well (round_trip_window)
            m = n_len - 4096
            v_in, v_pk, v_out = d_in[5 : 5 + m], d_packed[1 : 1 + (m + 31) // 32], d_out[77 : 77 + m]
            fused["off_grid"] = stats_ms(timed_calls(torch, lambda: cn.round_trip_dev(v_in, out_bits=v_pk, out_n=v_out), 5))
            fused["off_grid"]["nt"] = m
            if verified is not None:
                k = min(m, 1 << 28)
                fused["off_grid"]["verified"] = bool(torch.equal(v_out[:k], v_in[:k]) and torch.equal(v_out[-k:], v_in[-k:]))
                verified = verified and fused["off_grid"]["verified"]

    # ---- BASELINE.json configs[1] / [2]: 1 GiB, and a ragged size, on the first 2^30 nt of the same buffers (bench_measure.py) ----
    configs = measure_configs_1gib(torch, d_in, d_packed, d_out, verified is not None) if extras and n_len >= (1 << 30) else {}

    # ---- same-run, same-box ceilings: no-arithmetic streams issued like the shipped kernels -------
    ceilings = None
    if extras:
        P = load_probes()
        if P is not None and n_len % 16384 == 0 and n_len <= (1 << 35):
            ceilings = measure_ceilings(torch, P, d_in, d_out, n_len)
        else:
            ceilings = {"error": "bench/libcnt_probes.so not built (run __graft_entry__.build())" if P is None else "size not probe-able"}

    # ---- BASELINE.json configs[4]: this rank's shard of "256 GiB over 8 GPUs" = 2^35 nt, encode ---------------
    shard = None
    del d_out, d_packed, d_in
    torch.cuda.empty_cache()
    if extras and args.shard_log2_nt > 0:
        s_len = 1 << args.shard_log2_nt
        free, _ = torch.cuda.mem_get_info()
        need = 2.3 * s_len + (4 << 30)
        fits = allreduce([1.0 if free > need else 0.0], dist.ReduceOp.MIN)[0] == 1.0  # every rank takes the same branch
        if fits:
            s_lo, s_hi = sharding.shard_range_c(s_len * world, world, rank)
            s_in = torch.empty(s_hi - s_lo, dtype=torch.uint8, device=dev)
            s_packed = torch.empty((s_hi - s_lo) // 32, dtype=torch.int64, device=dev)
            devutil.fill_random_acgt(s_in, args.seed + 4, first_nt=s_lo)
            sl = timed_calls(torch, lambda: cn.n_to_bits_dev(s_in, out=s_packed), 8, warm=2)
            # all ranks at once, wall clock between barriers: what "aggregate" means for the sharded job
            reps = 8
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                cn.n_to_bits_dev(s_in, out=s_packed)
            barrier()
            wall = allreduce([time.perf_counter() - t0], dist.ReduceOp.MAX)[0]
            shard = {"nt": s_hi - s_lo, "first_nt": s_lo, "encode_ms": stats_ms(sl), "wall_ms_per_encode_all_ranks": round(wall / reps * 1e3, 4)}
            if verified is not None:
                s_back = cn.bits_to_n_dev(s_packed, s_hi - s_lo)
                shard["round_trip_verified"] = devutil.count_mismatch(s_in, s_back) == 0
                verified = verified and shard["round_trip_verified"]
                del s_back
            del s_in, s_packed
            torch.cuda.empty_cache()
        else:
            shard = {"skipped": "needs %.0f GiB of free HBM on every rank, %.0f free here" % (need / 2**30, free / 2**30)}

    # ---- BASELINE.json configs[3] at its own size: 64 GiB (2^36 nt), two passes and fused, N = 1 only (bench_measure.py) ----
    if extras and world == 1:
        rows3, ok3 = measure_config3_64gib(torch, dev, args.seed + 3, verified is not None)
        configs.update(rows3)
        if verified is not None and ok3 is not None:
            verified = verified and ok3

    # ---- SURVEY 8 f-1 / f-4 on the driver line (rank 0 at N = 1): 5-letter codec and packed-domain ops at the metric size ---
    codec5 = packed_ops = None
    if extras and world == 1:
        free, _ = torch.cuda.mem_get_info()
        if free > 2.6 * n_per + (4 << 30):
            codec5 = measure_codec5(torch, args.seed, args.log2_nt)
            packed_ops = measure_packed_ops(torch, args.seed, args.log2_nt)
            if verified is not None:
                verified = verified and codec5["verified"] and packed_ops["verified"]
        else:
            codec5 = packed_ops = {"skipped": "needs %.0f GiB of free HBM, %.0f GiB free" % ((2.6 * n_per + (4 << 30)) / 2**30, free / 2**30)}

    if verified is not None:
        verified = bool(allreduce([1.0 if verified else 0.0], dist.ReduceOp.MIN)[0] == 1.0)

    # ---- one report row per rank, gathered through the control plane (no data-path collective anywhere) ------
    enc_st, dec_st = stats_ms(enc_list), stats_ms(dec_list)
    row = dict(ident)
    row.update({
        "nt": n_len, "first_nt": lo,
        "encode_ms": enc_st, "decode_ms": dec_st,
        "encode_gnts": round(n_len / (enc_ms * 1e-3) / 1e9, 1), "decode_gnts": round(n_len / (dec_ms * 1e-3) / 1e9, 1),
        "encode_frac": round(gbs(BYTES_PER_NT * n_len, enc_ms) / HBM_PEAK_GBS, 4),
        "decode_frac": round(gbs(BYTES_PER_NT * n_len, dec_ms) / HBM_PEAK_GBS, 4),
        "encode_read_view_frac": round(gbs(n_len, enc_ms) / HBM_PEAK_GBS, 4),
        "fused_ms_median": fused["median"] if fused else None,
        "configs4_shard": shard,
    })
    rows = [row]
    if world > 1:
        rows = [None] * world
        dist.all_gather_object(rows, row)

    if rank == 0:
        nt_per_step = 2 * n_global  # N encoded + N decoded, all ranks
        value = nt_per_step * args.steps / elapsed / 1e9
        enc_gbs = gbs(BYTES_PER_NT * n_len, enc_ms)
        dec_gbs = gbs(BYTES_PER_NT * n_len, dec_ms)
        # every buffer of this process is free by now; at N > 1 the child passes see rank 0's device only
        traffic, traffic_source, live = traffic_for_line(args.log2_nt, world, extras and not args.no_live_traffic,
                                                         child_device=local_rank if world > 1 else None)
        span = lambda key: {"min": min(r[key] for r in rows), "max": max(r[key] for r in rows)}
        devices = first_contact_devices(rows, world, shared_gpu, control_plane=backend if world > 1 else None, processes=world)
        line = {
            "metric": "Gnt/s encode+decode on 16 GiB random ACGT; % HBM read roofline at 1/8 GPU",
            "value": round(value, 3), "unit": "Gnt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": "n_to_bits encode + bits_to_n decode of a device-resident uniform random ACGT buffer, "
                            "%.3g GiB (2^%d nt) per GPU" % (n_per / 2**30, args.log2_nt),
                "nt_per_gpu": n_per, "nt_per_step": nt_per_step, "seed": hex(args.seed),
                "sharding": ("contiguous chunks on word boundaries (cnt_shard_range), no data-path collective; control plane: " + backend)
                if world > 1 else "single GPU",
                "encode_kernel": dict(devutil.variants("encode"))[devutil.get_tuning("encode")],
                "decode_kernel": dict(devutil.variants("decode"))[devutil.get_tuning("decode")],
            },
            "encode_gnts_per_gpu": round(n_len / (enc_ms * 1e-3) / 1e9, 3),
            "decode_gnts_per_gpu": round(n_len / (dec_ms * 1e-3) / 1e9, 3),
            "encode_gnts_all_gpus": round(n_global / (enc_ms_max * 1e-3) / 1e9, 3),
            "decode_gnts_all_gpus": round(n_global / (dec_ms_max * 1e-3) / 1e9, 3),
            "hbm_read_roofline_frac_encode_per_gpu": round(n_per / (enc_ms_max * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": {
                "kernel": "n_to_bits (encode)", "bound": "hbm", "achieved": round(enc_gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(enc_gbs / HBM_PEAK_GBS, 4),
                "traffic": (traffic or {}).get("encode_bytes_per_launch"), "traffic_source": traffic_source,
                "avg_kernel_ms": round(enc_ms, 4), "kernel_ms": enc_st, "frac_at_median": round(gbs(BYTES_PER_NT * n_len, enc_st["median"]) / HBM_PEAK_GBS, 4),
                "frac_at_min": round(gbs(BYTES_PER_NT * n_len, enc_st["min"]) / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": int(BYTES_PER_NT * n_len),
                "read_only_view": {"achieved": round(gbs(n_len, enc_ms), 1), "frac": round(gbs(n_len, enc_ms) / HBM_PEAK_GBS, 4)},
            },
            "roofline_decode": {
                "kernel": "bits_to_n (decode)", "bound": "hbm", "achieved": round(dec_gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(dec_gbs / HBM_PEAK_GBS, 4),
                "traffic": (traffic or {}).get("decode_bytes_per_launch"), "traffic_source": traffic_source,
                "avg_kernel_ms": round(dec_ms, 4), "kernel_ms": dec_st, "frac_at_median": round(gbs(BYTES_PER_NT * n_len, dec_st["median"]) / HBM_PEAK_GBS, 4),
                "frac_at_min": round(gbs(BYTES_PER_NT * n_len, dec_st["min"]) / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": int(BYTES_PER_NT * n_len),
            },
            "traffic_live": live,
            "roofline_over_ranks": {"encode_frac": span("encode_frac"), "decode_frac": span("decode_frac"),
                                    "encode_read_view_frac": span("encode_read_view_frac")},
            "ranks": rows,
            "devices": devices,
            "verified": verified,
            "value_definition": "nucleotides converted per second over all ranks: each step encodes nt_per_gpu and decodes "
                                "nt_per_gpu on every rank (nt_per_step = 2 x n_gpus x nt_per_gpu); equals the harmonic mean "
                                "of the encode and decode rates",
        }
        if fused is not None:
            fgbs = gbs(2.25 * n_len, fused["median"])
            line["fused_round_trip"] = {
                "what": "cnt_round_trip_dev: one pass reads the ASCII and writes packed words + decoded ASCII "
                        "(BASELINE.json configs[3] at the metric size; its own 64 GiB size is in `configs`); measured after the "
                        "timed region, not part of value",
                "ms": fused["median"], "ms_stats": fused, "nt_converted_gnts": round(2 * n_len / (fused["median"] * 1e-3) / 1e9, 3),
                "bytes_per_nt": 2.25, "achieved": round(fgbs, 1), "unit": "GB/s", "frac": round(fgbs / HBM_PEAK_GBS, 4),
            }
            if "off_grid" in fused:  # d_n + 5 B, d_bits + 8 B, d_back + 77 B: still one launch
                og = fused["off_grid"]
                line["fused_round_trip"]["off_grid_frac"] = round(gbs(2.25 * og["nt"], og["median"]) / HBM_PEAK_GBS, 4)
                line["fused_round_trip"]["off_grid_vs_aligned"] = round((og["median"] / og["nt"]) / (fused["median"] / n_len), 4)
        if ceilings is not None:
            line["ceilings"] = {
                "what": "no-arithmetic streams issued exactly like the shipped kernels (bench/probes.hip probe_shipped), same process, "
                        "same buffers, 2^%d-byte wide side, median of 5 launches each; GB/s of all bytes moved" % args.log2_nt,
                "rank0": ceilings,
            }
            if "error" not in ceilings:
                c = ceilings
                line["ceilings"]["encode_vs"] = {
                    "total_traffic_view_GBs": round(enc_gbs, 1),
                    "of_spec_8000": round(enc_gbs / HBM_PEAK_GBS, 4),
                    "of_read4_write1_ceiling": round(enc_gbs / c["read4_write1_encode_shape"]["GBs"], 4),
                    "of_copy_1to1_ceiling": round(enc_gbs / c["copy_1to1"]["GBs"], 4),
                    "of_read_only_ceiling": round(enc_gbs / c["read_only"]["GBs"], 4),
                    "read_only_view_GBs": round(gbs(n_len, enc_ms), 1),
                    "read_only_view_of_spec_8000": round(gbs(n_len, enc_ms) / HBM_PEAK_GBS, 4),
                    "read_only_view_of_read_only_ceiling": round(gbs(n_len, enc_ms) / c["read_only"]["GBs"], 4),
                }
                line["ceilings"]["decode_vs"] = {
                    "total_traffic_view_GBs": round(dec_gbs, 1),
                    "of_spec_8000": round(dec_gbs / HBM_PEAK_GBS, 4),
                    "of_read1_write4_ceiling": round(dec_gbs / c["read1_write4_decode_shape"]["GBs"], 4),
                    "of_copy_1to1_ceiling": round(dec_gbs / c["copy_1to1"]["GBs"], 4),
                    "of_write_only_ceiling": round(dec_gbs / c["write_only"]["GBs"], 4),
                }
                if fused is not None:  # 1 B read : 1.25 B written per nt -- the nearest arithmetic-free stream is the 1:1 copy
                    line["ceilings"]["fused_vs"] = {
                        "total_traffic_view_GBs": round(fgbs, 1),
                        "of_spec_8000": round(fgbs / HBM_PEAK_GBS, 4),
                        "of_copy_1to1_ceiling": round(fgbs / c["copy_1to1"]["GBs"], 4),
                    }
        if configs:
            line["configs"] = configs
        if codec5 is not None:
            line["codec5"] = codec5
            line["packed_ops"] = packed_ops
        if any(r.get("configs4_shard") and "encode_ms" in r["configs4_shard"] for r in rows):
            sh = [r["configs4_shard"] for r in rows if r.get("configs4_shard") and "encode_ms" in r["configs4_shard"]]
            slow = max(s["encode_ms"]["median"] for s in sh)
            tot = sum(s["nt"] for s in sh)
            line["configs4_sharded_encode"] = {
                "what": "BASELINE.json configs[4]: n_to_bits encode of this run's share of '256 GiB over 8 GPUs' -- a 2^%d-nt (%.0f GiB) "
                        "contiguous chunk per GPU (cnt_shard_range), every rank at once, no collective; at N = 8 the whole 256 GiB"
                        % (args.shard_log2_nt, 2**args.shard_log2_nt / 2**30),
                "nt_per_gpu": sh[0]["nt"], "total_GiB": round(tot / 2**30, 1), "ranks_measured": len(sh),
                "per_gpu_gnts": {"min": round(min(s["nt"] / (s["encode_ms"]["median"] * 1e-3) / 1e9 for s in sh), 1),
                                 "max": round(max(s["nt"] / (s["encode_ms"]["median"] * 1e-3) / 1e9 for s in sh), 1)},
                "aggregate_gnts": round(tot / (sh[0]["wall_ms_per_encode_all_ranks"] * 1e-3) / 1e9, 1),
                "aggregate_definition": "all ranks' nucleotides / wall time per encode between two barriers (max over ranks), 8 "
                                        "back-to-back encodes per rank; per-GPU figures from each rank's own HIP events",
                "aggregate_gnts_from_slowest_rank_events": round(tot / (slow * 1e-3) / 1e9, 1),
                "per_gpu_frac": {"min": round(min(gbs(BYTES_PER_NT * s["nt"], s["encode_ms"]["median"]) for s in sh) / HBM_PEAK_GBS, 4),
                                 "max": round(max(gbs(BYTES_PER_NT * s["nt"], s["encode_ms"]["median"]) for s in sh) / HBM_PEAK_GBS, 4)},
                "per_gpu_read_view_frac": {"min": round(min(gbs(s["nt"], s["encode_ms"]["median"]) for s in sh) / HBM_PEAK_GBS, 4),
                                           "max": round(max(gbs(s["nt"], s["encode_ms"]["median"]) for s in sh) / HBM_PEAK_GBS, 4)},
            }
        if args.cpu_seconds > 0:  # every N (north_star: the CPU path timed in the same run at 1, 2, 4 and 8 GPUs); all GPU work is done
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, faithful_tables=world == 1)
            if world > 1:
                line["cpu_baseline"]["when"] = "after the timed region and every extra of all %d ranks (they wait at the final barrier)" % world
            if extras and world == 1:
                cpu_rows = line["cpu_baseline"].get("reference_faithful_GiBs_1thread_alloc_inclusive") or {}
                line["host_tier"] = host_tier_block(torch, args.seed, cpu_rows, ident.get("pci_bus_id"))  # bench/bench_measure.py
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
