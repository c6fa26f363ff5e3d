#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement on MI355X.

Metric (BASELINE.json): Gnt/s encode+decode on 16 GiB random ACGT; fraction of the HBM
roofline.  One "step" = one pass of the hot path over one batch: n_to_bits encode of the
device-resident 16 GiB ASCII buffer (-> 4 GiB packed) followed by bits_to_n decode of the
packed words (-> 16 GiB ASCII).  Inputs are generated on the device (counter-based ACGT
generator, seed in the JSON) and are resident in HBM before the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]                     (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (N>1, one rank per GPU)

N>1: the global buffer (N x per-GPU size) is cut into contiguous chunks on word boundaries,
rank r owns chunk r (cute_nucleotides_amd/sharding.py); there is no data-path collective --
the process group (gloo over loopback by default, RCCL with CNT_BENCH_BACKEND=nccl) is only used
for the barrier and the max-over-ranks of the timing.  Weak scaling: per-GPU work is fixed.

Rank 0 prints ONE JSON line.  `value` counts every nucleotide converted per second over all
ranks (N encoded + N decoded per step and rank), inputs already in HBM.  `roofline` is for
the n_to_bits encode kernel (the north-star target), `roofline_decode` for bits_to_n; both
use ALGORITHMIC bytes (1.25 B/nt: encode 1 read + 0.25 written, decode 0.25 read + 1
written) over the kernel's average duration measured live with HIP events on the
launch stream.  `cpu_baseline` times the oracle's ports of the reference's fastest AVX2
paths (n_to_bits_movemask / bits_to_n_shuffle) on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_NT = 1.25    # algorithmic bytes per nucleotide, each direction


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--log2-nt", type=int, default=34, help="per-GPU nucleotides = 2^k (default 34 = 16 GiB, the metric size)")
    p.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    p.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU-baseline time budget (0 = skip)")
    p.add_argument("--no-verify", action="store_true")
    return p.parse_args()


def cpu_baseline(seconds):
    """Time the oracle's x86 ports of the reference's best encoder/decoder on the host cores.
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
    cores = len(cpus)
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
    encN, decN = parallel(enc, cores, q), parallel(dec, cores, q)
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
    both = lambda e, d: 2.0 / (1.0 / e + 1.0 / d)  # nt converted per second over an encode pass + a decode pass
    return {
        "reference_faithful_40k_GiBs": faithful,
        "value": round(both(encN, decN), 3), "unit": "Gnt/s", "cores": cores, "kind": "port",
        "sample": "%d Mi random ACGT nt (%d pinned threads x >=16 Mi contiguous nt each, re-run until ~%.0f s total); "
                  "n_to_bits_movemask + bits_to_n_shuffle ports (oracle/cnt_simd_port.c), output preallocated"
                  % (n_len >> 20, cores, seconds),
        "encode_gnts": round(encN, 3), "decode_gnts": round(decN, 3),
        "one_thread": {"encode_gnts": round(enc1, 3), "decode_gnts": round(dec1, 3), "value": round(both(enc1, dec1), 3),
                       "note": "one thread over the whole sample"},
        "scalar_lut_encode_gnts_1thread": round(lut_enc, 3), "cpu_model": model,
    }


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Test hooks (never set by the driver): CNT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # CNT_BENCH_BACKEND=gloo swaps the process group's backend, so the N>1 code path can be
    # exercised on a 1-GPU box (RCCL refuses two ranks on one device).
    if os.environ.get("CNT_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    # The process group carries no data: it is the barrier + a MAX/MIN of two scalars.  gloo over
    # loopback is the default because it needs no GPU IPC and has been exercised with 2 ranks on the
    # GPU box (tests/test_gpu_bench.py); CNT_BENCH_BACKEND=nccl runs the same control plane over RCCL.
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
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
            else:
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                dist.init_process_group(backend=backend)
            dist.barrier()  # first collective: lazy connection chatter happens here, not later
        finally:
            sys.stdout.flush()
            if rank == 0:
                os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil, sharding

    n_per = 1 << args.log2_nt
    n_global = n_per * world
    lo, hi = sharding.partition(n_global, world)[rank]  # contiguous chunk on a word boundary
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

    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    dec_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    # slowest rank's per-kernel averages (for the aggregate encode / decode rates); rank 0's own
    # times stay in `roofline*`, which describe one GPU
    km = torch.tensor([enc_ms, dec_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
    enc_ms_max, dec_ms_max = float(km[0].item()), float(km[1].item())

    # BASELINE.json configs[3] in passing: the FUSED round trip (cnt_round_trip_dev) over the same
    # buffers, outside the timed region -- reported beside the headline, never part of `value`
    fused_ms = None
    if world == 1:
        fe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        cn.round_trip_dev(d_in, out_bits=d_packed, out_n=d_out)
        fe[0].record()
        for _ in range(3):
            cn.round_trip_dev(d_in, out_bits=d_packed, out_n=d_out)
        fe[1].record()
        fe[1].synchronize()
        fused_ms = fe[0].elapsed_time(fe[1]) / 3

    verified = None
    if not args.no_verify:
        # Outside the timed region, and WITHOUT the oracle (bench.py may touch oracle/ only in its
        # cpu_baseline leg; bit-exact parity against the oracle is what tests/ -m gpu establish):
        #  (1) the timed buffers round-trip: decode(encode(x)) == x, compared on the device;
        #  (2) the reference's own bench input (benches/bench_n_to_bits.rs:68-78, "ATCG" x 10000)
        #      encodes to 1250 x 0xD8D8D8D8D8D8D8D8 -- the reference's unit-test vector
        #      (n_to_bits.rs:414-415) -- through the same kernels, and decodes back.
        import numpy as np

        ok = devutil.count_mismatch(d_in, d_out) == 0
        kat = torch.from_numpy(np.frombuffer(b"ATCG" * 10000, dtype=np.uint8).copy()).to(dev)
        kat_bits = cn.n_to_bits_dev(kat)
        ok = ok and bool((kat_bits.cpu().numpy().view(np.uint64) == np.uint64(0xD8D8D8D8D8D8D8D8)).all()) and kat_bits.numel() == 1250
        ok = ok and bool(torch.equal(cn.bits_to_n_dev(kat_bits, 40000), kat))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())

    if rank == 0:
        nt_per_step = 2 * n_global  # N encoded + N decoded, all ranks
        value = nt_per_step * args.steps / elapsed / 1e9
        enc_gbs = BYTES_PER_NT * n_len / (enc_ms * 1e-3) / 1e9
        dec_gbs = BYTES_PER_NT * n_len / (dec_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and args.log2_nt == 34:
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = None
        line = {
            "metric": "Gnt/s encode+decode on 16 GiB random ACGT; % HBM read roofline at 1/8 GPU",
            "value": round(value, 3), "unit": "Gnt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": "n_to_bits encode + bits_to_n decode of a device-resident uniform random ACGT buffer, "
                            "%.3g GiB (2^%d nt) per GPU" % (n_per / 2**30, args.log2_nt),
                "nt_per_gpu": n_per, "nt_per_step": nt_per_step, "seed": hex(args.seed),
                "sharding": ("contiguous chunks on word boundaries, no data-path collective; control plane: " + backend)
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
                "traffic": (traffic or {}).get("encode_bytes_per_launch"),
                "avg_kernel_ms": round(enc_ms, 4), "algorithmic_bytes_per_launch": int(BYTES_PER_NT * n_len),
                "read_only_view": {"achieved": round(n_len / (enc_ms * 1e-3) / 1e9, 1),
                                   "frac": round(n_len / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            },
            "roofline_decode": {
                "kernel": "bits_to_n (decode)", "bound": "hbm", "achieved": round(dec_gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(dec_gbs / HBM_PEAK_GBS, 4),
                "traffic": (traffic or {}).get("decode_bytes_per_launch"),
                "avg_kernel_ms": round(dec_ms, 4), "algorithmic_bytes_per_launch": int(BYTES_PER_NT * n_len),
            },
            "verified": verified,
            "value_definition": "nucleotides converted per second over all ranks: each step encodes nt_per_gpu and decodes "
                                "nt_per_gpu on every rank (nt_per_step = 2 x n_gpus x nt_per_gpu); equals the harmonic mean "
                                "of the encode and decode rates",
        }
        if fused_ms is not None:
            fgbs = 2.25 * n_len / (fused_ms * 1e-3) / 1e9
            line["fused_round_trip"] = {
                "what": "cnt_round_trip_dev: one pass reads the ASCII and writes packed words + decoded ASCII "
                        "(BASELINE.json configs[3]); measured after the timed region, not part of value",
                "ms": round(fused_ms, 4), "nt_converted_gnts": round(2 * n_len / (fused_ms * 1e-3) / 1e9, 3),
                "bytes_per_nt": 2.25, "achieved": round(fgbs, 1), "unit": "GB/s", "frac": round(fgbs / HBM_PEAK_GBS, 4),
            }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
