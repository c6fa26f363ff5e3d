"""Measurement helpers of bench.py (the repo-root driver contract): statistics, HIP-event timing loops, the same-run
ceilings, the 5-letter / packed-ops / host-tier / PCIe blocks of the N = 1 line and the live HBM-traffic measurement.
bench.py keeps argument parsing, the timed region, the CPU-baseline leg (the only code that may touch oracle/) and the
JSON line; `python bench.py --gpus N` without a launcher is bench/bench_single_process.py.  Imported by bench.py as a
plain module from this directory (the name `bench` belongs to bench.py itself)."""
import ctypes
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOST_TIER_LOG2 = (12, 14, 16, 18, 20, 22, 24, 26, 28, 30)  # sizes of the host-tier / one-CPU-thread crossover table
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_NT = 1.25    # algorithmic bytes per nucleotide, each direction


def physical_cores(cpus):
    """distinct (package, core) pairs among the CPUs this process may run on"""
    seen = set()
    for c in cpus:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            return None
    return len(seen)


def sockets():
    try:
        return len({open(os.path.join(d, "topology", "physical_package_id")).read().strip()
                    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*") if os.path.exists(os.path.join(d, "topology"))})
    except OSError:
        return None


def stats_ms(ms):
    """mean / median / min / max and, from 10 samples on, p10 / p90 (decode's launches spread 3-9 % on one box with the
    clock flat: profiles/r04_decode_spread.md -- a mean alone hides that)"""
    s = sorted(ms)
    out = {"mean": round(statistics.fmean(ms), 4), "median": round(statistics.median(ms), 4), "min": round(s[0], 4),
           "max": round(s[-1], 4), "n": len(ms)}
    if len(s) >= 10:
        out["p10"], out["p90"] = round(s[len(s) // 10], 4), round(s[(9 * len(s)) // 10], 4)
    return out


def gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9


def timed_calls(torch, fn, iters, warm=1):
    """[ms] of `iters` individual fn() calls, each between two HIP events on torch's current stream (the stream
    the C ABI is handed)."""
    for _ in range(warm):
        fn()
    out = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return out


def timed_queued(torch, fn, reps, queue, warm=1):
    """[ms per call] of `reps` measurements, each `queue` back-to-back fn() calls between ONE pair of HIP events: the
    average launch duration with the launches queued behind each other, as in the timed region of the headline (and
    as rocprofv3's kernel trace sees them).  A single call between two events also counts the 10-50 us the host needs
    to get from the event record to the launch -- 5 % of a 0.2 ms kernel at the 1 GiB configs."""
    for _ in range(warm):
        fn()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(queue):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / queue)
    return out


def numpy_pack(n):
    """(byte>>1)&3, 32 codes per u64, LSB first (n_to_bits.rs:38-43, :85) -- three lines of numpy for the sampled
    check of the TIMED packed buffer; bench.py touches oracle/ only in cpu_baseline()"""
    import numpy as np

    c = ((n >> 1) & 3).astype(np.uint64).reshape(-1, 32)
    return (c << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)


def load_probes():
    """bench/libcnt_probes.so (built by __graft_entry__.build(), travels with the snapshot): no-arithmetic streams
    issued like the shipped kernels, for the same-run ceilings.  None if it is not there."""
    path = os.path.join(ROOT, "bench", "libcnt_probes.so")
    if not os.path.exists(path):
        return None
    P = ctypes.CDLL(path)
    P.probe_shipped.restype = ctypes.c_int
    P.probe_shipped.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    return P


def measure_ceilings(torch, P, d_a, d_b, nbytes, iters=5):
    """TB/s (decimal) of the five no-arithmetic streams over the headline's own buffers: d_a is only read, d_b is
    overwritten (call after verification).  nbytes = the 16-B-per-lane side (the ASCII side of the codec)."""
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = {}
    for name, kind, moved in (("read_only", 0, nbytes), ("write_only", 4, nbytes), ("copy_1to1", 1, 2 * nbytes),
                              ("read4_write1_encode_shape", 2, 1.25 * nbytes), ("read1_write4_decode_shape", 3, 1.25 * nbytes)):
        def call():
            rc = P.probe_shipped(kind, d_a.data_ptr(), d_b.data_ptr(), nbytes, stream)
            if rc:
                raise RuntimeError("probe_shipped(%d) -> %d" % (kind, rc))
        ms = timed_calls(torch, call, iters)
        rows[name] = {"GBs": round(gbs(moved, statistics.median(ms)), 1), "best_GBs": round(gbs(moved, min(ms)), 1),
                      "bytes_moved": int(moved), "ms_median": round(statistics.median(ms), 4)}
    return rows


def measure_configs_1gib(torch, d_in, d_packed, d_out, verify):
    """BASELINE.json configs[1] / [2] (1 GiB encode / decode) on the first 2^30 nt of the headline's own buffers, and SURVEY
    8d's size that is not a multiple of 32 (2^30 - 19: thirteen nucleotides in the last word) in the same run.  Average
    launch duration with 10 launches queued per event pair (what the headline's timed region and rocprofv3 see); the isolated
    single-launch figure, which also counts the host's event-to-launch gap, is kept beside it."""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil

    configs = {}
    m = 1 << 30
    enc30 = lambda: cn.n_to_bits_dev(d_in[:m], out=d_packed[: m // 32])
    dec30 = lambda: cn.bits_to_n_dev(d_packed[: m // 32], m, out=d_out[:m])
    e30, d30 = timed_queued(torch, enc30, 8, 10, warm=2), timed_queued(torch, dec30, 8, 10, warm=2)
    e30s, d30s = timed_calls(torch, enc30, 10, warm=1), timed_calls(torch, dec30, 10, warm=1)
    for key, lst, single in (("configs[1] n_to_bits encode, 1 GiB (2^30 nt)", e30, e30s), ("configs[2] bits_to_n decode, 1 GiB (2^30 nt)", d30, d30s)):
        st = stats_ms(lst)
        configs[key] = {"ms": st, "timing": "10 launches queued per HIP-event pair, 8 pairs", "gnts": round(m / (st["median"] * 1e-3) / 1e9, 1),
                        "achieved_GBs": round(gbs(BYTES_PER_NT * m, st["median"]), 1),
                        "frac": round(gbs(BYTES_PER_NT * m, st["median"]) / HBM_PEAK_GBS, 4),
                        "isolated_single_launch": {"ms": stats_ms(single), "frac": round(gbs(BYTES_PER_NT * m, statistics.median(single)) / HBM_PEAK_GBS, 4)}}
    if verify:
        configs["configs[2] bits_to_n decode, 1 GiB (2^30 nt)"]["round_trip_verified"] = devutil.count_mismatch(d_in[:m], d_out[:m]) == 0
    r = m - 19  # 2^30 - 19 = 32 * (2^25 - 1) + 13
    if d_in.numel() >= r:
        encr = lambda: cn.n_to_bits_dev(d_in[:r], out=d_packed[: r // 32 + 1])
        decr = lambda: cn.bits_to_n_dev(d_packed[: r // 32 + 1], r, out=d_out[:r])
        er, dr = timed_queued(torch, encr, 8, 10, warm=2), timed_queued(torch, decr, 8, 10, warm=2)
        assert r % 32 == 13
        row = {"encode_ms": stats_ms(er), "decode_ms": stats_ms(dr), "timing": "10 launches queued per HIP-event pair, 8 pairs",
               "launches_per_call": 1,  # head words and ragged end ride in the tile kernel's grid (tests/test_gpu_codec2.py counts the graph nodes)
               "encode_frac": round(gbs(BYTES_PER_NT * r, statistics.median(er)) / HBM_PEAK_GBS, 4),
               "decode_frac": round(gbs(BYTES_PER_NT * r, statistics.median(dr)) / HBM_PEAK_GBS, 4),
               "encode_vs_aligned_2p30": round(statistics.median(er) / statistics.median(e30), 4),
               "decode_vs_aligned_2p30": round(statistics.median(dr) / statistics.median(d30), 4)}
        if verify:
            last = int(d_packed[r // 32].item()) & 0xFFFFFFFFFFFFFFFF
            row["round_trip_verified"] = devutil.count_mismatch(d_in[:r], d_out[:r]) == 0 and (last >> 26) == 0  # 13 nt used, 38 high bits zero
        configs["ragged: 2^30 - 19 nt (13 nt in the last word, zero-padded)"] = row
    return configs


def measure_config3_64gib(torch, dev, seed, verify):
    """BASELINE.json configs[3] at its own size: 2^36 nt (64 GiB of ASCII, 144 GiB resident) as two passes and as the fused
    round trip; ({rows}, all verified | None).  Skips VISIBLY when less than 150 GiB of HBM are free."""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil

    b_len = 1 << 36
    free, _ = torch.cuda.mem_get_info()
    key2, keyf = "configs[3] encode + decode as two passes, 64 GiB (2^36 nt)", "configs[3] fused round trip, 64 GiB (2^36 nt)"
    if free <= 150 * 2**30:
        skipped = {"skipped": "needs 150 GiB of free HBM, %.0f GiB free" % (free / 2**30)}
        return {key2: skipped, keyf: skipped}, None
    b_in = torch.empty(b_len, dtype=torch.uint8, device=dev)
    b_packed = torch.empty(b_len // 32, dtype=torch.int64, device=dev)
    b_out = torch.empty(b_len, dtype=torch.uint8, device=dev)
    devutil.fill_random_acgt(b_in, seed)
    be = timed_calls(torch, lambda: cn.n_to_bits_dev(b_in, out=b_packed), 3)
    bd = timed_calls(torch, lambda: cn.bits_to_n_dev(b_packed, b_len, out=b_out), 3)
    ok2 = devutil.count_mismatch(b_in, b_out) == 0 if verify else None
    two_sum = devutil.checksum_words(b_packed) if verify else None
    b_out.zero_()
    bf = timed_calls(torch, lambda: cn.round_trip_dev(b_in, out_bits=b_packed, out_n=b_out), 3)
    okf = (devutil.count_mismatch(b_in, b_out) == 0 and devutil.checksum_words(b_packed) == two_sum) if verify else None
    t2 = statistics.median(be) + statistics.median(bd)
    tf = statistics.median(bf)
    rows = {key2: {"encode_ms": stats_ms(be), "decode_ms": stats_ms(bd), "bytes_per_nt": 2.5, "resident_GiB": 144,
                   "nt_converted_gnts": round(2 * b_len / (t2 * 1e-3) / 1e9, 1), "achieved_GBs": round(gbs(2.5 * b_len, t2), 1),
                   "frac": round(gbs(2.5 * b_len, t2) / HBM_PEAK_GBS, 4), "round_trip_verified": ok2},
            keyf: {"ms": stats_ms(bf), "bytes_per_nt": 2.25, "resident_GiB": 144,
                   "nt_converted_gnts": round(2 * b_len / (tf * 1e-3) / 1e9, 1), "achieved_GBs": round(gbs(2.25 * b_len, tf), 1),
                   "frac": round(gbs(2.25 * b_len, tf) / HBM_PEAK_GBS, 4), "verified_against_two_passes": okf}}
    del b_in, b_packed, b_out
    torch.cuda.empty_cache()
    return rows, (bool(ok2 and okf) if verify else None)


def measure_codec5(torch, seed, log2_nt, reps=4, queue=4):
    """SURVEY 8 f-1 on the driver line: the 5-letter codec (n_to_bits2 / bits_to_n2, n_to_bits2.rs:37-107) at the metric's
    size on device-resident random ACGTN.  Algorithmic bytes per nucleotide 1 + 8/27 in each direction (27 nt per u64).
    Verified in-run: the reference's own vector ("ATCGN" x 7, n_to_bits2.rs:276-277) through the same device entry
    points, and decode(encode(x)) == x over the whole timed buffer."""
    import numpy as np

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil

    n = 1 << log2_nt
    words = (n + 26) // 27
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    packed = torch.empty(words, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    devutil.fill_random_acgtn(d, seed + 5)
    enc = lambda: cn.n_to_bits2_dev(d, out=packed)
    dec = lambda: cn.bits_to_n2_dev(packed, n, out=back)
    e, dd = timed_queued(torch, enc, reps, queue, warm=1), timed_queued(torch, dec, reps, queue, warm=1)
    ok = devutil.count_mismatch(d, back) == 0
    kat = torch.from_numpy(np.frombuffer(b"ATCGN" * 7, dtype=np.uint8).copy()).to(dev)
    kb = cn.n_to_bits2_dev(kat)
    ok = ok and [int(x) & 0xFFFFFFFFFFFFFFFF for x in kb.cpu().tolist()] == [0x36A45D1F46D48BA3, 0x5D1F4]
    ok = ok and bool(torch.equal(cn.bits_to_n2_dev(kb, 35), kat))
    bpn = 1.0 + 8.0 / 27.0
    row = lambda ms: {"ms": stats_ms(ms), "gnts": round(n / (statistics.median(ms) * 1e-3) / 1e9, 1),
                      "achieved_GBs": round(gbs(bpn * n, statistics.median(ms)), 1), "frac": round(gbs(bpn * n, statistics.median(ms)) / HBM_PEAK_GBS, 4)}
    out = {"what": "5-letter codec {A,C,G,T/U,N} (n_to_bits2.rs): 3 nt -> a + 5b + 25c, 27 nt per u64; device tier, 2^%d nt of random ACGTN "
                   "(P(N) = 1/16), %d launches queued per HIP-event pair" % (log2_nt, queue),
           "nt": n, "algorithmic_bytes_per_nt": round(bpn, 4), "encode": row(e), "decode": row(dd), "verified": bool(ok),
           "encode_kernel": dict(devutil.variants("encode2"))[devutil.get_tuning("encode2")],
           "decode_kernel": dict(devutil.variants("decode2"))[devutil.get_tuning("decode2")]}
    del d, packed, back
    torch.cuda.empty_cache()
    return out


def measure_packed_ops(torch, seed, log2_nt, reps=4, queue=5):
    """SURVEY 8 f-4 on the driver line: the packed-domain operations at the metric's size (README.md:20-25,45 names them,
    the reference does not implement them: parity unpinned by construction).  Algorithmic bytes per nucleotide: hamming
    2 x 0.25 read, complement / reverse complement 0.25 read + 0.25 written, validate 1 read.  Verified in-run by
    algebraic properties over the whole buffers: two independent uniform sequences differ in 3/4 of the positions,
    hamming(x, complement(x)) == len, both complements are involutions, a generated ACGT buffer validates clean."""
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil, packed_ops as po

    n = 1 << log2_nt
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    devutil.fill_random_acgt(d, seed + 6)
    x = cn.n_to_bits_dev(d)
    devutil.fill_random_acgt(d, seed + 7)
    y = cn.n_to_bits_dev(d)
    out = torch.empty_like(x)
    acc = torch.zeros(1, dtype=torch.int64, device=dev)  # the reductions ADD to a caller-owned counter: nothing but their kernel is timed
    ops = (("hamming", 0.5, lambda: po.hamming_dev(x, y, n, acc=acc)),
           ("complement", 0.5, lambda: po.complement_dev(x, n, out=out)),
           ("reverse_complement", 0.5, lambda: po.reverse_complement_dev(x, n, out=out)),
           ("validate", 1.0, lambda: po.validate_dev(d, acc=acc)))
    rows = {}
    for name, bpn, fn in ops:
        ms = timed_queued(torch, fn, reps, queue, warm=1)
        med = statistics.median(ms)
        rows[name] = {"ms": stats_ms(ms), "bytes_per_nt": bpn, "gnts": round(n / (med * 1e-3) / 1e9, 1),
                      "achieved_GBs": round(gbs(bpn * n, med), 1), "frac": round(gbs(bpn * n, med) / HBM_PEAK_GBS, 4)}
    # the VALIDATED encode (round 6): cnt_n_to_bits_checked_dev packs and counts the bytes outside the alphabet in ONE pass --
    # 1.25 B/nt where cnt_validate_dev + cnt_n_to_bits_dev move 2.25.  Timed interleaved with the plain encoder on the same
    # buffers (A, B, A, B ...: one drifting box cannot favour either), plus the two-pass form it replaces.
    plain_ms, checked_ms, two_ms = [], [], []
    for _ in range(3):
        plain_ms += timed_queued(torch, lambda: cn.n_to_bits_dev(d, out=out), 2, queue, warm=1)
        checked_ms += timed_queued(torch, lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc), 2, queue, warm=1)
    two_ms = timed_queued(torch, lambda: (po.validate_dev(d, acc=acc), cn.n_to_bits_dev(d, out=out)), 2, queue, warm=1)
    med_p, med_c, med_2 = statistics.median(plain_ms), statistics.median(checked_ms), statistics.median(two_ms)
    rows["checked_encode"] = {"ms": stats_ms(checked_ms), "bytes_per_nt": 1.25, "gnts": round(n / (med_c * 1e-3) / 1e9, 1),
                              "achieved_GBs": round(gbs(1.25 * n, med_c), 1), "frac": round(gbs(1.25 * n, med_c) / HBM_PEAK_GBS, 4),
                              "plain_encode_ms": stats_ms(plain_ms), "checked_over_plain": round(med_c / med_p, 4),
                              "validate_then_encode_ms": stats_ms(two_ms), "checked_over_two_pass": round(med_c / med_2, 4),
                              "what": "cnt_n_to_bits_checked_dev: encode + count of bytes outside ACGTUacgtu in one launch; interleaved A/B with cnt_n_to_bits_dev on the same buffers"}
    acc.zero_()
    cn.n_to_bits_checked_dev(d, out=out, acc=acc)
    ok_checked = int(acc.item()) == 0 and devutil.checksum_words(out) == devutil.checksum_words(cn.n_to_bits_dev(d))
    spots = torch.tensor(sorted({0, 2047, 2048, n // 2 - 1, n // 2, n - 1} | {(k * 0x9E3779B1) % n for k in range(1, 100)}), device=dev)
    keep = d[spots].clone()
    d[spots] = 0x0A  # line feeds
    acc.zero_()
    cn.n_to_bits_checked_dev(d, out=out, acc=acc)
    ok_checked = ok_checked and int(acc.item()) == spots.numel()
    d[spots] = keep
    # the other end of the scale: EVERY 2-KiB tile holds a stray (a line feed after every 60 letters, a FASTA file fed in whole)
    # -- every wave recounts its registers exactly and issues one atomic on the single counter -- and a buffer of nothing but strays
    d[60::61] = 0x0A
    acc.zero_()
    fasta_ms = timed_queued(torch, lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc), 2, queue, warm=1)
    fasta_ok = int(acc.item()) == (2 * queue + 1) * len(range(60, n, 61))
    acc256 = torch.zeros(2048, dtype=torch.int64, device=dev)  # CNT_SPREAD_COUNT: workgroup b adds to slot b % 2048, the count is the sum
    spread_ms = timed_queued(torch, lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc256, spread=True), 2, queue, warm=1)
    spread_ok = int(acc256.sum().item()) == (2 * queue + 1) * len(range(60, n, 61))
    d.fill_(0x4E)
    all_ms = timed_queued(torch, lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc), 2, queue, warm=1)
    all_spread_ms = timed_queued(torch, lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc256, spread=True), 2, queue, warm=1)
    devutil.fill_random_acgt(d, seed + 7)
    clean_spread_ms = timed_queued(torch, lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc256, spread=True), 2, queue, warm=1)
    over = lambda ms: round(statistics.median(ms) / med_c, 3)
    rows["checked_encode"]["dirty_input"] = {
        "a_line_feed_every_61st_byte": {"one_counter": {"ms": stats_ms(fasta_ms), "over_clean": over(fasta_ms)},
                                        "CNT_SPREAD_COUNT": {"ms": stats_ms(spread_ms), "over_clean": over(spread_ms)}, "count_exact": bool(fasta_ok and spread_ok)},
        "every_byte_a_stray": {"one_counter": {"ms": stats_ms(all_ms), "over_clean": over(all_ms)},
                               "CNT_SPREAD_COUNT": {"ms": stats_ms(all_spread_ms), "over_clean": over(all_spread_ms)}},
        "clean_input_CNT_SPREAD_COUNT": {"ms": stats_ms(clean_spread_ms), "over_clean": over(clean_spread_ms)},
        "what": "a wave that saw a stray issues one no-return atomic: on ONE device counter (the default) 8.4 M of them serialise at 2^34 nt when every 2-KiB tile is dirty; "
                "with CNT_SPREAD_COUNT they go to 2048 counters (16 KiB, 128 cache lines) whose sum is the count"}
    ok_checked = ok_checked and fasta_ok and spread_ok
    rows["checked_encode"]["verified"] = bool(ok_checked)
    dist = int(po.hamming_dev(x, y, n).item())
    ok = ok_checked and abs(dist / n - 0.75) < 1e-3
    comp = po.complement_dev(x, n)
    ok = ok and int(po.hamming_dev(x, comp, n).item()) == n
    ok = ok and bool(torch.equal(po.complement_dev(comp, n), x))
    del comp
    rc = po.reverse_complement_dev(x, n)
    ok = ok and bool(torch.equal(po.reverse_complement_dev(rc, n), x))
    del rc
    ok = ok and int(po.validate_dev(d).item()) == 0
    rows.update({"what": "packed-domain operations on 2-bit words without decoding; device tier, 2^%d nt, %d calls queued per HIP-event pair"
                         % (log2_nt, queue), "nt": n, "verified": bool(ok), "parity": "unpinned (no reference vectors exist); property checks in-run"})
    del d, x, y, out
    torch.cuda.empty_cache()
    return rows


def measure_pcie(torch, nbytes=1 << 30, reps=5):
    """Same-run PCIe ceilings for the host tier: hipMemcpy between PINNED host memory and the device, both directions,
    GiB/s -- what a transfer-bound host-pointer call could reach with free staging."""
    dev = torch.device("cuda", torch.cuda.current_device())
    h = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    h.zero_()
    out = {}
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        ms = timed_calls(torch, fn, reps, warm=1)
        out[name + "_GiBs"] = round(nbytes / (statistics.median(ms) * 1e-3) / 2**30, 2)
    del h, d
    return out


def measure_host_tier(seed):
    """The drop-in host-pointer tier (cnt_n_to_bits / cnt_bits_to_n: H2D + kernel + D2H inside, PCIe-bound) at the
    sizes of the crossover table, timed like the reference's harness times its functions: one calling thread, the
    output allocated inside every timed call (np.empty -> fresh pages, like a fresh Vec), and again into a reused
    output.  GiB/s of nucleotides.  Never part of `value`."""
    import numpy as np

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(seed)
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 1 << HOST_TIER_LOG2[-1], dtype=np.uint8)]
    rows = {}
    for log2 in HOST_TIER_LOG2:
        m = 1 << log2
        n = big[:m]
        words = m // 32
        bits = np.empty(words, dtype=np.uint64)
        back = np.empty(m, dtype=np.uint8)
        p = lambda a: ctypes.c_void_p(a.ctypes.data)

        def enc_fresh():
            out = np.empty(words, dtype=np.uint64)
            return L.cnt_n_to_bits(p(n), m, p(out), words)

        def dec_fresh():
            out = np.empty(m, dtype=np.uint8)
            return L.cnt_bits_to_n(p(bits), words, m, p(out))

        kept = []  # "dropped later": the result outlives the call and is freed outside the timed loop (criterion's iter_with_large_drop)

        def enc_fresh_kept():
            out = np.empty(words, dtype=np.uint64)
            kept.append(out)
            return L.cnt_n_to_bits(p(n), m, p(out), words)

        def dec_fresh_kept():
            out = np.empty(m, dtype=np.uint8)
            kept.append(out)
            return L.cnt_bits_to_n(p(bits), words, m, p(out))

        def enc_reuse():
            return L.cnt_n_to_bits(p(n), m, p(bits), words)

        def dec_reuse():
            return L.cnt_bits_to_n(p(bits), words, m, p(back))

        into_w, into_n = np.empty(words, dtype=np.uint64), np.empty(m, dtype=np.uint8)  # the caller's own vectors, kept across calls

        def enc_into():  # the `_into` form of the mirrors (n_to_bits_hip_into): nothing allocated or dropped in the call
            cn.n_to_bits_hip_into(n, into_w)
            return 0

        def dec_into():
            cn.bits_to_n_hip_into(bits, m, into_n)
            return 0

        # both sides in PINNED memory (cnt_host_alloc): no staging copy either way, the copy engines use the arrays in place
        cases = [("n_to_bits_hip reused out", enc_reuse), ("bits_to_n_hip reused out", dec_reuse),
                 ("n_to_bits_hip_into", enc_into), ("bits_to_n_hip_into", dec_into),
                 ("n_to_bits_hip fresh out", enc_fresh), ("bits_to_n_hip fresh out", dec_fresh)]
        # (calls of <= 2^20 nt take the zero-copy small path either way)
        pin_n, pin_w, pin_back = cn.pinned_empty(m, np.uint8), cn.pinned_empty(words, np.uint64), cn.pinned_empty(m, np.uint8)
        pin_n[:] = n
        cases[2:2] = [("n_to_bits_hip pinned in + out", lambda: L.cnt_n_to_bits(p(pin_n), m, p(pin_w), words)),
                      ("bits_to_n_hip pinned in + out", lambda: L.cnt_bits_to_n(p(pin_w), words, m, p(pin_back)))]
        row = {}
        for name, fn in cases:
            assert fn() == 0
            t0, k = time.perf_counter(), 0
            while True:
                fn()
                k += 1
                dt = time.perf_counter() - t0
                if dt > 0.12 or k >= 20000:
                    break
            row[name] = round(m / (dt / k) / 2**30, 3)
            row[name + " us"] = round(dt / k * 1e6, 2)
        if log2 >= 26:  # large outputs: the same fresh-output calls with the DROP of the result outside the timed loop
            for name, fn in (("n_to_bits_hip fresh out, dropped later", enc_fresh_kept), ("bits_to_n_hip fresh out, dropped later", dec_fresh_kept)):
                reps = 3 if log2 >= 30 else 6
                fn()
                kept.clear()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                dt = time.perf_counter() - t0
                kept.clear()
                row[name + " us"] = round(dt / reps * 1e6, 2)
        assert np.array_equal(back, n) and np.array_equal(into_n, n) and np.array_equal(into_w, bits)
        assert np.array_equal(pin_w, bits) and np.array_equal(pin_back, n)
        del pin_n, pin_w, pin_back
        rows["2^%d" % log2] = row
    rows["placement"] = {"input": host_placement(big), "reused_output": host_placement(back)}
    return rows


def host_tier_block(torch, seed, cpu_rows, pci_bus_id):
    """The `host_tier` block of the N = 1 line: the drop-in host-slice calls at the crossover table's sizes (reused / `_into` /
    fresh outputs), the same-run PCIe ceilings and the fraction of them reached at 1 GiB, the three timing conventions side by
    side, where the buffers lie, and the crossover against one CPU thread of the reference's fastest AVX2 path."""
    rows = measure_host_tier(seed)
    pcie = measure_pcie(torch)
    names = ("n_to_bits_hip reused out", "bits_to_n_hip reused out", "n_to_bits_hip pinned in + out", "bits_to_n_hip pinned in + out",
             "n_to_bits_hip fresh out", "bits_to_n_hip fresh out")
    big = rows["2^%d" % HOST_TIER_LOG2[-1]]
    # encode moves N bytes up and N/4 down (full duplex: the H2D leg bounds it), decode N/4 up and N down (the D2H leg)
    frac = {nm: round(big[nm] / pcie["h2d_GiBs" if nm.startswith("n_to_bits") else "d2h_GiBs"], 4) for nm in names}
    return {
        "pcie_ceiling": dict(pcie, what="pinned hipMemcpy of 1 GiB, median of 5, same run"),
        "frac_of_pcie_ceiling_at_2^%d" % HOST_TIER_LOG2[-1]: frac,
        "timing_conventions": {
            "what": "the same host-slice call under the three ways a caller can time it, microseconds per call: `drop_inside` = result "
                    "allocated AND the previous one freed inside the timed loop (the reference's harness, benches/bench_n_to_bits.rs:6-7; on "
                    "this host munmap alone is ~47 ms per GiB of huge pages, 120 ms per GiB of 4-KiB pages, with or without HIP in the "
                    "process: profiles/r03_munmap_lab.log), `drop_outside` = allocated inside, freed later, `into` = the `_into` form of the "
                    "mirrors (n_to_bits_hip_into / bits_to_n_hip_into: the caller keeps the vector; rust/src/hip.rs, cute_nucleotides.hpp, "
                    "n_to_bits.py)",
            **{"2^%d" % k: {fn: {"drop_inside_us": rows["2^%d" % k][fn + " fresh out us"],
                                 "drop_outside_us": rows["2^%d" % k].get(fn + " fresh out, dropped later us"),
                                 "into_us": rows["2^%d" % k][fn + "_into us"]}
                            for fn in ("n_to_bits_hip", "bits_to_n_hip")} for k in HOST_TIER_LOG2 if k >= 26}},
        "fresh_over_reused_at_2^%d" % HOST_TIER_LOG2[-1]: {
            "what": "time of a call whose output is allocated inside it over the time into a reused output (see timing_conventions)",
            "n_to_bits_hip": {"drop_inside": round(big["n_to_bits_hip fresh out us"] / big["n_to_bits_hip reused out us"], 3),
                              "drop_outside": round(big["n_to_bits_hip fresh out, dropped later us"] / big["n_to_bits_hip reused out us"], 3),
                              "into": round(big["n_to_bits_hip_into us"] / big["n_to_bits_hip reused out us"], 3)},
            "bits_to_n_hip": {"drop_inside": round(big["bits_to_n_hip fresh out us"] / big["bits_to_n_hip reused out us"], 3),
                              "drop_outside": round(big["bits_to_n_hip fresh out, dropped later us"] / big["bits_to_n_hip reused out us"], 3),
                              "into": round(big["bits_to_n_hip_into us"] / big["bits_to_n_hip reused out us"], 3)}},
        "what": "the drop-in host-slice calls (H2D + kernel + D2H inside; PCIe-bound, never `value`), one calling thread; "
                "`fresh out` allocates the output inside the timed call like the reference's functions do; microseconds per call "
                "at 2^k nt (GiB/s of the fresh-out calls are in the crossover table)",
        "log2_nt": list(HOST_TIER_LOG2),
        "placement": dict(rows["placement"], gpu_numa_node=gpu_numa_node(pci_bus_id)),
        "us_per_call": {nm: [rows["2^%d" % k][nm + " us"] for k in HOST_TIER_LOG2] for nm in names},
        "crossover_vs_one_cpu_thread": crossover(rows, cpu_rows)}


def host_placement(arr):
    """Where a host array's pages lie (NUMA nodes, share in transparent huge pages) and which CPU the caller is on: the host
    tier's 1-GiB rows move by 15 % with these (profiles/r03_host_numa_placement.jsonl), so the line says what it ran on."""
    out = {}
    try:
        addr = arr.ctypes.data + arr.nbytes // 2  # the middle: an madvise'd array is several mappings (numpy advises its interior)
        start = None
        for line in open("/proc/self/maps"):
            lo, hi = (int(x, 16) for x in line.split()[0].split("-"))
            if lo <= addr < hi:
                start = lo
                break
        if start is not None:
            for line in open("/proc/self/numa_maps"):
                f = line.split()
                if int(f[0], 16) == start:
                    out["pages_per_node"] = {x.split("=")[0]: int(x.split("=")[1]) for x in f[2:] if x[0] == "N" and x[1:2].isdigit()}
                    break
            want, size, huge = "%x-" % start, None, None
            hit = False
            for line in open("/proc/self/smaps"):
                if line.startswith(want):
                    hit = True
                elif hit and line.startswith("Size:"):
                    size = int(line.split()[1])
                elif hit and line.startswith("AnonHugePages:"):
                    huge = int(line.split()[1])
                    break
            if size:
                out["huge_page_share"] = round(huge / size, 3)
        cpu = ctypes.CDLL(None).sched_getcpu()
        node = [d for d in os.listdir("/sys/devices/system/cpu/cpu%d" % cpu) if d.startswith("node")]
        out["caller_cpu"] = cpu
        out["caller_node"] = node[0] if node else None
    except (OSError, ValueError, AttributeError, IndexError):
        pass
    return out


def gpu_numa_node(bdf):
    try:
        return int(open("/sys/bus/pci/devices/%s/numa_node" % str(bdf).lower()).read())
    except (OSError, ValueError):
        return None


def crossover(host_rows, cpu_rows):
    """smallest table size from which the host tier stays ahead of ONE CPU thread running the reference's fastest
    AVX2 path, both allocating their output inside the call (the reference's bench rule)"""
    out = {}
    for gpu_key, cpu_key in (("n_to_bits_hip fresh out", "n_to_bits_movemask"), ("bits_to_n_hip fresh out", "bits_to_n_shuffle")):
        sizes = [k for k in host_rows if k in cpu_rows and cpu_key in cpu_rows[k]]
        ahead = [host_rows[k][gpu_key] > cpu_rows[k][cpu_key] for k in sizes]
        first = next((sizes[i] for i in range(len(sizes)) if all(ahead[i:])), None)
        out[gpu_key.split()[0] + " vs " + cpu_key] = {
            "host_tier_ahead_from": first,
            "table_GiBs": {k: [host_rows[k][gpu_key], cpu_rows[k][cpu_key]] for k in sizes},
            "columns": ["host tier (PCIe inside)", "one CPU thread, " + cpu_key]}
    return out


# ---- first contact with real multi-device hardware: every N > 1 line diagnoses itself -------------
EXPECTED_CHIP = {"compute_units": 256, "lds_bytes_per_cu": 163840, "xcds": 8}  # an MI355X in SPX mode (cnt_chip_info)


def chip_info(L, index):
    """what the library's launch geometry for this device is sized from (cnt_chip_info): CUs, LDS per CU, XCDs"""
    v = [ctypes.c_int(0) for _ in range(3)]
    rc = L.cnt_chip_info(index, *[ctypes.byref(x) for x in v])
    if rc != 0:
        return {"error": "cnt_chip_info -> %d" % rc}
    return dict(zip(("compute_units", "lds_bytes_per_cu", "xcds"), (x.value for x in v)))


def device_row(torch, devutil, L, index):
    """identity of one visible device for a `ranks` row: index, PCI address, NUMA node, name, UUID, HBM size, chip geometry"""
    ident = dict(devutil.device_identity(index))
    props = torch.cuda.get_device_properties(index)
    ident.update({"name": props.name, "uuid": str(getattr(props, "uuid", "")) or None, "hbm_GiB": round(props.total_memory / 2**30, 1),
                  "chip": chip_info(L, index)})
    return ident


def first_contact_devices(rows, n_gpus, shared_hook, **extra):
    """The `devices` block of a bench line, and the refusal of a run that is not what it says: N ranks must sit on N
    DISTINCT devices (UUID, else PCI address) unless the test hook that folds them onto fewer is on -- a folded run
    mislabelled as an N-GPU line is the one silent failure the first hardware run could have.  A device whose chip geometry
    is not an SPX-mode MI355X's (a partitioned mode: fewer CUs / XCDs per visible device) stays legal but is named on the
    line, next to every rank's own `chip`."""
    ids = [r.get("uuid") or r["pci_bus_id"] for r in rows]
    distinct = len(set(ids))
    if len(rows) != n_gpus:
        raise SystemExit("bench: %d report rows for %d GPUs" % (len(rows), n_gpus))
    if distinct != n_gpus and not shared_hook:
        dup = sorted({i for i in ids if ids.count(i) > 1})
        raise SystemExit("bench: --gpus %d but the ranks sit on %d distinct device(s) (shared: %s) and the fold-onto-one-GPU test "
                         "hook is off: refusing to print a folded run as an %d-GPU line" % (n_gpus, distinct, ", ".join(map(str, dup)), n_gpus))
    odd = [{"rank": r.get("rank"), "pci_bus_id": r.get("pci_bus_id"), "chip": r.get("chip")} for r in rows
           if r.get("chip") and r["chip"] != EXPECTED_CHIP]
    block = {"distinct": distinct, "expected_distinct": n_gpus, "shared_gpu_test_hook": bool(shared_hook),
             "numa_nodes": sorted({r.get("numa_node") for r in rows if r.get("numa_node") is not None}),
             "not_an_spx_mi355x": odd, "data_path_collective": None}
    block.update(extra)
    return block


def require_free_hbm(torch, index, need_bytes, what):
    """fail loudly BEFORE allocating: another tenant, a partitioned device or a leaked context on one of N devices would
    otherwise surface as an out-of-memory error from the middle of the run"""
    free, total = torch.cuda.mem_get_info(index)
    if free < need_bytes:
        raise SystemExit("bench: device %d has %.1f of %.1f GiB of HBM free, %s needs %.1f GiB" %
                         (index, free / 2**30, total / 2**30, what, need_bytes / 2**30))
    return {"free_GiB": round(free / 2**30, 1), "total_GiB": round(total / 2**30, 1)}


def traffic_for_line(log2_nt, n_gpus, allow_live, child_device=None):
    """(traffic, traffic_source, live): HBM bytes per launch of the two timed kernels for `roofline.traffic` -- measured by
    this run (two child rocprofv3 --pmc passes on rank 0's device, every buffer of this process freed first) or, when that
    is not possible / not wanted, the committed profiles/hbm_traffic.json with a label that says so.  The same code at
    every N: at N > 1 the figure is rank 0's device's (the kernels and sizes are the same on every rank)."""
    traffic, source, live = None, None, None
    if allow_live and not os.environ.get("ROCPROFILER_SDK_TOOL_LIBRARIES") and not os.environ.get("ROCP_TOOL_LIBRARIES"):
        live = measure_traffic_live(log2_nt, child_device=child_device)
        if "error" not in live:
            traffic = {"encode_bytes_per_launch": live["encode"]["hbm_bytes"], "decode_bytes_per_launch": live["decode"]["hbm_bytes"]}
            source = ("measured by this run on this box%s: child `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes "
                      "over bench/pmc_workload.py --codec-only (2 launches each), calibrated on read-only / write-only probes of "
                      "known size in the same pass (fetch x%.3f, write x%.3f)" %
                      (" (rank 0's device; same kernels and sizes on all %d)" % n_gpus if n_gpus > 1 else "",
                       live["calibration"]["fetch_scale"], live["calibration"]["write_scale"]))
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if traffic is None and os.path.exists(tpath) and log2_nt == 34:
        try:
            traffic = json.load(open(tpath))
            source = ("static%s: profiles/hbm_traffic.json, from %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                      "calibrated; bench/profile.sh) -- NOT measured by this run; kernels recorded there: %s" %
                      (", N = 1 measurement quoted on an N = %d line" % n_gpus if n_gpus > 1 else "",
                       traffic.get("source"), traffic.get("kernels", "the shipped defaults of that round")))
        except Exception:  # noqa: BLE001
            traffic = None
    return traffic, source, live


def one_device_env(parent, hip_index):
    """Environment entries that leave a child process with exactly ONE device: the parent's HIP device `hip_index`.  Hidden at
    the ROCr level (ROCR_VISIBLE_DEVICES), so that rocprofv3 itself -- which enumerates HSA agents, not HIP devices -- sets its
    counters up on that agent only; HIP then sees it as device 0.  HIP_VISIBLE_DEVICES indexes into the ROCr-visible set, so a
    list the parent was given is resolved first; entries that are not plain indices (UUIDs) fall back to filtering in HIP.
    HIP on ROCm honours CUDA_VISIBLE_DEVICES as well (torch launchers set it): when HIP_VISIBLE_DEVICES is unset it IS the index
    list, and either way the child must not inherit it -- left in place it would be applied on top of the narrowed set and a
    parent started with CUDA_VISIBLE_DEVICES=4,5 would profile physical GPU 0, somebody else's.  A value of None
    means: remove the variable from the child's environment (apply_env does)."""
    hip = [x.strip() for x in parent.get("HIP_VISIBLE_DEVICES", "").split(",") if x.strip()]
    if not hip:
        hip = [x.strip() for x in parent.get("CUDA_VISIBLE_DEVICES", "").split(",") if x.strip()]
    rocr = [x.strip() for x in parent.get("ROCR_VISIBLE_DEVICES", "").split(",") if x.strip()]
    try:
        idx = int(hip[hip_index]) if hip else hip_index
        if rocr:
            return {"ROCR_VISIBLE_DEVICES": rocr[idx], "HIP_VISIBLE_DEVICES": "0", "CUDA_VISIBLE_DEVICES": None}
        return {"ROCR_VISIBLE_DEVICES": str(idx), "HIP_VISIBLE_DEVICES": "0", "CUDA_VISIBLE_DEVICES": None}
    except (ValueError, IndexError):
        return {"HIP_VISIBLE_DEVICES": hip[hip_index] if hip_index < len(hip) else str(hip_index), "CUDA_VISIBLE_DEVICES": None}


def apply_env(env, updates):
    """env.update(updates) where a value of None removes the variable"""
    for k, v in updates.items():
        if v is None:
            env.pop(k, None)
        else:
            env[k] = v
    return env


def measure_traffic_live(log2_nt, timeout_s=90, child_device=None):
    """HBM bytes per launch of the two timed kernels, measured by THIS run on THIS box: two child processes of
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC has 4 counter slots,
    3 + 2 do not fit; counters are never combined with any other tracing domain) over bench/pmc_workload.py
    --codec-only, i.e. a read-only and a write-only probe of known size for the calibration (gfx950 reports half of
    wide coalesced reads, MI355X_MICROARCH.md section HBM) followed by the encode and decode kernels at the same
    size as the headline.  Returns the summary of bench/parse_profiles.py:traffic_summary, or {"error": ...}."""
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    sys.path.insert(0, os.path.join(ROOT, "bench"))
    import parse_profiles

    tmp = tempfile.mkdtemp(prefix="cnt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CNT_BENCH_SHARE_GPU"):
        env.pop(k, None)  # the child is a plain one-device process, whatever launched the parent
    if child_device is not None:  # N > 1: the child -- rocprofv3's counter collection included -- sees rank 0's device only
        apply_env(env, one_device_env(os.environ, child_device))
    try:
        csvs = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench", "pmc_workload.py"), "--log2-nt", str(log2_nt), "--reps", "2", "--codec-only"]
            # its own session: a timeout takes rocprofv3 AND the workload it spawned, never a GPU process left behind
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                out_text, _ = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(proc.pid, signal.SIGKILL)
                proc.communicate()
                return {"error": "rocprofv3 --pmc %s: no result within %d s (process group killed)" % (counter, timeout_s)}
            r = subprocess.CompletedProcess(cmd, proc.returncode, out_text)
            csvs[counter] = os.path.join(out, "pmc_counter_collection.csv")
            if r.returncode != 0 or not os.path.exists(csvs[counter]):
                return {"error": "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stdout[-300:])}
        return parse_profiles.traffic_summary(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"], 1 << log2_nt, "live")
    except Exception as exc:  # noqa: BLE001 -- the headline must not die with its evidence leg
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
