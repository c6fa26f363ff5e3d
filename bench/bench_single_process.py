"""`python bench.py --gpus N` with N > 1 and NO launcher: ONE process drives all N devices through the C ABI's
enqueue-only device-resident sharded tier (cnt_sharded_dev_open / cnt_*_sharded_dev_enqueue / cnt_sharded_dev_wait).

Shard k of the global buffer lives on device k.  The timed region queues ALL K steps -- on every shard's own stream the
decode of step s behind its encode, the encode of step s+1 behind that -- and waits ONCE, exactly like the N = 1 timed
region queues its K steps on one stream: the host's per-launch cost (SetDevice + launch + event, 10-20 us per shard and
direction) never reaches a device, because the host is a whole queue ahead.  No process group, no collective, no host
staging: north_star's "shards trivially by contiguous chunk across the 8 GPUs of one node" (word w depends on nucleotides
[32w, 32w+32) only, n_to_bits.rs:38-43 -- nothing needs a barrier).

`scaling_overhead_us` on the line: what driving N devices from one thread costs PER STEP that driving one does not --
(wall per step of the N-shard run) - ceil(N / visible devices) x (wall per step of the SAME K steps queued for shard 0
alone through a one-shard queue, same process, measured right after).  On N real devices the factor is 1 and the number
is the exposed overhead of the fan-out; with shards folded onto fewer devices (test hook) the folded shards share one
device's bandwidth, which the factor accounts for.  `host_enqueue_us_per_step` is the host time spent inside the
enqueue calls (hidden behind the devices as long as it is below the device time per step)."""
import ctypes
import json
import math
import os
import statistics
import time

from bench_measure import BYTES_PER_NT, HBM_PEAK_GBS, gbs, numpy_pack, stats_ms


def main_single_process(args):
    import numpy as np
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib, devutil, sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    N = args.gpus
    visible = torch.cuda.device_count()
    shared_gpu = os.environ.get("CNT_BENCH_SHARE_GPU") == "1"  # test support, never set by the driver: fold N shards onto the visible devices
    if visible < N:
        if not shared_gpu:
            raise SystemExit("--gpus %d: only %d HIP device(s) visible to this process (one process drives all N devices; "
                             "a torch.distributed.run launch with %d ranks works too)" % (N, visible, N))
        sharding.alias_devices(True)
    L = _lib.lib()
    n_per = 1 << args.log2_nt
    n_global = n_per * N
    parts = [sharding.shard_range_c(n_global, N, k) for k in range(N)]  # cnt_shard_range: contiguous chunks on word boundaries
    assert parts == sharding.partition(n_global, N)
    devs = [torch.device("cuda", k % visible) for k in range(N)]
    per_dev = math.ceil(N / min(N, visible))

    def sync_all():
        for i in range(min(N, visible)):
            torch.cuda.synchronize(i)

    def arrays(tensors, n=N):
        return (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])

    def sizes(values, n=N):
        return (ctypes.c_size_t * n)(*values)

    def open_queue(ndev):
        q = ctypes.c_void_p()
        _lib.check(L.cnt_sharded_dev_open(ndev, _lib.CNT_QUEUE_TIMED, ctypes.byref(q)))
        return q

    lens_l = [hi - lo for lo, hi in parts]
    words_l = [cn.n_to_bits.words_for(x) for x in lens_l]
    d_in = [torch.empty(x, dtype=torch.uint8, device=d) for x, d in zip(lens_l, devs)]
    d_pk = [torch.empty(w, dtype=torch.int64, device=d) for w, d in zip(words_l, devs)]
    d_out = [torch.empty(x, dtype=torch.uint8, device=d) for x, d in zip(lens_l, devs)]
    for t, (lo, _) in zip(d_in, parts):
        devutil.fill_random_acgt(t, args.seed, first_nt=lo)
    sync_all()
    a_in, a_pk, a_out, a_len, a_words = arrays(d_in), arrays(d_pk), arrays(d_out), sizes(lens_l), sizes(words_l)

    def queue_steps(q, steps, ins, lens, pks, words, outs):
        """enqueue `steps` encode -> decode steps without waiting; returns the host seconds spent enqueueing"""
        t = time.perf_counter()
        for _ in range(steps):
            _lib.check(L.cnt_n_to_bits_sharded_dev_enqueue(q, ins, lens, pks, words, 0))
            _lib.check(L.cnt_bits_to_n_sharded_dev_enqueue(q, pks, words, lens, outs, 0))
        return time.perf_counter() - t

    q = open_queue(N)
    if args.warmup:
        queue_steps(q, args.warmup, a_in, a_len, a_pk, a_words, a_out)
        _lib.check(L.cnt_sharded_dev_wait(q, None))
    sync_all()
    t0 = time.perf_counter()
    host_s = queue_steps(q, args.steps, a_in, a_len, a_pk, a_words, a_out)  # returns with (almost) everything still queued
    _lib.check(L.cnt_sharded_dev_wait(q, None))                             # ONE wait for all K steps on all devices
    elapsed = time.perf_counter() - t0
    ms_e, ms_d = [], []
    for k in range(args.steps):
        e, d = (ctypes.c_float * N)(), (ctypes.c_float * N)()
        _lib.check(L.cnt_sharded_dev_op_ms(q, 2 * k, e))
        _lib.check(L.cnt_sharded_dev_op_ms(q, 2 * k + 1, d))
        ms_e.append(list(e))
        ms_d.append(list(d))

    # the same K steps for shard 0 alone, through a one-shard queue: what one device costs without the fan-out
    q1 = open_queue(1)
    one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
    b_in, b_pk, b_out, b_len, b_words = one(d_in[0]), one(d_pk[0]), one(d_out[0]), sizes(lens_l[:1], 1), sizes(words_l[:1], 1)
    queue_steps(q1, max(1, args.warmup), b_in, b_len, b_pk, b_words, b_out)
    _lib.check(L.cnt_sharded_dev_wait(q1, None))
    t1 = time.perf_counter()
    queue_steps(q1, args.steps, b_in, b_len, b_pk, b_words, b_out)
    _lib.check(L.cnt_sharded_dev_wait(q1, None))
    elapsed_one = time.perf_counter() - t1
    _lib.check(L.cnt_sharded_dev_close(q1))
    scaling_overhead_us = (elapsed - per_dev * elapsed_one) / args.steps * 1e6

    verified = None
    if not args.no_verify:
        ok = all(devutil.count_mismatch(a, b) == 0 for a, b in zip(d_in, d_out))
        m = min(1 << 20, lens_l[0] // 32 * 32)
        for k in (0, N - 1):  # first and last 2^20 nt of the job against the numpy restatement of the layout
            off = 0 if k == 0 else (lens_l[k] - m) // 32 * 32
            host_n = d_in[k][off : off + m].cpu().numpy()
            got = d_pk[k][off // 32 : (off + m) // 32].cpu().numpy().view(np.uint64)
            ok = ok and bool(np.array_equal(got, numpy_pack(host_n)))
        kat = torch.from_numpy(np.frombuffer(b"ATCG" * (1 << 18), dtype=np.uint8).copy()).to(devs[0])
        kat_bits = cn.n_to_bits_dev(kat)
        ok = ok and bool((kat_bits.cpu().numpy().view(np.uint64) == np.uint64(0xD8D8D8D8D8D8D8D8)).all())
        verified = bool(ok)
        del kat, kat_bits

    # ---- BASELINE.json configs[4]: 2^35 nt per GPU through the same queue ------------------------------------------------
    del d_out, d_pk, d_in, a_in, a_pk, a_out, b_in, b_pk, b_out
    for i in range(min(N, visible)):
        with torch.cuda.device(i):
            torch.cuda.empty_cache()
    shard_rows, shard_wall = [None] * N, None
    if not args.no_extras and args.shard_log2_nt > 0:
        s_len = 1 << args.shard_log2_nt
        fits = all(torch.cuda.mem_get_info(i)[0] > per_dev * 2.3 * s_len + (4 << 30) for i in range(min(N, visible)))
        if fits:
            s_parts = [sharding.shard_range_c(s_len * N, N, k) for k in range(N)]
            s_lens = [hi - lo for lo, hi in s_parts]
            s_in = [torch.empty(x, dtype=torch.uint8, device=d) for x, d in zip(s_lens, devs)]
            s_pk = [torch.empty(x // 32, dtype=torch.int64, device=d) for x, d in zip(s_lens, devs)]
            for t, (lo, _) in zip(s_in, s_parts):
                devutil.fill_random_acgt(t, args.seed + 4, first_nt=lo)
            sync_all()
            c_in, c_pk, c_len, c_words = arrays(s_in), arrays(s_pk), sizes(s_lens), sizes([x // 32 for x in s_lens])
            _lib.check(L.cnt_n_to_bits_sharded_dev_enqueue(q, c_in, c_len, c_pk, c_words, 0))
            _lib.check(L.cnt_sharded_dev_wait(q, None))
            reps = 8
            sync_all()
            t2 = time.perf_counter()
            for _ in range(reps):  # eight encodes queued back to back, one wait
                _lib.check(L.cnt_n_to_bits_sharded_dev_enqueue(q, c_in, c_len, c_pk, c_words, 0))
            _lib.check(L.cnt_sharded_dev_wait(q, None))
            shard_wall = (time.perf_counter() - t2) / reps
            s_ms = []
            for r in range(reps):
                row = (ctypes.c_float * N)()
                _lib.check(L.cnt_sharded_dev_op_ms(q, r, row))
                s_ms.append(list(row))
            for k in range(N):
                shard_rows[k] = {"nt": s_lens[k], "first_nt": s_parts[k][0], "encode_ms": stats_ms([s_ms[r][k] for r in range(reps)]),
                                 "wall_ms_per_encode_all_ranks": round(shard_wall * 1e3, 4)}
            if verified is not None:
                backs = sharding.bits_to_n_sharded_dev(s_pk, s_lens)
                for k in range(N):
                    shard_rows[k]["round_trip_verified"] = devutil.count_mismatch(s_in[k], backs[k]) == 0
                    verified = verified and shard_rows[k]["round_trip_verified"]
                del backs
            del s_in, s_pk
        else:
            shard_rows = [{"skipped": "needs %.0f GiB of free HBM per device" % ((per_dev * 2.3 * s_len + (4 << 30)) / 2**30)}] * N
    _lib.check(L.cnt_sharded_dev_close(q))

    rows = []
    for k in range(N):
        enc_list = [ms_e[i][k] for i in range(args.steps)]
        dec_list = [ms_d[i][k] for i in range(args.steps)]
        enc_ms, dec_ms = statistics.fmean(enc_list), statistics.fmean(dec_list)
        ident = {"rank": k, "local_rank": k, "pid": os.getpid()}
        ident.update(devutil.device_identity(k % visible))
        props = torch.cuda.get_device_properties(k % visible)
        ident.update({"name": props.name, "uuid": str(getattr(props, "uuid", "")) or None, "hbm_GiB": round(props.total_memory / 2**30, 1)})
        ident.update({
            "nt": lens_l[k], "first_nt": parts[k][0], "encode_ms": stats_ms(enc_list), "decode_ms": stats_ms(dec_list),
            "encode_gnts": round(lens_l[k] / (enc_ms * 1e-3) / 1e9, 1), "decode_gnts": round(lens_l[k] / (dec_ms * 1e-3) / 1e9, 1),
            "encode_frac": round(gbs(BYTES_PER_NT * lens_l[k], enc_ms) / HBM_PEAK_GBS, 4),
            "decode_frac": round(gbs(BYTES_PER_NT * lens_l[k], dec_ms) / HBM_PEAK_GBS, 4),
            "encode_read_view_frac": round(gbs(lens_l[k], enc_ms) / HBM_PEAK_GBS, 4),
            "fused_ms_median": None, "configs4_shard": shard_rows[k]})
        rows.append(ident)

    nt_per_step = 2 * n_global
    value = nt_per_step * args.steps / elapsed / 1e9
    r0 = rows[0]
    enc_ms0, dec_ms0 = r0["encode_ms"]["mean"], r0["decode_ms"]["mean"]
    enc_slow, dec_slow = max(r["encode_ms"]["mean"] for r in rows), max(r["decode_ms"]["mean"] for r in rows)
    span = lambda key: {"min": min(r[key] for r in rows), "max": max(r[key] for r in rows)}
    roof = lambda name, ms, st: {
        "kernel": name, "bound": "hbm", "achieved": round(gbs(BYTES_PER_NT * lens_l[0], ms), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(gbs(BYTES_PER_NT * lens_l[0], ms) / HBM_PEAK_GBS, 4), "traffic": None,
        "traffic_source": "not measured at N > 1 (the N = 1 line measures it live with rocprofv3 --pmc)", "avg_kernel_ms": round(ms, 4), "kernel_ms": st,
        "frac_at_median": round(gbs(BYTES_PER_NT * lens_l[0], st["median"]) / HBM_PEAK_GBS, 4),
        "frac_at_min": round(gbs(BYTES_PER_NT * lens_l[0], st["min"]) / HBM_PEAK_GBS, 4),
        "algorithmic_bytes_per_launch": int(BYTES_PER_NT * lens_l[0]),
        "of": "device 0's shard (an event behind every op on its stream; with the ops queued back to back the interval is the kernel); all devices in roofline_over_ranks"}
    line = {
        "metric": "Gnt/s encode+decode on 16 GiB random ACGT; % HBM read roofline at 1/8 GPU",
        "value": round(value, 3), "unit": "Gnt/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": "n_to_bits encode + bits_to_n decode of a device-resident uniform random ACGT buffer, "
                        "%.3g GiB (2^%d nt) per GPU" % (n_per / 2**30, args.log2_nt),
            "nt_per_gpu": n_per, "nt_per_step": nt_per_step, "seed": hex(args.seed),
            "sharding": "contiguous chunks on word boundaries (cnt_shard_range), shard k resident on device k, no data-path collective",
            "launch": "single process, enqueue-only: all %d steps (encode + decode per shard) queued on one library stream per shard through "
                      "cnt_*_sharded_dev_enqueue, ONE cnt_sharded_dev_wait" % args.steps,
            "encode_kernel": dict(devutil.variants("encode"))[devutil.get_tuning("encode")],
            "decode_kernel": dict(devutil.variants("decode"))[devutil.get_tuning("decode")],
        },
        "scaling_overhead_us": round(scaling_overhead_us, 2),
        "scaling_overhead": {
            "per_step_us": round(scaling_overhead_us, 2), "share_of_step": round(scaling_overhead_us * 1e-6 / (elapsed / args.steps), 5),
            "host_enqueue_us_per_step": round(host_s / args.steps * 1e6, 2),
            "wall_ms_per_step_all_shards": round(elapsed / args.steps * 1e3, 4), "wall_ms_per_step_shard0_alone": round(elapsed_one / args.steps * 1e3, 4),
            "shards_per_device": per_dev,
            "definition": "wall per step over all N shards - shards_per_device x wall per step of the same K steps queued for shard 0 alone "
                          "(one-shard queue, same process): what the fan-out from one thread costs a step; host_enqueue_us_per_step is host "
                          "time inside the enqueue calls, hidden while it is below the device time per step"},
        "encode_gnts_per_gpu": round(lens_l[0] / (enc_ms0 * 1e-3) / 1e9, 3), "decode_gnts_per_gpu": round(lens_l[0] / (dec_ms0 * 1e-3) / 1e9, 3),
        "encode_gnts_all_gpus": round(n_global / (enc_slow * 1e-3) / 1e9, 3), "decode_gnts_all_gpus": round(n_global / (dec_slow * 1e-3) / 1e9, 3),
        "hbm_read_roofline_frac_encode_per_gpu": round(n_per / (enc_slow * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": dict(roof("n_to_bits (encode)", enc_ms0, r0["encode_ms"]),
                         read_only_view={"achieved": round(gbs(lens_l[0], enc_ms0), 1), "frac": round(gbs(lens_l[0], enc_ms0) / HBM_PEAK_GBS, 4)}),
        "roofline_decode": roof("bits_to_n (decode)", dec_ms0, r0["decode_ms"]),
        "roofline_over_ranks": {"encode_frac": span("encode_frac"), "decode_frac": span("decode_frac"), "encode_read_view_frac": span("encode_read_view_frac")},
        "ranks": rows,
        "devices": {"distinct": len({r.get("uuid") or r["pci_bus_id"] for r in rows}), "visible": visible, "shared_gpu_test_hook": shared_gpu and visible < N,
                    "data_path_collective": None, "control_plane": None, "processes": 1},
        "verified": verified,
        "value_definition": "nucleotides converted per second over all devices: each step encodes nt_per_gpu and decodes nt_per_gpu on every "
                            "device (nt_per_step = 2 x n_gpus x nt_per_gpu); wall clock around the K queued steps and the one wait, "
                            "device-synchronised on both sides",
    }
    sh = [r["configs4_shard"] for r in rows if r.get("configs4_shard") and "encode_ms" in r["configs4_shard"]]
    if sh:
        tot = sum(x["nt"] for x in sh)
        slow = max(x["encode_ms"]["median"] for x in sh)
        line["configs4_sharded_encode"] = {
            "what": "BASELINE.json configs[4]: n_to_bits encode of this run's share of '256 GiB over 8 GPUs' -- a 2^%d-nt (%.0f GiB) contiguous "
                    "chunk per GPU (cnt_shard_range), all devices at once from one process, no collective; at N = 8 the whole 256 GiB"
                    % (args.shard_log2_nt, 2**args.shard_log2_nt / 2**30),
            "nt_per_gpu": sh[0]["nt"], "total_GiB": round(tot / 2**30, 1), "ranks_measured": len(sh),
            "per_gpu_gnts": {"min": round(min(x["nt"] / (x["encode_ms"]["median"] * 1e-3) / 1e9 for x in sh), 1),
                             "max": round(max(x["nt"] / (x["encode_ms"]["median"] * 1e-3) / 1e9 for x in sh), 1)},
            "aggregate_gnts": round(tot / shard_wall / 1e9, 1),
            "aggregate_definition": "all devices' nucleotides / wall time per encode (8 encodes queued back to back on every shard's stream, one "
                                    "wait); per-GPU figures from each shard's own HIP events",
            "aggregate_gnts_from_slowest_rank_events": round(tot / (slow * 1e-3) / 1e9, 1),
            "per_gpu_frac": {"min": round(min(gbs(BYTES_PER_NT * x["nt"], x["encode_ms"]["median"]) for x in sh) / HBM_PEAK_GBS, 4),
                             "max": round(max(gbs(BYTES_PER_NT * x["nt"], x["encode_ms"]["median"]) for x in sh) / HBM_PEAK_GBS, 4)},
            "per_gpu_read_view_frac": {"min": round(min(gbs(x["nt"], x["encode_ms"]["median"]) for x in sh) / HBM_PEAK_GBS, 4),
                                       "max": round(max(gbs(x["nt"], x["encode_ms"]["median"]) for x in sh) / HBM_PEAK_GBS, 4)},
        }
    print(json.dumps(line), flush=True)
