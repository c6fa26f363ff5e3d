#!/usr/bin/env python3
"""A/B lab for the single-GPU HOST tier (cnt_n_to_bits / cnt_bits_to_n on host slices: PCIe-bound, never `value`), round 6:

  numa      VERDICT r05 next-3: where the CALLER sits (near / far socket, by `taskset`) x CNT_HOST_NUMA (staging ring allocated from
            the GPU's node + copy helpers pinned there) x copy-team size, reused outputs at 2^26 / 2^28 / 2^30 nt.  The input is
            allocated and first-touched by the pinned child, i.e. it lies on the caller's node.
  pipeline  VERDICT r05 next-5: chunk size (CNT_HOST_CHUNK_MI) x minimum pieces (CNT_HOST_PIECES) x copy-team size at 2^22 ... 2^30
            nt, near socket, against the same-run pinned-hipMemcpy ceiling.

Every cell is its own child process (the knobs are read once per process).  One JSON line per cell on stdout.

    python bench/host_tier_lab.py numa     > gpurun_out/host_numa.jsonl
    python bench/host_tier_lab.py pipeline > gpurun_out/host_pipeline.jsonl"""
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def node_cpus():
    """{node: [cpus]} from sysfs"""
    out = {}
    base = "/sys/devices/system/node"
    for d in sorted(os.listdir(base)):
        if d.startswith("node") and d[4:].isdigit():
            cpus = []
            for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                if not part:
                    continue
                lo, _, hi = part.partition("-")
                cpus += list(range(int(lo), int(hi or lo) + 1))
            out[int(d[4:])] = cpus
    return out


def child(sizes, reps, pin_after=None, data_cpus=None):
    import numpy as np
    import torch  # noqa: F401

    import cute_nucleotides_amd as cn  # noqa: F401
    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(1)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    rows = {}
    if pin_after:  # "the scheduler put the caller on that socket": the library meets an unrestricted thread first (its placement
        # decisions -- staging ring, helper affinity -- are taken against the full mask), THEN the calling thread alone moves
        w = np.frombuffer(b"ACGT" * (1 << 20), dtype=np.uint8)
        wb = np.empty(w.size // 32, dtype=np.uint64)
        assert L.cnt_n_to_bits(p(w), w.size, p(wb), wb.size) == 0
        os.sched_setaffinity(0, {int(c) for c in pin_after.split(",")})  # the calling thread only
    for log2 in sizes:
        m = 1 << log2
        block = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, min(m, 1 << 20), dtype=np.uint8)]
        if data_cpus:  # the INPUT is first touched from these CPUs (another node than the one the calls are made from)
            back_to = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {int(c) for c in data_cpus.split(",")})
        n = np.tile(block, m // block.size)  # allocated and first touched here: on the caller's node
        if data_cpus:
            os.sched_setaffinity(0, back_to)
        bits = np.empty(m // 32, dtype=np.uint64)
        back = np.empty(m, dtype=np.uint8)
        pin = os.environ.get("CNT_LAB_PINNED", "")  # "in", "out", "both": which of the caller's arrays are pinned (cnt_host_alloc)
        if pin:
            from cute_nucleotides_amd import pinned_empty

            if pin in ("in", "both"):
                src = n
                n = pinned_empty(m, np.uint8)
                n[:] = src
                del src
            if pin in ("out", "both"):
                bits, back = pinned_empty(m // 32, np.uint64), pinned_empty(m, np.uint8)
            if pin == "big":  # only the LARGE leg of either direction: the encode's letters (above) and the decode's output
                src = n
                n = pinned_empty(m, np.uint8)
                n[:] = src
                del src
                back = pinned_empty(m, np.uint8)
        row = {}
        for name, fn in (("enc", lambda: L.cnt_n_to_bits(p(n), m, p(bits), m // 32)), ("dec", lambda: L.cnt_bits_to_n(p(bits), m // 32, m, p(back)))):
            t0 = time.perf_counter()
            assert fn() == 0
            if not rows and name == "enc":
                row["first_call_of_the_process_us"] = round((time.perf_counter() - t0) * 1e6, 1)  # streams, staging ring, engine warm-up
            assert fn() == 0
            ts = []
            k = reps if log2 >= 28 else reps * 3
            for _ in range(k):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e6)
            row[name + "_us"] = round(statistics.median(ts), 1)
            row[name + "_min_us"] = round(min(ts), 1)
        assert np.array_equal(back, n)
        rows["2^%d" % log2] = row
    v = [ctypes.c_int(-9) for _ in range(4)]
    L.cnt_host_tier_info(*[ctypes.byref(x) for x in v])
    info = dict(zip(("device", "gpu_numa_node", "helper_cpus_pinned", "staging_node"), (x.value for x in v)))
    cpu = ctypes.CDLL(None).sched_getcpu()
    node = [d for d in os.listdir("/sys/devices/system/cpu/cpu%d" % cpu) if d.startswith("node")]
    info["caller_cpu"], info["caller_node"] = cpu, node[0] if node else None
    print(json.dumps({"rows": rows, "info": info}))


def pcie_ceiling():
    import torch

    h = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    h.zero_()
    out = {}
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name + "_GiBs"] = round(1.0 / statistics.median(ts), 2)
    return out


def duplex():
    """What the LINK gives the pipeline's own traffic with no host work at all: 1 GiB one way in 8-MiB pieces and 0.25 GiB the
    other way in 2-MiB pieces, pinned memory both ends, no kernel, no staging copies.  The host tier's per-GiB time can be no
    better than these; the one-way pinned-hipMemcpy figure the fractions are quoted against ignores the quarter-size return leg."""
    import torch

    big, small, pieces = 8 << 20, 2 << 20, 128
    hb = torch.empty(3 * big, dtype=torch.uint8, pin_memory=True).zero_()
    hs = torch.empty(3 * small, dtype=torch.uint8, pin_memory=True).zero_()
    db = torch.empty(3 * big, dtype=torch.uint8, device="cuda")
    ds = torch.empty(3 * small, dtype=torch.uint8, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(4)]
    seg = lambda t, size, k: t[(k % 3) * size:(k % 3 + 1) * size]

    def timed(body):
        for _ in range(2):
            body()
            torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            body()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return round(statistics.median(ts), 2)

    def one_way(up):
        def body():
            with torch.cuda.stream(streams[0]):
                for k in range(pieces):
                    (seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True) if up else seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True))
        return body

    def two_streams(up_big):
        def body():
            for k in range(pieces):
                with torch.cuda.stream(streams[0]):
                    (seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True) if up_big else seg(ds, small, k).copy_(seg(hs, small, k), non_blocking=True))
                with torch.cuda.stream(streams[1]):
                    (seg(hs, small, k).copy_(seg(ds, small, k), non_blocking=True) if up_big else seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True))
        return body

    def ring(up_big, nslots):
        def body():
            for k in range(pieces):
                with torch.cuda.stream(streams[k % nslots]):
                    if up_big:
                        seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True)
                        seg(hs, small, k).copy_(seg(ds, small, k), non_blocking=True)
                    else:
                        seg(ds, small, k).copy_(seg(hs, small, k), non_blocking=True)
                        seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True)
        return body

    def ring_with_kernel(up_big, nslots):
        """the tier's real shape: copy up, a kernel over the slot (~ the codec's 5 us), copy down, all on the slot's stream"""
        def body():
            for k in range(pieces):
                with torch.cuda.stream(streams[k % nslots]):
                    if up_big:
                        seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True)
                        seg(ds, small, k).copy_(seg(db, big, k)[:small])
                        seg(hs, small, k).copy_(seg(ds, small, k), non_blocking=True)
                    else:
                        seg(ds, small, k).copy_(seg(hs, small, k), non_blocking=True)
                        seg(db, big, k)[:small].copy_(seg(ds, small, k))
                        seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True)
        return body

    ev_up = [torch.cuda.Event() for _ in range(3)]
    ev_k = [torch.cuda.Event() for _ in range(3)]
    ev_down = [torch.cuda.Event() for _ in range(3)]

    def three_streams(up_big):
        """an up stream that only copies up, a kernel stream, a down stream that only copies down; events hand each slot on, and
        guard the ring (the up copy of piece k + 3 waits for kernel k, kernel k + 3 for down copy k)"""
        U, K, D = streams[0], streams[1], streams[2]

        def body():
            for k in range(pieces):
                s = k % 3
                with torch.cuda.stream(U):
                    if k >= 3:
                        U.wait_event(ev_k[s])
                    (seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True) if up_big else seg(ds, small, k).copy_(seg(hs, small, k), non_blocking=True))
                    ev_up[s].record(U)
                with torch.cuda.stream(K):
                    K.wait_event(ev_up[s])
                    if k >= 3:
                        K.wait_event(ev_down[s])
                    (seg(ds, small, k).copy_(seg(db, big, k)[:small]) if up_big else seg(db, big, k)[:small].copy_(seg(ds, small, k)))
                    ev_k[s].record(K)
                with torch.cuda.stream(D):
                    D.wait_event(ev_k[s])
                    (seg(hs, small, k).copy_(seg(ds, small, k), non_blocking=True) if up_big else seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True))
                    ev_down[s].record(D)
        return body

    def two_streams_one_event(up_big, kernel_on_up, npieces, nslots=3):
        """an up stream and a down stream; the kernel rides on one of them; one event hands a piece from up to down, one guards
        the ring (the up side of piece k + nslots waits for the down copy of piece k)"""
        U, D = streams[0], streams[1]
        eu = [torch.cuda.Event() for _ in range(nslots)]
        ed = [torch.cuda.Event() for _ in range(nslots)]

        def kernel(k):
            (seg(ds, small, k).copy_(seg(db, big, k)[:small]) if up_big else seg(db, big, k)[:small].copy_(seg(ds, small, k)))

        def body():
            for k in range(npieces):
                s = k % nslots
                with torch.cuda.stream(U):
                    if k >= nslots:
                        U.wait_event(ed[s])
                    (seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True) if up_big else seg(ds, small, k).copy_(seg(hs, small, k), non_blocking=True))
                    if kernel_on_up:
                        kernel(k)
                    eu[s].record(U)
                with torch.cuda.stream(D):
                    D.wait_event(eu[s])
                    if not kernel_on_up:
                        kernel(k)
                    (seg(hs, small, k).copy_(seg(ds, small, k), non_blocking=True) if up_big else seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True))
                    ed[s].record(D)
        return body

    if os.environ.get("CNT_LAB_DUPLEX") == "short":
        # short calls: the fixed part (fill, drain, lockstep of the slot streams) decides; microseconds per call
        res = {}
        for npieces in (8, 32, 128):
            for up_big in (True, False):
                name = "%s traffic, %d pieces: " % ("encode" if up_big else "decode", npieces)
                def ring_body(nslots, up_big=up_big, npieces=npieces):
                    def body():
                        for k in range(npieces):
                            with torch.cuda.stream(streams[k % nslots]):
                                if up_big:
                                    seg(db, big, k).copy_(seg(hb, big, k), non_blocking=True)
                                    seg(ds, small, k).copy_(seg(db, big, k)[:small])
                                    seg(hs, small, k).copy_(seg(ds, small, k), non_blocking=True)
                                else:
                                    seg(ds, small, k).copy_(seg(hs, small, k), non_blocking=True)
                                    seg(db, big, k)[:small].copy_(seg(ds, small, k))
                                    seg(hb, big, k).copy_(seg(db, big, k), non_blocking=True)
                    return body
                res[name + "slot ring, 3 slots"] = round(timed(ring_body(3)) * 1e3, 1)
                res[name + "slot ring, 2 slots"] = round(timed(ring_body(2)) * 1e3, 1)
                res[name + "up + down stream, kernel on up"] = round(timed(two_streams_one_event(up_big, True, npieces)) * 1e3, 1)
                res[name + "up + down stream, kernel on down"] = round(timed(two_streams_one_event(up_big, False, npieces)) * 1e3, 1)
                res[name + "ideal (large leg at the pinned-memcpy rate)"] = round(npieces * 8 / 1024 / 53.4 * 1e6, 1)
        print(json.dumps({"us_per_call": res}), flush=True)
        return

    out = {"ms_per_GiB": {
        "encode traffic, slot ring WITH a kernel between the copies, 3 slots": timed(ring_with_kernel(True, 3)),
        "decode traffic, slot ring WITH a kernel between the copies, 3 slots": timed(ring_with_kernel(False, 3)),
        "encode traffic, up stream / kernel stream / down stream + events": timed(three_streams(True)),
        "decode traffic, up stream / kernel stream / down stream + events": timed(three_streams(False)),
        "1 GiB up alone, 128 pieces, one stream": timed(one_way(True)),
        "1 GiB down alone, 128 pieces, one stream": timed(one_way(False)),
        "encode traffic (1 GiB up + 0.25 down), an up stream and a down stream": timed(two_streams(True)),
        "decode traffic (0.25 GiB up + 1 down), an up stream and a down stream": timed(two_streams(False)),
        "encode traffic, each slot's stream copies up then down, 3 slots": timed(ring(True, 3)),
        "decode traffic, each slot's stream copies up then down, 3 slots": timed(ring(False, 3)),
        "encode traffic, each slot's stream copies up then down, 4 slots": timed(ring(True, 4)),
        "decode traffic, each slot's stream copies up then down, 4 slots": timed(ring(False, 4)),
    }}
    print(json.dumps(out), flush=True)


def trace():
    """Where a pipelined call's time goes, stage by stage, from the calling thread's own clock (hooks build: cnt_test_host_trace):
    per size, the median over calls of the time spent waiting for a slot's stream, copying out, staging and submitting, and the
    stamps of one call."""
    import numpy as np
    import torch  # noqa: F401

    from cute_nucleotides_amd import _lib, build

    build.build_hooks()
    _lib.use_build("hooks")
    L = _lib.lib()
    rng = np.random.default_rng(1)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    for log2 in (24, 26, 28):
        m = 1 << log2
        n = np.tile(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 1 << 20, dtype=np.uint8)], m >> 20)
        bits = np.empty(m // 32, dtype=np.uint64)
        back = np.empty(m, dtype=np.uint8)
        if os.environ.get("CNT_LAB_PINNED"):
            from cute_nucleotides_amd import pinned_empty

            src, n = n, pinned_empty(m, np.uint8)
            n[:] = src
            bits, back = pinned_empty(m // 32, np.uint64), pinned_empty(m, np.uint8)
            t0 = time.perf_counter()
            for _ in range(100):
                L.cnt_host_is_pinned(p(n), m)
            print(json.dumps({"cnt_host_is_pinned_us": round((time.perf_counter() - t0) * 1e4, 2)}), flush=True)
        for name, fn in (("enc", lambda: L.cnt_n_to_bits(p(n), m, p(bits), m // 32)), ("dec", lambda: L.cnt_bits_to_n(p(bits), m // 32, m, p(back)))):
            for _ in range(3):
                assert fn() == 0
            calls = []
            for _ in range(9):
                t0 = time.perf_counter()
                fn()
                wall = (time.perf_counter() - t0) * 1e6
                tags = (ctypes.c_int * 4096)()
                us = (ctypes.c_double * 4096)()
                k = L.cnt_test_host_trace(tags, us, 4096)
                calls.append((wall, [(tags[i], us[i]) for i in range(k)]))
            calls.sort(key=lambda c: c[0])
            wall, st = calls[len(calls) // 2]
            spent = {1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0}
            st = [x for x in st if x[0] <= 4]  # 5 / 6: which sides were pinned, not stamps
            for (_, a), (tag, b) in zip(st, st[1:]):
                spent[tag] += b - a
            print(json.dumps({"log2_nt": log2, "call": name, "wall_us": round(wall, 1), "loop_us": round(st[-1][1], 1),
                              "before_loop_us": round(wall - st[-1][1], 1),
                              "waiting_for_slot_us": round(spent[1], 1), "copy_out_us": round(spent[2], 1), "staging_us": round(spent[3], 1),
                              "submitting_us": round(spent[4], 1), "pieces": sum(1 for t, _ in st if t == 4),
                              "stamps": [[t, round(u, 1)] for t, u in st][:80]}), flush=True)


def run_cell(env, cpus, sizes, reps, pin_after=None, data_cpus=None):
    cmd = [sys.executable, os.path.abspath(__file__), "child", ",".join(map(str, sizes)), str(reps)]
    if pin_after or data_cpus:
        cmd.append(",".join(map(str, pin_after)) if pin_after else "-")
    if data_cpus:
        cmd.append(",".join(map(str, data_cpus)))
    if cpus:
        cmd = ["taskset", "-c", ",".join(map(str, cpus))] + cmd
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not line:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(line[-1])


def main():
    mode = sys.argv[1]
    if mode == "child":
        return child([int(x) for x in sys.argv[2].split(",")], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None,
                     sys.argv[5] if len(sys.argv) > 5 else None)
    if mode == "duplex_child":
        return duplex()
    if mode == "trace":
        return trace()
    if mode == "traces":  # the same under several settings, one process each
        for env in ({}, {"CNT_HOST_COPY_THREADS": "6"}, {"CNT_HOST_COPY_THREADS": "8"}, {"CNT_HOST_BLOCK_KI": "512"}, {"CNT_HOST_BLOCK_KI": "2048"},
                    {"CNT_HOST_NUMA": "0"}, {"CNT_HOST_COPY_THREADS": "2"}):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "trace"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
            for l in r.stdout.splitlines():
                if l.startswith("{"):
                    d = json.loads(l)
                    d.pop("stamps")
                    print(json.dumps(dict(d, env=env)), flush=True)
        return
    import torch

    from cute_nucleotides_amd import devutil

    ident = devutil.device_identity(0)
    nodes = node_cpus()
    gpu_node = ident["numa_node"]
    print(json.dumps({"device": ident, "nodes": {k: len(v) for k, v in nodes.items()}, "pcie": pcie_ceiling()}), flush=True)
    del torch
    near = nodes.get(gpu_node)
    far = next((v for k, v in sorted(nodes.items()) if k != gpu_node and v), None)
    if mode == "duplex":
        for _ in range(3):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "duplex_child"], capture_output=True, text=True, timeout=600)
            print(r.stdout.strip() or json.dumps({"error": r.stderr[-400:]}), flush=True)
    elif mode == "numa":
        cells = []
        for where, cpus in (("near", near), ("far", far), ("anywhere", None)):
            if where != "anywhere" and not cpus:
                continue
            for numa in ("1", "0"):
                for threads in ("4", "8"):
                    cells.append((where, cpus, {"CNT_HOST_NUMA": numa, "CNT_HOST_COPY_THREADS": threads}))
        for rnd in range(2):  # every cell twice, interleaved: one drifting box cannot favour a setting
            for where, cpus, env in cells:
                out = run_cell(env, cpus, (26, 28, 30), 7)
                print(json.dumps(dict(out, caller=where, env=env, round=rnd)), flush=True)
    elif mode == "numa2":
        # the far-socket caller WITHOUT a process-wide taskset (the BENCH_r05 situation: the scheduler's choice): only the calling
        # thread sits on the far socket, so CNT_HOST_NUMA=1 can keep the copy helpers next to the staging ring
        for rnd in range(3):
            for where, cpus in (("near", near), ("far", far)):
                for numa in ("1", "0"):
                    for threads in ("4",):
                        env = {"CNT_HOST_NUMA": numa, "CNT_HOST_COPY_THREADS": threads}
                        out = run_cell(env, None, (26, 28, 30), 7, pin_after=cpus)
                        print(json.dumps(dict(out, caller_thread=where, env=env, round=rnd)), flush=True)
    elif mode == "threads":
        for rnd in range(2):
            for threads in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("2", "3", "4", "5", "6", "8", "12")):
                out = run_cell({"CNT_HOST_COPY_THREADS": threads}, None, (22, 24, 26, 28, 30), 5)
                print(json.dumps(dict(out, env={"CNT_HOST_COPY_THREADS": threads}, round=rnd)), flush=True)
    elif mode == "blocks":
        for rnd in range(2):
            for threads in ("4", "6", "8"):
                for ki in ("1024", "512", "256"):
                    env = {"CNT_HOST_COPY_THREADS": threads, "CNT_HOST_BLOCK_KI": ki}
                    out = run_cell(env, None, (24, 26, 28, 30), 5)
                    print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    elif mode == "pinned":
        # the caller's arrays in pinned memory: no staging copy on that side.  "in" pins the encode's letters (and nothing of the
        # decode), "out" both outputs, "both" everything -- so the decode's packed INPUT is the encode's pinned output
        for rnd in range(3):
            for pin in ((sys.argv[2].split(",") if len(sys.argv) > 2 else ("", "both", "in", "out"))):
                pin = "" if pin == "none" else pin
                out = run_cell({"CNT_LAB_PINNED": pin}, None, (21, 22, 24, 25, 26, 27, 28, 30), 7)
                print(json.dumps(dict(out, pinned=pin or "none", round=rnd)), flush=True)
    elif mode == "pinned_ramp":
        # both sides pinned: nothing staggers the three slot streams, they copy in lockstep; do ramped pieces pull them apart?
        for rnd in range(3):
            for ramp in ("0", "22"):
                for slots in ("3", "2"):
                    env = {"CNT_LAB_PINNED": "both", "CNT_HOST_RAMP": ramp, "CNT_HOST_SLOTS": slots}
                    out = run_cell(env, None, (22, 24, 25, 26, 27, 28, 30), 7)
                    print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    elif mode == "slots2":
        for rnd in range(3):
            for pin in ("", "both"):
                for slots in ("3", "4"):
                    env = {"CNT_HOST_SLOTS": slots, "CNT_LAB_PINNED": pin}
                    out = run_cell(env, None, (21, 22, 24, 26, 28, 30), 7)
                    print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    elif mode == "small_blocks":
        # the SMALL leg's copies (2 MiB per 8-Mi-nt piece) in 256-KiB blocks instead of two 1-MiB ones: more of the team on them
        for rnd in range(3):
            for ki in ("1024", "2048", "4096"):
                env = {"CNT_HOST_SMALL_COPY_KI": ki}
                out = run_cell(env, None, (22, 24, 26, 28, 30), 5)
                print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    elif mode == "far_slots":
        # the bench line's unlucky case: the calling thread AND its arrays on the far socket; three against four slots, team 4 / 6
        for rnd in range(3):
            for where, cpus in (("far", far), ("near", near)):
                for slots, threads in ([("4", t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else (("4", "4"), ("3", "4"), ("4", "6"))):
                    env = {"CNT_HOST_SLOTS": slots, "CNT_HOST_COPY_THREADS": threads}
                    out = run_cell(env, None, (26, 28, 30), 5, pin_after=cpus)
                    print(json.dumps(dict(out, caller_thread=where, env=env, round=rnd)), flush=True)
    elif mode == "zerocopy_max":
        # ordinary memory: up to which size is memcpy + ONE kernel over the link + memcpy better than the four-slot pipeline?
        for rnd in range(3):
            for mx in ("20", "21", "22", "23"):
                env = {"CNT_ZEROCOPY_MAX_NT": str(1 << int(mx))}
                out = run_cell(env, None, (19, 20, 21, 22, 23), 7)
                print(json.dumps(dict(out, env=env, zero_copy_max_log2=mx, round=rnd)), flush=True)
    elif mode == "big_leg":
        for rnd in range(3):
            for ramp in ("default", "0"):
                env = {"CNT_LAB_PINNED": "big"}
                if ramp != "default":
                    env["CNT_HOST_RAMP"] = ramp
                out = run_cell(env, None, (22, 24, 25, 26, 27, 28, 30), 5)
                print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    elif mode == "mixed_ramp":
        # ONE side pinned: does the ramp (and which slot count) help when only the large leg goes unstaged?
        for rnd in range(2):
            for pin in ("in", "out"):
                for ramp, slots in (("0", "4"), ("25", "4"), ("25", "3"), ("0", "3")):
                    env = {"CNT_LAB_PINNED": pin, "CNT_HOST_RAMP": ramp, "CNT_HOST_SLOTS": slots}
                    out = run_cell(env, None, (24, 26, 28, 30), 5)
                    print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    elif mode == "direct_max":
        # both slices pinned: up to which size is ONE kernel over the link (no copies at all) better than the pipeline?
        for rnd in range(3):
            for mx in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "20", "21", "22", "23")):
                env = {"CNT_LAB_PINNED": "both", "CNT_DIRECT_MAX_NT": str((1 << int(mx)) if mx != "0" else 0)}
                out = run_cell(env, None, (16, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27), 7)
                print(json.dumps(dict(out, env=env, direct_max_log2=mx, round=rnd)), flush=True)
    elif mode == "slots":
        # ordinary memory, 2 against 3 (and 4) slots at the small pipelined sizes
        for rnd in range(3):
            for slots in ("3", "2", "4"):
                out = run_cell({"CNT_HOST_SLOTS": slots}, None, (21, 22, 23, 24, 25, 26, 28), 7)
                print(json.dumps(dict(out, env={"CNT_HOST_SLOTS": slots}, round=rnd)), flush=True)
    elif mode == "ramp":
        # piece sizes ramped up and down (chunk/4, chunk/2, full ..., chunk/2, chunk/4) against equal pieces; the value is the
        # smallest log2(nt) that is ramped
        for rnd in range(3):
            for ramp in ("0", "22", "25"):
                out = run_cell({"CNT_HOST_RAMP": ramp}, None, (22, 23, 24, 25, 26, 27, 28, 30), 7)
                print(json.dumps(dict(out, env={"CNT_HOST_RAMP": ramp}, round=rnd)), flush=True)
    elif mode == "fardata":
        # the calling thread next to the GPU, its INPUT on the other socket (BENCH-style: the array was made before the scheduler moved
        # the thread): do helpers pinned to the GPU's node (remote reads by the whole team) lose against helpers left alone?
        for rnd in range(3):
            for numa in ("1", "0"):
                for where, data in (("input near", None), ("input far", far)):
                    out = run_cell({"CNT_HOST_NUMA": numa}, None, (20, 26, 30), 5, pin_after=near, data_cpus=data)
                    print(json.dumps(dict(out, data=where, env={"CNT_HOST_NUMA": numa}, round=rnd)), flush=True)
    elif mode == "history":
        # does the size of the calls a process made BEFORE decide how fast its 1-GiB call runs?  (the staging ring grows on demand)
        for rnd in range(2):
            for warm, pre in (("0", "0"), ("1", "1")):
                for sizes in ((30,), (22, 30), (20, 22, 30), (21, 30)):
                    out = run_cell({"CNT_HOST_PREALLOC": pre, "CNT_HOST_WARM": warm}, near, sizes, 5)
                    print(json.dumps(dict(out, sizes_in_order=list(sizes), prealloc=pre, warm=warm, round=rnd)), flush=True)
    elif mode == "nt":
        # non-temporal stores for the staging copies (bit 0) / the copy-outs (bit 1), against plain memcpy; 1 GiB and 64 MiB
        for rnd in range(3):
            for nt in ("0", "1", "3", "2"):
                out = run_cell({"CNT_HOST_NT": nt}, near, (30, 26), 5)
                print(json.dumps(dict(out, env={"CNT_HOST_NT": nt}, round=rnd)), flush=True)
    elif mode == "pipeline":
        for rnd in range(2):
            for threads in ("4", "8"):
                for chunk in ("16", "8"):
                    for pieces in ("4", "8", "16"):
                        env = {"CNT_HOST_COPY_THREADS": threads, "CNT_HOST_CHUNK_MI": chunk, "CNT_HOST_PIECES": pieces}
                        out = run_cell(env, near, (22, 24, 26, 27, 28, 30), 7)
                        print(json.dumps(dict(out, env=env, round=rnd)), flush=True)
    else:
        raise SystemExit("usage: host_tier_lab.py numa | pipeline")


if __name__ == "__main__":
    main()
