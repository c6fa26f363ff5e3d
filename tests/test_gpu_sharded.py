"""The sharded tier with ndev > 1 (BASELINE.json configs[4]: "chunk-sharded across 8 x MI355X ... no
collective; per-GPU outputs concatenated on the host").  The GPU box has ONE device, so these tests
switch on the library's test support cnt_test_alias_devices(1) (shard k -> device k % count): the
partition arithmetic of cnt_*_sharded, the multi-worker pool, the empty-shard path, the ragged
last shard and the per-shard output offsets all run exactly as they would on an 8-GPU node -- only the
device binding is folded onto cuda:0.  Everything is compared with the CPU oracle bit for bit, with
guard words / bytes around every output.  The property this rests on is the reference's word
independence (n_to_bits.rs:38-43: word w depends on nt [32w, 32w+32) only)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GUARD = 8  # words / 64 bytes of sentinel on both sides of every output


@pytest.fixture(scope="module")
def L():
    from cute_nucleotides_amd import _lib

    return _lib.lib()


@pytest.fixture()
def alias(L):
    """cnt_test_alias_devices(1) for the duration of one test (an explicit call: no environment variable exists)"""
    prev = L.cnt_test_alias_devices(1)
    yield
    L.cnt_test_alias_devices(prev)


def _p(a, off_bytes=0):
    return ctypes.c_void_p(a.ctypes.data + off_bytes)


def _encode_sharded(L, n, ndev, five):
    words = (n.size + 26) // 27 if five else (n.size + 31) // 32
    buf = np.full(words + 2 * GUARD, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
    fn = L.cnt_n_to_bits2_sharded if five else L.cnt_n_to_bits_sharded
    rc = fn(_p(n) if n.size else None, n.size, _p(buf, 8 * GUARD), words, ndev)
    assert rc == 0, rc
    assert (buf[:GUARD] == 0xA5A5A5A5A5A5A5A5).all() and (buf[GUARD + words :] == 0xA5A5A5A5A5A5A5A5).all(), "guard words overwritten"
    return buf[GUARD : GUARD + words].copy()


def _decode_sharded(L, bits, length, ndev, five):
    buf = np.full(length + 128, 0x2A, dtype=np.uint8)
    fn = L.cnt_bits_to_n2_sharded if five else L.cnt_bits_to_n_sharded
    rc = fn(_p(bits) if bits.size else None, bits.size, length, _p(buf, 64), ndev)
    assert rc == 0, rc
    assert (buf[:64] == 0x2A).all() and (buf[64 + length :] == 0x2A).all(), "guard bytes overwritten"
    return buf[64 : 64 + length].copy()


# sizes chosen against the 16384-nt (13824-nt) shard granule: below one granule in total (every shard but
# the first is empty), one granule + a bit with 8 shards (6 empty), whole granules only, a ragged last
# shard, fewer granules than shards, and sizes past the host tier's zero-copy limit (2^20 nt per shard)
# so the double-buffered pipeline runs inside every worker
SIZES2 = [1, 31, 33, 5000, 16384, 16385, 3 * 16384, 16384 * 8, 16384 * 8 + 1, 100003, (1 << 22) + 13, (1 << 24) + 16384 * 3 + 77]
SIZES5 = [1, 26, 27, 28, 5000, 13824, 13825, 3 * 13824, 13824 * 8 + 5, 100003, (1 << 22) + 13, 27 * (1 << 19) + 13824 * 3 + 11]


@pytest.mark.parametrize("ndev", [2, 3, 4, 8])
def test_sharded_2bit_ndev_gt_1_matches_the_oracle(L, oracle, alias, ndev):
    from cute_nucleotides_amd import sharding

    for n_len in SIZES2:
        n = oracle.fill_random_acgt(n_len, 1000 + n_len % 997)
        want = oracle.n_to_bits_lut(n)
        got = _encode_sharded(L, n, ndev, five=False)
        assert np.array_equal(got, want), (ndev, n_len)
        back = _decode_sharded(L, got, n_len, ndev, five=False)
        assert np.array_equal(back, oracle.bits_to_n_lut(want, n_len)), (ndev, n_len)
        # decode of a prefix: len < 32 * words, shards are cut by `len`, not by the words passed
        if n_len > 40:
            part = _decode_sharded(L, got, n_len - 37, ndev, five=False)
            assert np.array_equal(part, n[: n_len - 37]), (ndev, n_len)
    # the partition really had empty and ragged shards in this sweep
    parts = [sharding.shard_range_c(16385, 8, k) for k in range(8)]
    assert parts[0] == (0, 16384) and parts[1] == (16384, 16385) and all(p == (16385, 16385) for p in parts[2:])


@pytest.mark.parametrize("ndev", [2, 3, 4, 8])
def test_sharded_5letter_ndev_gt_1_matches_the_oracle(L, oracle, alias, ndev):
    for n_len in SIZES5:
        n = oracle.fill_random_acgtn(n_len, 2000 + n_len % 991)
        want = oracle.n_to_bits2_lut(n)
        got = _encode_sharded(L, n, ndev, five=True)
        assert np.array_equal(got, want), (ndev, n_len)
        back = _decode_sharded(L, got, n_len, ndev, five=True)
        assert np.array_equal(back, oracle.bits_to_n2_lut(want, n_len)), (ndev, n_len)


def test_sharded_off_alphabet_bytes_and_lowercase(L, oracle, alias):
    """every shard goes through the same default semantics as the unsharded call: (byte>>1)&3 on ALL bytes"""
    rng = np.random.default_rng(5)
    n = rng.integers(0, 256, 16384 * 5 + 321, dtype=np.uint8)
    whole = np.empty((n.size + 31) // 32, dtype=np.uint64)
    assert L.cnt_n_to_bits(_p(n), n.size, _p(whole), whole.size) == 0
    for ndev in (2, 5, 8):
        assert np.array_equal(_encode_sharded(L, n, ndev, five=False), whole)
    low = np.frombuffer(b"acgtu" * 20000, dtype=np.uint8)
    assert np.array_equal(_encode_sharded(L, low, 4, five=False), oracle.n_to_bits_lut(low))


def test_alias_hook_is_opt_in_and_bounded(L, oracle, monkeypatch):
    import torch

    from cute_nucleotides_amd import _lib

    count = torch.cuda.device_count()
    n = oracle.fill_random_acgt(40000, 3)
    out = np.zeros(1250, dtype=np.uint64)
    assert L.cnt_test_alias_devices(0) == 0  # off by default
    monkeypatch.setenv("CNT_SHARD_ALIAS_DEVICES", "1")  # round 2's environment hook is gone: a variable changes nothing
    assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, count + 1) == _lib.CNT_ENODEV  # production behaviour
    assert L.cnt_test_alias_devices(1) == 0
    try:
        assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, 65) == _lib.CNT_ENODEV  # the hook stops at 64 shards
        assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, 64) == 0
    finally:
        assert L.cnt_test_alias_devices(0) == 1
    assert np.array_equal(out, oracle.n_to_bits_lut(n))
    # argument errors come before any device work, as for the unsharded calls
    assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), 1249, 4) == _lib.CNT_ECAP
    assert L.cnt_bits_to_n_sharded(_p(out), 1250, 40001, _p(n), 4) == _lib.CNT_ELEN
    assert L.cnt_bits_to_n2_sharded(_p(out), 1250, 1250 * 27 + 1, _p(n), 4) == _lib.CNT_ELEN


def test_worker_placement_and_copy_thread_budget(L, oracle, alias, monkeypatch):
    """worker k is bound to device k % count; its staging-copy pool is sized so that the total over
    all shards stays within CNT_SHARD_COPY_THREADS_TOTAL; when the platform names the GPU's NUMA node the
    worker is pinned to that node's CPUs (and never to CPUs outside the process's own mask)."""
    import os

    import torch

    from cute_nucleotides_amd import sharding

    count = torch.cuda.device_count()
    n = oracle.fill_random_acgt(8 * (16 << 20) + 5, 11)  # >= 16 Mi nt per shard: 4-MiB staging copies, so the pools start
    want = oracle.n_to_bits_lut(n)
    allowed = len(os.sched_getaffinity(0))
    for ndev, total in ((8, 32), (8, 8), (2, 32), (3, 2)):
        monkeypatch.setenv("CNT_SHARD_COPY_THREADS_TOTAL", str(total))
        assert np.array_equal(_encode_sharded(L, n, ndev, five=False), want)
        used = 0
        for k in range(ndev):
            info = sharding.worker_info(k)
            assert info["device"] == k % count
            assert info["numa_node"] >= -1
            assert 0 <= info["n_cpus"] <= allowed
            if info["numa_node"] < 0:
                assert info["n_cpus"] == 0  # unknown node -> not pinned
            assert 1 <= info["copy_threads"] <= max(1, min(4, total // ndev)), info
            used += info["copy_threads"]
        assert used <= max(total, ndev)
    monkeypatch.setenv("CNT_SHARD_NUMA", "0")  # only consulted when a worker (re)binds; results never depend on it
    assert np.array_equal(_encode_sharded(L, n, 8, five=False), want)


def test_sharded_calls_interleave_with_shutdown_and_other_tiers(L, oracle, alias):
    import threading

    n = oracle.fill_random_acgt((1 << 21) + 99, 21)
    want = oracle.n_to_bits_lut(n)
    for ndev in (8, 2, 5, 8):
        assert np.array_equal(_encode_sharded(L, n, ndev, five=False), want)
        assert L.cnt_shutdown() == 0  # releases the workers' streams / staging; the next call rebuilds them
    bad = []

    def caller(seed):  # concurrent sharded callers queue on the pool; unsharded host-tier calls run beside them
        m = oracle.fill_random_acgt(200000 + 1000 * seed, seed)
        w = oracle.n_to_bits_lut(m)
        for _ in range(3):
            if not np.array_equal(_encode_sharded(L, m, 2 + seed, five=False), w):
                bad.append(seed)
            one = np.empty(w.size, dtype=np.uint64)
            if L.cnt_n_to_bits(_p(m), m.size, _p(one), one.size) != 0 or not np.array_equal(one, w):
                bad.append(-seed)

    ts = [threading.Thread(target=caller, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


# ---- device-resident sharded tier ----------------------------------------------------------
@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
def test_device_resident_shards_match_the_oracle(L, oracle, alias, monkeypatch, ndev):
    """cnt_*_sharded_dev: one entry per shard, shards of different (ragged, empty, misaligned) sizes, every
    one checked against the oracle; the per-shard device times come back positive."""
    import torch

    from cute_nucleotides_amd import sharding

    if ndev == 1:
        L.cnt_test_alias_devices(0)  # the production path: one real device, no hook
    sizes = [(1 << 21) + 13, 0, 40000, 2048 * 5, 77, (1 << 20), 16384 * 3 + 1, 1][:ndev]
    for five in (False, True):
        gen = oracle.fill_random_acgtn if five else oracle.fill_random_acgt
        enc = oracle.n_to_bits2_lut if five else oracle.n_to_bits_lut
        dec = oracle.bits_to_n2_lut if five else oracle.bits_to_n_lut
        host = [gen(s, 50 + k) if s else np.empty(0, dtype=np.uint8) for k, s in enumerate(sizes)]
        bufs = [torch.zeros(s + 64, dtype=torch.uint8, device="cuda") for s in sizes]
        shards = []
        for k, (b, h) in enumerate(zip(bufs, host)):
            v = b[(k % 3) : (k % 3) + h.size]  # pointer phases 0, 1, 2: the any-alignment plan per shard
            if h.size:
                v.copy_(torch.from_numpy(h))
            shards.append(v)
        outs, ms = sharding.n_to_bits_sharded_dev(shards, five_letter=five, want_ms=True)
        assert len(ms) == len(sizes)
        for k, (o, h) in enumerate(zip(outs, host)):
            want = enc(h) if h.size else np.empty(0, dtype=np.uint64)
            assert np.array_equal(o.cpu().numpy().view(np.uint64), want), (five, ndev, k)
            assert ms[k] >= 0.0 and (ms[k] > 0.0 or h.size == 0)
        backs = sharding.bits_to_n_sharded_dev(outs, sizes, five_letter=five)
        for k, (b, h) in enumerate(zip(backs, host)):
            want = dec(enc(h), h.size) if h.size else np.empty(0, dtype=np.uint8)
            assert np.array_equal(b.cpu().numpy(), want), (five, ndev, k)
        # strict-LUT flag reaches every shard
        if not five:
            raw = [torch.from_numpy(np.random.default_rng(k).integers(0, 256, max(s, 1), dtype=np.uint8)).cuda() for k, s in enumerate(sizes)]
            outs = sharding.n_to_bits_sharded_dev(raw, strict_lut=True)
            for o, r in zip(outs, raw):
                assert np.array_equal(o.cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(r.cpu().numpy()))
    assert torch.cuda.current_device() == 0


@pytest.mark.parametrize("ndev", [1, 2, 4, 8])
def test_enqueue_only_queue_runs_steps_ahead_and_matches_the_oracle(L, oracle, alias, ndev):
    """cnt_sharded_dev_open / *_enqueue / cnt_sharded_dev_wait: three encode -> decode steps queued without a wait in
    between (each step over different inputs, decode k behind encode k on every shard's stream), one wait, every shard of
    every step against the oracle; per-op device times come back for every op and add up to the batch time."""
    import torch

    from cute_nucleotides_amd import sharding

    if ndev == 1:
        L.cnt_test_alias_devices(0)
    sizes = [(1 << 20) + 13, 0, 40000, 2048 * 5, 77, (1 << 19), 16384 * 3 + 1, 1][:ndev]
    steps = 3
    for five in (False, True):
        gen = oracle.fill_random_acgtn if five else oracle.fill_random_acgt
        enc = oracle.n_to_bits2_lut if five else oracle.n_to_bits_lut
        unit = 27 if five else 32
        host = [[gen(s, 900 + 10 * st + k) if s else np.empty(0, dtype=np.uint8) for k, s in enumerate(sizes)] for st in range(steps)]
        d_in = [[torch.from_numpy(h).cuda() if h.size else torch.empty(0, dtype=torch.uint8, device="cuda") for h in row] for row in host]
        d_pk = [[torch.zeros((s + unit - 1) // unit, dtype=torch.int64, device="cuda") for s in sizes] for _ in range(steps)]
        d_out = [[torch.zeros(s, dtype=torch.uint8, device="cuda") for s in sizes] for _ in range(steps)]
        torch.cuda.synchronize()
        with sharding.DevQueue(ndev, timed=True) as q:
            assert q.ndev == ndev
            for st in range(steps):
                q.n_to_bits(d_in[st], d_pk[st], five_letter=five)
                q.bits_to_n(d_pk[st], sizes, d_out[st], five_letter=five)
            total = q.wait()
            per_op = [q.op_ms(i) for i in range(2 * steps)]
            with pytest.raises(Exception):
                q.op_ms(2 * steps)
            for k, s in enumerate(sizes):
                assert abs(sum(op[k] for op in per_op) - total[k]) < 1e-3 * max(1.0, total[k])
                assert all(op[k] > 0.0 for op in per_op) or s == 0
            # a second batch on the same queue reuses streams and events
            q.n_to_bits(d_in[0], d_pk[1], five_letter=five)
            assert len(q.wait()) == ndev
        for st in range(steps):
            for k, h in enumerate(host[st]):
                want = enc(h) if h.size else np.empty(0, dtype=np.uint64)
                got = d_pk[st][k].cpu().numpy().view(np.uint64) if st != 1 else None
                if got is not None:
                    assert np.array_equal(got, want), (five, ndev, st, k)
                assert np.array_equal(d_out[st][k].cpu().numpy(), h), (five, ndev, st, k)
        for k, h in enumerate(host[0]):  # the second batch's encode overwrote step 1's words with step 0's
            want = enc(h) if h.size else np.empty(0, dtype=np.uint64)
            assert np.array_equal(d_pk[1][k].cpu().numpy().view(np.uint64), want)
    assert torch.cuda.current_device() == 0


def test_queues_of_several_threads_are_independent(L, oracle, alias):
    """Four host threads, each with its own two-shard queue (and, interleaved, the synchronous entry points, which use a
    queue cached per calling thread): every thread's results equal the oracle's, no thread waits on another's streams."""
    import threading

    import torch

    from cute_nucleotides_amd import sharding

    sizes = [(1 << 19) + 7, 40000]
    bad = []

    def body(tid):
        try:
            torch.cuda.set_device(0)
            host = [oracle.fill_random_acgt(s, 4000 + 10 * tid + k) for k, s in enumerate(sizes)]
            d_in = [torch.from_numpy(h).cuda() for h in host]
            d_pk = [torch.zeros((s + 31) // 32, dtype=torch.int64, device="cuda") for s in sizes]
            d_out = [torch.zeros(s, dtype=torch.uint8, device="cuda") for s in sizes]
            torch.cuda.synchronize()
            with sharding.DevQueue(2, timed=(tid % 2 == 0)) as q:
                for _ in range(5):
                    q.n_to_bits(d_in, d_pk)
                    q.bits_to_n(d_pk, sizes, d_out)
                q.wait()
            for k, h in enumerate(host):
                if not np.array_equal(d_pk[k].cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(h)) or not np.array_equal(d_out[k].cpu().numpy(), h):
                    bad.append((tid, k, "queue"))
            outs = sharding.n_to_bits_sharded_dev(d_in)  # the synchronous form: this thread's cached queue
            for k, h in enumerate(host):
                if not np.array_equal(outs[k].cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(h)):
                    bad.append((tid, k, "sync"))
        except Exception as exc:  # noqa: BLE001
            bad.append((tid, repr(exc)))

    ts = [threading.Thread(target=body, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad
    assert L.cnt_shutdown() == 0  # releases the calling thread's cached queues too


def test_enqueue_only_queue_error_paths(L, alias):
    import torch

    from cute_nucleotides_amd import _lib, sharding

    q = ctypes.c_void_p()
    assert L.cnt_sharded_dev_open(65, 0, ctypes.byref(q)) == _lib.CNT_ENODEV and not q.value
    assert L.cnt_sharded_dev_open(2, 2, ctypes.byref(q)) == _lib.CNT_EINVAL  # unknown flag
    assert L.cnt_sharded_dev_wait(ctypes.c_void_p(0x1234), None) == _lib.CNT_EINVAL  # not a live queue
    assert L.cnt_sharded_dev_open(2, 0, ctypes.byref(q)) == 0 and q.value
    a = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    o = torch.zeros(256, dtype=torch.int64, device="cuda")
    ptr = (ctypes.c_void_p * 2)(a.data_ptr(), a.data_ptr())
    n_len = (ctypes.c_size_t * 2)(4096, 4096)
    optr = (ctypes.c_void_p * 2)(o.data_ptr(), o.data_ptr() + 1024)
    caps = (ctypes.c_size_t * 2)(128, 127)
    # shard 1's capacity is short: shard 0 is already queued, the op is refused, the wait still drains shard 0
    assert L.cnt_n_to_bits_sharded_dev_enqueue(q, ptr, n_len, optr, caps, 0) == _lib.CNT_ECAP
    assert L.cnt_n_to_bits_sharded_dev_enqueue(q, None, n_len, optr, caps, 0) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_wait(q, None) == 0
    ms = (ctypes.c_float * 2)()
    assert L.cnt_sharded_dev_op_ms(q, 0, ms) == _lib.CNT_EINVAL  # not a timed queue
    assert L.cnt_sharded_dev_close(q) == 0
    assert L.cnt_sharded_dev_close(q) == _lib.CNT_EINVAL  # closed handles are refused, not dereferenced
    assert L.cnt_n_to_bits_sharded_dev_enqueue(q, ptr, n_len, optr, caps, 0) == _lib.CNT_EINVAL
    with sharding.DevQueue(2) as dq:
        with pytest.raises(ValueError):
            dq.n_to_bits([a], [o])  # one shard handed to a two-shard queue
    assert torch.cuda.current_device() == 0


def test_device_resident_shards_error_paths(L, oracle, alias):
    import torch

    from cute_nucleotides_amd import _lib, sharding

    a = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        sharding.n_to_bits_sharded_dev([a, a], outs=[torch.zeros(128, dtype=torch.int64, device="cuda"), torch.zeros(3, dtype=torch.int64, device="cuda")])
    with pytest.raises(ValueError, match="The length is greater"):
        sharding.bits_to_n_sharded_dev([torch.zeros(4, dtype=torch.int64, device="cuda")], [129])
    # the C level reports the first failing shard's status after waiting for the shards already enqueued
    ptr = (ctypes.c_void_p * 2)(a.data_ptr(), a.data_ptr())
    n_len = (ctypes.c_size_t * 2)(4096, 4096)
    o = torch.zeros(256, dtype=torch.int64, device="cuda")
    optr = (ctypes.c_void_p * 2)(o.data_ptr(), o.data_ptr() + 1024)
    caps = (ctypes.c_size_t * 2)(128, 127)
    assert L.cnt_n_to_bits_sharded_dev(ptr, n_len, optr, caps, 2, 0, None) == _lib.CNT_ECAP
    assert L.cnt_n_to_bits_sharded_dev(None, n_len, optr, caps, 2, 0, None) == _lib.CNT_EINVAL
    assert L.cnt_n_to_bits_sharded_dev(ptr, n_len, optr, caps, 65, 0, None) == _lib.CNT_ENODEV


def test_sharded_host_tier_throughput_is_sane(L, oracle, alias):
    """not a benchmark (one GPU behind all shards): eight aliased shards of a 2^28-nt buffer must not be
    pathologically slower than the unsharded call -- catches a pool that serialises or re-pins per call"""
    import time

    n = oracle.fill_random_acgt(1 << 28, 7)
    out = np.empty(1 << 23, dtype=np.uint64)
    L.cnt_n_to_bits(_p(n), n.size, _p(out), out.size)
    t0 = time.perf_counter()
    assert L.cnt_n_to_bits(_p(n), n.size, _p(out), out.size) == 0
    t_one = time.perf_counter() - t0
    want = out.copy()
    L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, 8)
    out[:] = 0
    t0 = time.perf_counter()
    assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, 8) == 0
    t_eight = time.perf_counter() - t0
    assert np.array_equal(out, want)
    assert t_eight < 4.0 * t_one + 0.05, (t_one, t_eight)


def test_single_process_sharded_bench_script(alias):
    """bench/bench_sharded_dev.py (configs[4] from one process through cnt_n_to_bits_sharded_dev): four aliased
    shards of 2^28 nt on the 1-GPU box -- the JSON line, the partition it reports, verified round trip."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(root, "bench", "bench_sharded_dev.py"), "--ndev", "4", "--log2-nt", "28", "--iters", "3",
                          "--decode", "--alias"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["ndev"] == 4 and j["alias_test_hook"] is True and j["verified"] is True and j["data_path_collective"] is None
    assert j["partition"] == [[k << 28, (k + 1) << 28] for k in range(4)]
    assert len(j["shard_ms"]) == 4 and all(m > 0 for m in j["shard_ms"]) and j["aggregate_gnts"] > 100 and j["decode_aggregate_gnts"] > 100
    assert all(d["device_index"] == 0 for d in j["devices"])


def test_sharded_dev_wrappers_validate_their_lists(L):
    """the Python wrappers of cnt_*_sharded_dev hand ctypes arrays of len(shards) entries to the C
    side, which runs shard k on device k % visible unconditionally -- a shorter `outs` / `lengths` list made the library
    read past the arrays, a tensor on another device reached the kernel as a foreign pointer.  Both are ValueErrors
    now, before anything is enqueued."""
    import torch

    from cute_nucleotides_amd import _lib, sharding

    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    w = torch.zeros(128, dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        sharding.n_to_bits_sharded_dev([d, d], outs=[w])  # one output for two shards
    with pytest.raises(ValueError):
        sharding.bits_to_n_sharded_dev([w], [4096, 4096])  # two lengths for one shard
    with pytest.raises(ValueError):
        sharding.bits_to_n_sharded_dev([w, w], [4096, 4096], outs=[d])
    if torch.cuda.device_count() == 1:
        # two shards, one device, no test support switched on: the placement check passes (k % 1 == 0) and the library refuses
        assert L.cnt_test_alias_devices(0) == 0
        with pytest.raises(_lib.CuteNtError) as e:
            sharding.n_to_bits_sharded_dev([d, d])
        assert e.value.status == _lib.CNT_ENODEV
    else:
        other = torch.zeros(4096, dtype=torch.uint8, device="cuda:1")
        with pytest.raises(ValueError):
            sharding.n_to_bits_sharded_dev([other, d])  # shard 0 must live on device 0
    (bits,) = sharding.n_to_bits_sharded_dev([d])
    assert bits.numel() == 128 and int(bits.abs().sum().item()) == 0  # zero bytes -> code 0


def test_chip_info_is_what_the_launchers_use(L):
    """CU count, LDS per CU and XCD count come from the device (hipDeviceGetAttribute), not from
    literals; on an MI355X in SPX mode they are 256 / 160 KiB / 8."""
    import ctypes

    import torch

    cus, lds, xcds = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    assert L.cnt_chip_info(0, ctypes.byref(cus), ctypes.byref(lds), ctypes.byref(xcds)) == 0
    props = torch.cuda.get_device_properties(0)
    assert cus.value == props.multi_processor_count and cus.value >= 1
    assert lds.value >= 65536 and xcds.value >= 1 and (xcds.value & (xcds.value - 1)) == 0
    if "MI355" in props.name and cus.value == 256:
        assert (lds.value, xcds.value) == (163840, 8)
    assert L.cnt_chip_info(torch.cuda.device_count(), None, None, None) != 0


def test_check_device_range_tells_device_memory_from_host_memory():
    """cnt_check_device_range: the debug aid for FFI callers of the *_dev entry points (a wrong pointer there is a GPU page
    fault, not an error code)."""
    import ctypes

    import numpy as np
    import torch

    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    d = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    p = ctypes.c_void_p
    assert L.cnt_check_device_range(p(d.data_ptr()), d.numel(), -1) == 0
    assert L.cnt_check_device_range(p(d.data_ptr() + 12345), d.numel() - 12345, 0) == 0  # interior pointers are fine
    assert L.cnt_check_device_range(p(d.data_ptr()), 1 << 40, 0) == _lib.CNT_ECAP  # runs past the allocation
    host = np.zeros(4096, dtype=np.uint8)
    assert L.cnt_check_device_range(p(host.ctypes.data), 4096, 0) == _lib.CNT_EINVAL  # pageable host memory
    assert L.cnt_check_device_range(None, 16, 0) == _lib.CNT_EINVAL
    pinned = torch.empty(4096, dtype=torch.uint8, pin_memory=True)
    assert L.cnt_check_device_range(p(pinned.data_ptr()), 4096, 0) == 0  # pinned host memory is addressable by the device
    assert L.cnt_check_device_range(p(d.data_ptr()), 16, torch.cuda.device_count()) == _lib.CNT_ENODEV
