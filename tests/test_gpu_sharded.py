"""The sharded tier with ndev > 1 (BASELINE.json configs[4]: "chunk-sharded across 8 x MI355X ... no
collective; per-GPU outputs concatenated on the host").  The GPU box has ONE device, so these tests
load the TEST-HOOKS build of the library (tests/libcute_nt_hip_hooks.so: the product's sources and kernels
with -DCNT_TEST_HOOKS -- the product itself exports no hook) and switch on cnt_test_alias_devices(1)
(shard k -> device k % count); whatever needs no hook (ndev = 1, argument checks) runs on the product: the
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


class _ActiveLibrary:
    """forwards to whichever build is active at the time of the call: the product, or -- inside a test that took `alias` /
    `hooks_build` -- the test-hooks build"""

    def __getattr__(self, name):
        from cute_nucleotides_amd import _lib

        return getattr(_lib.lib(), name)


@pytest.fixture(scope="module")
def L():
    return _ActiveLibrary()


@pytest.fixture()
def alias(hooks_build):
    """the test-hooks build with cnt_test_alias_devices(1) for the duration of one test (an explicit call: no environment
    variable exists, and the product library has no such switch at all)"""
    assert hooks_build.cnt_test_alias_devices(1) == 0
    yield
    hooks_build.cnt_test_alias_devices(0)


def _production_path():
    """ndev == 1 needs no hook: that case runs on the PRODUCT library (the fixture's teardown restores the build)"""
    from cute_nucleotides_amd import _lib

    _lib.use_build("product")
    assert not hasattr(_lib.lib(), "cnt_test_alias_devices")


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

    from cute_nucleotides_amd import _lib, sharding

    count = torch.cuda.device_count()
    n = oracle.fill_random_acgt(40000, 3)
    out = np.zeros(1250, dtype=np.uint64)
    monkeypatch.setenv("CNT_SHARD_ALIAS_DEVICES", "1")  # round 2's environment hook is gone: a variable changes nothing
    # the PRODUCT: no hook exported, no switch inside, more shards than devices is always CNT_ENODEV
    assert _lib.active_build() == "product" and not hasattr(_lib.lib(), "cnt_test_alias_devices")
    assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, count + 1) == _lib.CNT_ENODEV
    assert sharding.alias_devices(False) is False
    with pytest.raises(RuntimeError, match="hooks"):
        sharding.alias_devices(True)
    # the test-hooks build: off by default, bounded
    prev = _lib.use_build("hooks")
    try:
        _alias_hook_checks(L, _lib, n, out, count)
    finally:
        L.cnt_test_alias_devices(0)
        _lib.use_build(prev)
    assert np.array_equal(out, oracle.n_to_bits_lut(n))
    # argument errors come before any device work, as for the unsharded calls
    assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), 1249, 4) == _lib.CNT_ECAP
    assert L.cnt_bits_to_n_sharded(_p(out), 1250, 40001, _p(n), 4) == _lib.CNT_ELEN
    assert L.cnt_bits_to_n2_sharded(_p(out), 1250, 1250 * 27 + 1, _p(n), 4) == _lib.CNT_ELEN


def _alias_hook_checks(L, _lib, n, out, count):
    assert L.cnt_test_alias_devices(0) == 0  # off by default
    assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, count + 1) == _lib.CNT_ENODEV  # production behaviour
    assert L.cnt_test_alias_devices(1) == 0
    try:
        assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, 65) == _lib.CNT_ENODEV  # the hook stops at 64 shards
        assert L.cnt_n_to_bits_sharded(_p(n), n.size, _p(out), out.size, 64) == 0
    finally:
        assert L.cnt_test_alias_devices(0) == 1


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
            assert 1 <= info["copy_threads"] <= max(1, min(6, total // ndev)), info  # 6 = a calling thread's default team
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
        _production_path()  # one real device, no hook: the product library
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
        _production_path()
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


@pytest.mark.parametrize("ndev", [1, 2, 8])
@pytest.mark.parametrize("form", ["events", "adopted_streams"])
def test_queue_runs_behind_the_callers_streams_without_any_host_sync(L, oracle, alias, ndev, form):
    """generator -> encode -> decode -> consumer on N devices with the host never waiting.
    Every shard is FILLED on a torch stream of its own that first sleeps on the device for milliseconds (so the codec ops
    are enqueued long before their input exists); the queue is ordered behind that stream on the device -- form "events":
    cnt_sharded_dev_wait_event on the producer's event, cnt_sharded_dev_record_event for the consumer stream; form
    "adopted_streams": cnt_sharded_dev_open_on_streams, everything on the caller's stream -- and a consumer copies the
    outputs away behind the queue.  The ONLY host synchronisation is at the very end.  Outputs against the oracle; the
    control without the ordering encodes the not-yet-written buffer (the test has teeth)."""
    import torch

    from cute_nucleotides_amd import devutil, sharding

    if ndev == 1:
        _production_path()
    sizes = [(1 << 22) + 13, 40000, 2048 * 7, 0, (1 << 20), 77, 16384 * 3 + 1, 1][:ndev]
    seeds = [7100 + k for k in range(ndev)]
    dev = torch.device("cuda", 0)
    d_in = [torch.zeros(s, dtype=torch.uint8, device=dev) for s in sizes]
    d_pk = [torch.zeros((s + 31) // 32, dtype=torch.int64, device=dev) for s in sizes]
    d_out = [torch.zeros(s, dtype=torch.uint8, device=dev) for s in sizes]
    sink_n = [torch.zeros(s, dtype=torch.uint8, device=dev) for s in sizes]
    sink_w = [torch.zeros((s + 31) // 32, dtype=torch.int64, device=dev) for s in sizes]
    producers = [torch.cuda.Stream(device=dev) for _ in sizes]
    consumers = producers if form == "adopted_streams" else [torch.cuda.Stream(device=dev) for _ in sizes]
    filled = [torch.cuda.Event() for _ in sizes]
    done = [torch.cuda.Event() for _ in sizes]
    for e, st in zip(done, consumers):
        e.record(st)  # torch creates the HIP event on first record: the queue re-records it below
    torch.cuda.synchronize()  # allocation and zeroing are over: from here on the host never waits until the final check

    def produce(k):
        with torch.cuda.stream(producers[k]):
            torch.cuda._sleep(40_000_000)  # the producer is late: milliseconds of device time before the fill even starts
            if sizes[k]:
                devutil.fill_random_acgt(d_in[k], seeds[k])
            filled[k].record(producers[k])

    def consume(k):
        with torch.cuda.stream(consumers[k]):
            sink_n[k].copy_(d_out[k], non_blocking=True)
            sink_w[k].copy_(d_pk[k], non_blocking=True)

    q = sharding.DevQueue(streams=producers) if form == "adopted_streams" else sharding.DevQueue(ndev, timed=True)
    assert q.ndev == ndev
    for k in range(ndev):
        produce(k)
        if form == "events":
            q.wait_event(k, filled[k])
    q.n_to_bits(d_in, d_pk)
    q.bits_to_n(d_pk, sizes, d_out)
    for k in range(ndev):
        if form == "events":
            q.record_event(k, done[k])
            consumers[k].wait_event(done[k])
        consume(k)
    still_running = not filled[0].query()  # everything is enqueued and shard 0's producer has not even finished: nobody waited
    for st in consumers:
        st.synchronize()  # the one host synchronisation, at the end of the pipeline
    assert still_running, "the host was stopped somewhere between the producer and the consumer"
    for k, s in enumerate(sizes):
        host = oracle.fill_random_acgt(s, seeds[k]) if s else np.empty(0, dtype=np.uint8)
        want = oracle.n_to_bits_lut(host) if s else np.empty(0, dtype=np.uint64)
        assert np.array_equal(sink_w[k].cpu().numpy().view(np.uint64), want), (form, ndev, k)
        assert np.array_equal(sink_n[k].cpu().numpy(), host), (form, ndev, k)
    if form == "events":
        ms = q.wait()
        assert len(ms) == ndev and all(m > 0 for m, s in zip(ms, sizes) if s)  # the batch's start event sits BEHIND the wait
        assert max(ms) < 50.0, ms  # ... so the producer's sleep is not in the ops' device times
    q.close()
    if form == "adopted_streams":  # close() left the caller's streams alone: they still work
        with torch.cuda.stream(producers[0]):
            d_out[0].fill_(65)
        producers[0].synchronize()
        assert int(d_out[0][0].item()) == 65
    elif ndev == 1:
        # control: the same pipeline WITHOUT cnt_sharded_dev_wait_event encodes the buffer before the producer wrote it.  Two HIP
        # streams that the runtime maps onto the same hardware queue run in order whatever the program says (round 6: the
        # control met exactly that once the process had created a different number of streams before it), so the control counts
        # only when the encode is SEEN to finish while the producer is still asleep -- and one of a few fresh queues must get there.
        want0 = oracle.n_to_bits_lut(oracle.fill_random_acgt(sizes[0], seeds[0]))
        raced = 0
        held = []  # the attempts' queues stay open: every further hipStreamCreate then lands on another hardware queue (torch's own
        # streams come out of a pool that exists already -- creating more of those shifts nothing)
        try:
            for attempt in range(8):
                d_in[0].zero_()
                d_pk[0].fill_(-1)
                torch.cuda.synchronize()
                q2 = sharding.DevQueue(1)
                held.append(q2)
                enc_done = torch.cuda.Event()
                enc_done.record(consumers[0])  # torch creates the HIP event on first record; the queue re-records it
                torch.cuda.synchronize()
                produce(0)
                q2.n_to_bits(d_in, d_pk)
                q2.record_event(0, enc_done)
                overtook = False
                while not filled[0].query():
                    if enc_done.query():
                        overtook = True
                        break
                q2.wait()
                torch.cuda.synchronize()
                if overtook:
                    raced += 1
                    assert not np.array_equal(d_pk[0].cpu().numpy().view(np.uint64), want0), attempt  # it packed the buffer of zeros
                    break
        finally:
            for q2 in held:
                q2.close()
        assert raced, "none of 8 fresh queues ran ahead of the sleeping producer: the control never tested anything"
    assert torch.cuda.current_device() == 0


def test_queue_ordering_entry_points_error_paths_and_the_timed_op_cap(L, alias):
    import torch

    from cute_nucleotides_amd import _lib, sharding

    q = ctypes.c_void_p()
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    two = (ctypes.c_void_p * 2)(st[0].cuda_stream, st[1].cuda_stream)
    assert L.cnt_sharded_dev_open_on_streams(2, None, 0, ctypes.byref(q)) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_open_on_streams(2, (ctypes.c_void_p * 2)(st[0].cuda_stream, None), 0, ctypes.byref(q)) == _lib.CNT_EINVAL  # never the legacy default stream
    assert L.cnt_sharded_dev_open_on_streams(65, two, 0, ctypes.byref(q)) == _lib.CNT_EINVAL  # a queue drives at most 64 shards
    assert L.cnt_sharded_dev_open_on_streams(0, two, 0, ctypes.byref(q)) == _lib.CNT_EINVAL  # the array's length IS ndev: "all devices" means nothing here
    assert L.cnt_sharded_dev_open_on_streams(2, two, 0, ctypes.byref(q)) == 0 and q.value
    # round 6: the queue ASKED both streams where they live -- created under device 0, they say 0, whatever
    # their position in the array -- and that is the device it makes current for the shard
    dev = ctypes.c_int(-1)
    for k in (0, 1):
        assert L.cnt_sharded_dev_device(q, k, ctypes.byref(dev)) == 0 and dev.value == st[k].device.index == 0
    assert L.cnt_sharded_dev_device(q, 2, ctypes.byref(dev)) == _lib.CNT_EINVAL and L.cnt_sharded_dev_device(q, 0, None) == _lib.CNT_EINVAL
    ev = torch.cuda.Event()
    ev.record()
    h = ctypes.c_void_p(ev.cuda_event)
    assert L.cnt_sharded_dev_wait_event(q, 2, h) == _lib.CNT_EINVAL and L.cnt_sharded_dev_wait_event(q, -1, h) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_wait_event(q, 0, None) == _lib.CNT_EINVAL and L.cnt_sharded_dev_record_event(q, 1, None) == _lib.CNT_EINVAL
    assert L.cnt_sharded_dev_wait_event(q, 1, h) == 0 and L.cnt_sharded_dev_record_event(q, 1, h) == 0
    assert L.cnt_sharded_dev_wait(q, None) == 0 and L.cnt_sharded_dev_close(q) == 0
    assert L.cnt_sharded_dev_wait_event(q, 0, h) == _lib.CNT_EINVAL  # closed handle
    with pytest.raises(ValueError):
        sharding.DevQueue(streams=[0])
    with pytest.raises(ValueError):
        sharding.DevQueue(3, streams=st)
    # a timed queue holds CNT_QUEUE_MAX_TIMED_OPS ops per batch: the next enqueue is CNT_ECAP and queues nothing; after a wait
    # the events are recycled (the event pool used to grow without bound)
    a = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(2, dtype=torch.int64, device="cuda")
    with sharding.DevQueue(1, timed=True) as dq:
        for _ in range(_lib.CNT_QUEUE_MAX_TIMED_OPS):
            dq.n_to_bits([a], [o])
        with pytest.raises(_lib.CuteNtError) as e:
            dq.n_to_bits([a], [o])
        assert e.value.status == _lib.CNT_ECAP
        assert len(dq.wait()) == 1 and dq.op_ms(_lib.CNT_QUEUE_MAX_TIMED_OPS - 1)[0] >= 0.0
        dq.n_to_bits([a], [o])
        dq.wait()
    with sharding.DevQueue(1) as dq:  # untimed: no per-op state, no limit
        for _ in range(_lib.CNT_QUEUE_MAX_TIMED_OPS + 8):
            dq.n_to_bits([a], [o])
        dq.wait()


def test_fused_round_trip_through_the_queue(L, oracle, alias):
    """cnt_round_trip_sharded_dev_enqueue: BASELINE.json configs[3]'s fused pass on every shard (ragged, empty, misaligned)"""
    import torch

    from cute_nucleotides_amd import sharding

    sizes = [(1 << 21) + 5, 0, 40000, 4096 * 9]
    host = [oracle.fill_random_acgt(s, 8800 + k) if s else np.empty(0, dtype=np.uint8) for k, s in enumerate(sizes)]
    bufs = [torch.zeros(s + 64, dtype=torch.uint8, device="cuda") for s in sizes]
    d_in = []
    for k, (b, h) in enumerate(zip(bufs, host)):
        v = b[k : k + h.size]
        if h.size:
            v.copy_(torch.from_numpy(h))
        d_in.append(v)
    d_pk = [torch.zeros((s + 31) // 32, dtype=torch.int64, device="cuda") for s in sizes]
    d_back = [torch.zeros(s, dtype=torch.uint8, device="cuda") for s in sizes]
    torch.cuda.synchronize()
    with sharding.DevQueue(len(sizes), timed=True) as q:
        q.round_trip(d_in, d_pk, d_back)
        ms = q.wait()
    for k, h in enumerate(host):
        want = oracle.n_to_bits_lut(h) if h.size else np.empty(0, dtype=np.uint64)
        assert np.array_equal(d_pk[k].cpu().numpy().view(np.uint64), want), k
        assert np.array_equal(d_back[k].cpu().numpy(), h), k
        assert ms[k] > 0 or h.size == 0


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
        # two shards, one device, the product library: the placement check passes (k % 1 == 0) and the library refuses
        assert _lib.active_build() == "product"
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
