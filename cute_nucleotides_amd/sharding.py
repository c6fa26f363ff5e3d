"""Contiguous-chunk partition of a nucleotide buffer over GPUs.

Word w depends only on nt [32w, 32w+32) (n_to_bits.rs:39-42), so the path shards into
independent units with no exchange step: GPU k of G gets nt [k*C, min(N,(k+1)*C)) with
C = ceil(N/G) rounded up to `gran` (a whole number of kernel tiles, hence of words), every
shard starts on a word boundary and only the last one has a tail.  The same arithmetic lives
in hip/sharded_tier.inc (shard_range, exported as cnt_shard_range); bench.py uses this copy for its one-process-
per-GPU launch.  No collective anywhere on the data path.
"""

SHARD_GRAN_NT = 16384  # one DIRECT/U=4 tile; multiple of 32


def shard_size(n_len, world, gran=SHARD_GRAN_NT):
    if world <= 0:
        raise ValueError("world must be positive")
    per = -(-n_len // world)
    return -(-per // gran) * gran


def partition(n_len, world, gran=SHARD_GRAN_NT):
    """[(lo, hi)] per rank, in nucleotides; empty shards are (n_len, n_len)."""
    per = shard_size(n_len, world, gran)
    return [(min(n_len, per * k), min(n_len, per * (k + 1))) for k in range(world)]


def word_range(lo, hi):
    """Packed-word range [w_lo, w_hi) a shard [lo, hi) of nucleotides owns (lo % 32 == 0)."""
    if lo % 32:
        raise ValueError("shard must start on a word boundary")
    return lo // 32, (hi + 31) // 32

SHARD_GRAN5_NT = 4 * 3456  # 5-letter codec: four 128-word tiles; multiple of 27


def shard_range_c(n_len, ndev, k, nt_per_word=32):
    """The C library's own copy of the partition (cnt_shard_range): what cnt_*_sharded use."""
    import ctypes

    from ._lib import check, lib

    lo, hi = ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(lib().cnt_shard_range(n_len, ndev, k, nt_per_word, ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def worker_info(k):
    """{device, numa_node, n_cpus, copy_threads} of sharded-tier worker k (cnt_shard_worker_info)."""
    import ctypes

    from ._lib import check, lib

    v = [ctypes.c_int(-1) for _ in range(4)]
    check(lib().cnt_shard_worker_info(k, *[ctypes.byref(x) for x in v]))
    return dict(zip(("device", "numa_node", "n_cpus", "copy_threads"), (x.value for x in v)))


def alias_devices(on):
    """TEST SUPPORT (cnt_test_alias_devices): let the sharded tiers fold ndev > visible devices onto the devices that
    exist (shard k -> device k % count).  Returns the previous setting.  Off by default; there is no environment
    variable for it."""
    from . import _lib

    if not _lib.has_test_hooks():
        if not on:
            return False  # the product has no such switch: shard k runs on device k, always
        raise RuntimeError("alias_devices is test support that the product library does not contain: switch to the test-hooks "
                           "build first (_lib.use_build('hooks'); tests: the `hooks_build` fixture)")
    return bool(_lib.lib().cnt_test_alias_devices(1 if on else 0))


def _check_placement(shards, outs, extra=None, devices=None):
    """The C side runs shard k on device k % visible unconditionally (under alias_devices; device k otherwise) -- or, for a
    queue opened on streams or on a device list, on the device the QUEUE reports for shard k (`devices`,
    cnt_sharded_dev_device): a tensor living anywhere else would reach the kernel as a foreign-device pointer and fault the
    process.  Lists of unequal length would make the library read past the ctypes arrays.  Both are ValueErrors here."""
    import ctypes

    import torch

    from ._lib import check, lib

    if len(outs) != len(shards) or (extra is not None and len(extra) != len(shards)):
        raise ValueError("shards, outs and lengths must have one entry per shard (%d, %d%s)"
                         % (len(shards), len(outs), "" if extra is None else ", %d" % len(extra)))
    if devices is not None:
        for k, (t, o) in enumerate(zip(shards, outs)):
            for x in (t, o):
                if x.device.index != devices[k]:
                    raise ValueError("shard %d lives on %s but this queue runs shard %d on cuda:%d" % (k, x.device, k, devices[k]))
        return
    count = ctypes.c_int(0)
    check(lib().cnt_device_count(ctypes.byref(count)))
    if count.value <= 0:
        raise ValueError("no HIP device visible")
    aliased = len(shards) > count.value  # only legal under alias_devices(True); the library says CNT_ENODEV otherwise
    for k, (t, o) in enumerate(zip(shards, outs)):
        want = k % count.value
        for x in (t, o):
            if x.device.index != want:
                raise ValueError("shard %d lives on %s but the library runs shard k on device k%s = cuda:%d"
                                 % (k, x.device, " %% %d" % count.value if aliased else "", want))


def _ptr_array(values):
    import ctypes

    return (ctypes.c_void_p * len(values))(*values)


def _size_array(values):
    import ctypes

    return (ctypes.c_size_t * len(values))(*values)


def n_to_bits_sharded_dev(shards, outs=None, five_letter=False, strict_lut=False, want_ms=False, tail_lut=False):
    """Device-resident sharded encode (cnt_n_to_bits[2]_sharded_dev): `shards[k]` is a uint8 CUDA
    tensor on device k (device k % visible under alias_devices(True)); returns the list of int64
    word tensors (and the per-shard device milliseconds when want_ms).  Synchronous."""
    import ctypes

    import torch

    from ._lib import check, lib
    from .n_to_bits import encode_flags

    L = lib()
    words_for = L.cnt_words2_for if five_letter else L.cnt_words_for
    words = [words_for(t.numel()) for t in shards]
    for t in shards:
        if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
            raise ValueError("shards must be contiguous uint8 CUDA tensors")
    if outs is None:
        outs = [torch.empty(w, dtype=torch.int64, device=t.device) for w, t in zip(words, shards)]
    if len(outs) != len(shards):
        raise ValueError("outs must have one entry per shard")
    for o, w, t in zip(outs, words, shards):
        if o.dtype != torch.int64 or not o.is_cuda or not o.is_contiguous() or o.device != t.device or o.numel() < w:
            raise ValueError("outs[k] must be a contiguous int64 CUDA tensor on shard k's device with >= words elements")
    _check_placement(shards, outs)
    for t in shards:  # the library enqueues on its own streams: everything queued on torch's must be done
        torch.cuda.synchronize(t.device)
    ms = (ctypes.c_float * len(shards))() if want_ms else None
    fn = L.cnt_n_to_bits2_sharded_dev if five_letter else L.cnt_n_to_bits_sharded_dev
    check(fn(_ptr_array([t.data_ptr() if t.numel() else 0 for t in shards]), _size_array([t.numel() for t in shards]),
             _ptr_array([o.data_ptr() if o.numel() else 0 for o in outs]), _size_array([o.numel() for o in outs]),
             len(shards), encode_flags(strict_lut, tail_lut), ms))
    outs = [o[:w] for o, w in zip(outs, words)]
    return (outs, list(ms)) if want_ms else outs


def bits_to_n_sharded_dev(shards, lengths, outs=None, five_letter=False, want_ms=False):
    """Device-resident sharded decode: `shards[k]` int64 words on device k, `lengths[k]` nucleotides."""
    import ctypes

    import torch

    from . import _lib
    from ._lib import check, lib

    L = lib()
    unit = 27 if five_letter else 32
    if len(lengths) != len(shards) or (outs is not None and len(outs) != len(shards)):
        raise ValueError("shards, lengths and outs must have one entry per shard")
    for t, n in zip(shards, lengths):
        if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
            raise ValueError("shards must be contiguous int64 CUDA tensors")
        if n > t.numel() * unit:
            check(_lib.CNT_ELEN)
    if outs is None:
        outs = [torch.empty(n, dtype=torch.uint8, device=t.device) for n, t in zip(lengths, shards)]
    for o, n, t in zip(outs, lengths, shards):
        if o.dtype != torch.uint8 or not o.is_cuda or not o.is_contiguous() or o.device != t.device or o.numel() < n:
            raise ValueError("outs[k] must be a contiguous uint8 CUDA tensor on shard k's device with >= length elements")
    _check_placement(shards, outs, lengths)
    for t in shards:
        torch.cuda.synchronize(t.device)
    ms = (ctypes.c_float * len(shards))() if want_ms else None
    fn = L.cnt_bits_to_n2_sharded_dev if five_letter else L.cnt_bits_to_n_sharded_dev
    check(fn(_ptr_array([t.data_ptr() if t.numel() else 0 for t in shards]), _size_array([t.numel() for t in shards]),
             _size_array(list(lengths)), _ptr_array([o.data_ptr() if o.numel() else 0 for o in outs]), len(shards), 0, ms))
    outs = [o[:n] for o, n in zip(outs, lengths)]
    return (outs, list(ms)) if want_ms else outs


class DevQueue:
    """Enqueue-only form of the device-resident sharded tier (cnt_sharded_dev_open / *_enqueue / cnt_sharded_dev_wait): one
    library stream per shard, any number of ops queued ahead, one wait.

        with sharding.DevQueue(ndev, timed=True) as q:
            for _ in range(steps):
                q.n_to_bits(d_in, d_packed)             # returns at once
                q.bits_to_n(d_packed, lens, d_out)      # queued behind the encode on every shard's stream
            ms = q.wait()                               # per-shard device milliseconds of the whole batch
            enc0 = q.op_ms(0)                           # ... and of each op

    The tensors handed to an enqueue call are kept alive until the next wait() -- wait regularly: a timed queue refuses the
    4097th op of a batch (CNT_ECAP), an untimed one has no limit and pins everything it was handed until then.  The queue's streams are the library's
    own: whatever produced a shard on a torch stream must either be complete before the ops that read it are enqueued, or
    -- no host synchronisation -- be ordered in front of them ON THE DEVICE:

        ev = torch.cuda.Event(); ev.record(producer_stream_k)   # behind the kernel that fills shard k
        q.wait_event(k, ev)                                     # shard k's stream waits for it (hipStreamWaitEvent)
        q.n_to_bits(...)                                        # runs behind the producer, the host never stopped
        done = torch.cuda.Event(); q.record_event(k, done)      # completes when shard k's queued ops have
        consumer_stream_k.wait_event(done)                      # a torch stream reads the outputs behind them

    or hand the queue the caller's streams (`DevQueue(streams=[torch.cuda.Stream(k) ...])`, cnt_sharded_dev_open_on_streams):
    shard k's ops are then enqueued on streams[k], in order with everything else the caller puts there -- the queue asks every
    stream which device it belongs to, so the streams may come in any order.  `DevQueue(devices=[2, 3, 2])` puts shard k on
    devices[k] (cnt_sharded_dev_open_on_devices: any subset, order or repetition of the visible devices).  `q.devices[k]` is
    where the queue runs shard k (cnt_sharded_dev_device): its tensors must live there."""

    def __init__(self, ndev=0, timed=False, streams=None, devices=None):
        import ctypes

        from . import _lib
        from ._lib import check, lib

        self._L = lib()
        self._h = ctypes.c_void_p()
        flags = _lib.CNT_QUEUE_TIMED if timed else 0
        self._streams = None
        explicit = streams is not None or devices is not None
        if streams is not None and devices is not None:
            raise ValueError("streams already say where every shard runs: pass streams or devices, not both")
        if streams is not None:
            if ndev not in (0, len(streams)):
                raise ValueError("ndev = %d but %d streams" % (ndev, len(streams)))
            handles = [getattr(st, "cuda_stream", st) for st in streams]  # torch.cuda.Stream or a raw hipStream_t
            if any(not h for h in handles):
                raise ValueError("a queue never adopts the legacy default stream (handle 0)")
            self._streams = list(streams)  # the caller's streams outlive the queue
            check(self._L.cnt_sharded_dev_open_on_streams(len(handles), _ptr_array(handles), flags, ctypes.byref(self._h)))
        elif devices is not None:
            if ndev not in (0, len(devices)):
                raise ValueError("ndev = %d but %d devices" % (ndev, len(devices)))
            check(self._L.cnt_sharded_dev_open_on_devices(len(devices), (ctypes.c_int * len(devices))(*[int(d) for d in devices]), flags, ctypes.byref(self._h)))
        else:
            check(self._L.cnt_sharded_dev_open(ndev, flags, ctypes.byref(self._h)))
        n = ctypes.c_int(0)
        check(self._L.cnt_sharded_dev_shards(self._h, ctypes.byref(n)))
        self.ndev, self.timed, self._keep, self.ops = n.value, timed, [], 0
        self.devices = []
        for k in range(self.ndev):
            d = ctypes.c_int(-1)
            check(self._L.cnt_sharded_dev_device(self._h, k, ctypes.byref(d)))
            self.devices.append(d.value)
        self._placement = self.devices if explicit else None  # default queues keep the k % visible rule's error text

    @property
    def handle(self):
        return self._h

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._h is not None and self._h.value:
            from ._lib import check

            h, self._h = self._h, None
            self._keep = []
            check(self._L.cnt_sharded_dev_close(h))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _own(self, shards):
        if self._h is None:
            raise ValueError("queue is closed")
        if len(shards) != self.ndev:
            raise ValueError("this queue drives %d shards, got %d" % (self.ndev, len(shards)))

    def n_to_bits(self, shards, outs, five_letter=False, strict_lut=False, tail_lut=False, invalid=None):
        """queue one encode of every shard (uint8 CUDA tensor on device k) into outs[k] (int64, >= words); returns at once.
        `invalid` (a list of device int64 scalars the caller zeroed, one per shard): the VALIDATED encode -- shard k's op also
        adds the number of its bytes outside the codec's alphabet to invalid[k], in the same pass"""
        import torch

        from ._lib import check
        from .n_to_bits import encode_flags

        self._own(shards)
        words_for = self._L.cnt_words2_for if five_letter else self._L.cnt_words_for
        for t, o in zip(shards, outs):
            if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
                raise ValueError("shards must be contiguous uint8 CUDA tensors")
            if o.dtype != torch.int64 or not o.is_cuda or not o.is_contiguous() or o.device != t.device or o.numel() < words_for(t.numel()):
                raise ValueError("outs[k] must be a contiguous int64 CUDA tensor on shard k's device with >= words elements")
        _check_placement(shards, outs, devices=self._placement)
        args = [self._h, _ptr_array([t.data_ptr() if t.numel() else 0 for t in shards]), _size_array([t.numel() for t in shards]),
                _ptr_array([o.data_ptr() if o.numel() else 0 for o in outs]), _size_array([o.numel() for o in outs]), encode_flags(strict_lut, tail_lut)]
        if invalid is None:
            fn = self._L.cnt_n_to_bits2_sharded_dev_enqueue if five_letter else self._L.cnt_n_to_bits_sharded_dev_enqueue
        else:  # the validated form: shard k's op adds its count of bytes outside the alphabet to invalid[k] (the caller zeroes it)
            self._counters(shards, invalid)
            fn = self._L.cnt_n_to_bits2_checked_sharded_dev_enqueue if five_letter else self._L.cnt_n_to_bits_checked_sharded_dev_enqueue
            args.append(_ptr_array([c.data_ptr() for c in invalid]))
        self._keep.append((shards, outs, invalid))
        check(fn(*args))
        self.ops += 1

    @staticmethod
    def _counters(shards, invalid):
        import torch

        if len(invalid) != len(shards):
            raise ValueError("invalid must have one counter per shard")
        for t, c in zip(shards, invalid):
            if c.dtype != torch.int64 or not c.is_cuda or c.device != t.device or c.numel() < 1 or not c.is_contiguous():
                raise ValueError("invalid[k] must be a contiguous int64 CUDA tensor on shard k's device")

    def bits_to_n(self, shards, lengths, outs, five_letter=False):
        """queue one decode of every shard (int64 words on device k, lengths[k] nucleotides) into outs[k] (uint8, >= length)"""
        import torch

        from . import _lib
        from ._lib import check

        self._own(shards)
        unit = 27 if five_letter else 32
        if len(lengths) != len(shards) or len(outs) != len(shards):
            raise ValueError("shards, lengths and outs must have one entry per shard")
        for t, n, o in zip(shards, lengths, outs):
            if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
                raise ValueError("shards must be contiguous int64 CUDA tensors")
            if n > t.numel() * unit:
                check(_lib.CNT_ELEN)
            if o.dtype != torch.uint8 or not o.is_cuda or not o.is_contiguous() or o.device != t.device or o.numel() < n:
                raise ValueError("outs[k] must be a contiguous uint8 CUDA tensor on shard k's device with >= length elements")
        _check_placement(shards, outs, lengths, devices=self._placement)
        fn = self._L.cnt_bits_to_n2_sharded_dev_enqueue if five_letter else self._L.cnt_bits_to_n_sharded_dev_enqueue
        self._keep.append((shards, outs))
        check(fn(self._h, _ptr_array([t.data_ptr() if t.numel() else 0 for t in shards]), _size_array([t.numel() for t in shards]),
                 _size_array(list(lengths)), _ptr_array([o.data_ptr() if o.numel() else 0 for o in outs]), 0))
        self.ops += 1

    def round_trip(self, shards, out_bits, out_n, strict_lut=False, tail_lut=False, invalid=None):
        """queue one FUSED encode + decode of every shard (cnt_round_trip_sharded_dev_enqueue): out_bits[k] = n_to_bits(shards[k]),
        out_n[k] = its decoded canonical spelling"""
        import torch

        from ._lib import check
        from .n_to_bits import encode_flags

        self._own(shards)
        if len(out_bits) != len(shards) or len(out_n) != len(shards):
            raise ValueError("shards, out_bits and out_n must have one entry per shard")
        for t, o, b in zip(shards, out_bits, out_n):
            if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
                raise ValueError("shards must be contiguous uint8 CUDA tensors")
            if o.dtype != torch.int64 or not o.is_cuda or not o.is_contiguous() or o.device != t.device or o.numel() < self._L.cnt_words_for(t.numel()):
                raise ValueError("out_bits[k] must be a contiguous int64 CUDA tensor on shard k's device with >= words elements")
            if b.dtype != torch.uint8 or not b.is_cuda or not b.is_contiguous() or b.device != t.device or b.numel() < t.numel():
                raise ValueError("out_n[k] must be a contiguous uint8 CUDA tensor on shard k's device with >= n_len elements")
        _check_placement(shards, out_bits, devices=self._placement)
        args = [self._h, _ptr_array([t.data_ptr() if t.numel() else 0 for t in shards]), _size_array([t.numel() for t in shards]),
                _ptr_array([o.data_ptr() if o.numel() else 0 for o in out_bits]), _size_array([o.numel() for o in out_bits]),
                _ptr_array([b.data_ptr() if b.numel() else 0 for b in out_n]), encode_flags(strict_lut, tail_lut)]
        fn = self._L.cnt_round_trip_sharded_dev_enqueue
        if invalid is not None:
            self._counters(shards, invalid)
            fn = self._L.cnt_round_trip_checked_sharded_dev_enqueue
            args.append(_ptr_array([c.data_ptr() for c in invalid]))
        self._keep.append((shards, out_bits, out_n, invalid))
        check(fn(*args))
        self.ops += 1

    @staticmethod
    def _event_handle(event):
        h = getattr(event, "cuda_event", event)  # torch.cuda.Event (created lazily: it must have been recorded / be recordable) or a raw hipEvent_t
        if not h:
            raise ValueError("the event has no HIP handle yet: record it (torch creates events lazily) or pass a raw hipEvent_t")
        return h

    def wait_event(self, k, event):
        """shard k's stream waits ON THE DEVICE for `event` (a recorded torch.cuda.Event or raw hipEvent_t): everything
        enqueued for shard k afterwards runs behind whatever the event covers.  The host does not wait."""
        import ctypes

        from ._lib import check

        if self._h is None:
            raise ValueError("queue is closed")
        self._keep.append((event,))
        check(self._L.cnt_sharded_dev_wait_event(self._h, k, ctypes.c_void_p(self._event_handle(event))))

    def record_event(self, k, event):
        """record `event` (created on shard k's device) on shard k's stream: it completes when everything enqueued for shard k
        so far has; a consumer stream that waits on it needs no host synchronisation.  A torch.cuda.Event must already own
        a HIP event (torch creates it on first record): record it once on any stream of that device first."""
        import ctypes

        from ._lib import check

        if self._h is None:
            raise ValueError("queue is closed")
        self._keep.append((event,))
        check(self._L.cnt_sharded_dev_record_event(self._h, k, ctypes.c_void_p(self._event_handle(event))))

    def wait(self):
        """drain every shard's stream; returns the per-shard device milliseconds of the batch (zeros unless timed)"""
        import ctypes

        from ._lib import check

        if self._h is None:
            raise ValueError("queue is closed")
        ms = (ctypes.c_float * self.ndev)()
        try:
            check(self._L.cnt_sharded_dev_wait(self._h, ms))
        finally:
            self._keep, self.last_ops, self.ops = [], self.ops, 0
        return list(ms)

    def op_ms(self, op):
        """per-shard device milliseconds of op `op` of the batch the last wait() drained (timed queues)"""
        import ctypes

        from ._lib import check

        ms = (ctypes.c_float * self.ndev)()
        check(self._L.cnt_sharded_dev_op_ms(self._h, op, ms))
        return list(ms)
