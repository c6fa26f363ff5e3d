"""GPU parity tests for the VALIDATED encode (round 6): cnt_n_to_bits_checked_dev, cnt_round_trip_checked_dev,
cnt_n_to_bits2_checked_dev, their host-tier and sharded-queue forms.  One pass over the ASCII does what cnt_validate_dev + the
encoder did in two: the words are bit-exact what the unchecked entry point writes (hence n_to_bits_lut's in strict mode on ANY
bytes, n_to_bits.rs:8-21,34-47) and the counter grows by exactly oracle.validate(n) -- every byte of the call counted by exactly
one tile or edge item, at every alignment phase, in all three flag modes.

How "exactly one" is shown without planting one bad byte at a time: for every bit b of the byte index the call is run on the
input whose bad bytes are the positions with bit b set, and on its complement.  A position that is missed, or counted twice,
makes one count of every pair wrong; two faulty positions differ in some bit and then land in different halves -- they cannot
cancel in every pair.  2 x ceil(log2(n)) launches stand in for n."""
import ctypes

import numpy as np
import pytest

from conftest import need_free_hbm

pytestmark = pytest.mark.gpu

VALID = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
VALID5 = np.frombuffer(b"ACGTUNacgtun", dtype=np.uint8)
BAD = np.frombuffer(b"N\x00\xff\n@Bn>x\x7f\x80 ", dtype=np.uint8)  # never letters of the 2-bit alphabet
BAD5 = np.frombuffer(b"\x00\xff\n@B>x\x7f\x80 MO", dtype=np.uint8)  # ... nor of the 5-letter one
MODES = [dict(), dict(strict_lut=True), dict(tail_lut=True)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "the gpu tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def cn():
    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import _lib

    _lib.lib()
    return cn


@pytest.fixture()
def tuning(lab_build):
    from cute_nucleotides_amd import devutil

    saved = {k: devutil.get_tuning(k) for k in ("small_nt", "launch_tiles")}
    yield devutil
    for k, v in saved.items():
        devutil.set_tuning(k, v)


def _mixed(n_len, seed, five=False):
    """valid letters with ~3 % strays, the first and the last byte among them"""
    rng = np.random.default_rng(seed)
    ok, bad = (VALID5, BAD5) if five else (VALID, BAD)
    n = ok[rng.integers(0, ok.size, n_len)].copy()
    where = rng.random(n_len) < 0.03
    where[[0, -1]] = True
    n[where] = bad[rng.integers(0, bad.size, int(where.sum()))]
    return n


def _bit_split_inputs(n_len, seed, five=False):
    """[(input, number of bad bytes)]: nothing bad, everything bad, and for every bit of the index the two halves it cuts"""
    rng = np.random.default_rng(seed)
    ok, bad = (VALID5, BAD5) if five else (VALID, BAD)
    clean = ok[rng.integers(0, ok.size, n_len)]
    dirty = bad[rng.integers(0, bad.size, n_len)]
    idx = np.arange(n_len)
    out = [(clean, 0), (dirty, n_len)]
    for b in range(max(1, int(n_len - 1).bit_length())):
        for half in (0, 1):
            pick = ((idx >> b) & 1) == half
            out.append((np.where(pick, dirty, clean), int(pick.sum())))
    return out


SIZES = [1, 3, 31, 32, 33, 64, 255, 2047, 2048, 2049, 4095, 4096, 4097, 16384 + 5, 65536, 100003, (1 << 17) + 2048 * 3 + 149, (1 << 20) + 13, (1 << 22) + 16384 + 31]


@pytest.mark.parametrize("n_len", SIZES)
def test_checked_encode_writes_the_unchecked_words_and_counts_what_validate_counts(cn, oracle, torch_cuda, n_len):
    torch = torch_cuda
    for seed, n in ((1, _mixed(n_len, n_len)), (2, np.random.default_rng(n_len).integers(0, 256, n_len, dtype=np.uint8)), (3, VALID[np.random.default_rng(5).integers(0, 10, n_len)])):
        d = torch.from_numpy(n).cuda()
        want_bad = oracle.validate(n)
        for mode in MODES:
            words, acc = cn.n_to_bits_checked_dev(d, **mode)
            assert torch.equal(words, cn.n_to_bits_dev(d, **mode)), (n_len, seed, mode)
            assert int(acc.item()) == want_bad, (n_len, seed, mode)
            cn.n_to_bits_checked_dev(d, acc=acc, **mode)  # the call ADDS: the caller owns (and zeroes) the counter
            assert int(acc.item()) == 2 * want_bad
        got = cn.n_to_bits_checked_dev(d, strict_lut=True)[0].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, oracle.n_to_bits_lut(n)), (n_len, seed)  # n_to_bits_lut itself, on any bytes
        # CNT_SPREAD_COUNT: 2048 counters, workgroup b adds to slot b % 2048, the count is their sum -- same words, same count
        w_s, acc_s = cn.n_to_bits_checked_dev(d, spread=True)
        assert acc_s.numel() == 2048 and int(acc_s.sum().item()) == want_bad and torch.equal(w_s, cn.n_to_bits_dev(d)), (n_len, seed)
        if n_len >= 65536 and seed != 3:
            assert int((acc_s != 0).sum().item()) > 1  # ... and it really is spread
        _, _, acc_r = cn.round_trip_checked_dev(d, spread=True)
        assert int(acc_r.sum().item()) == want_bad
    with pytest.raises(Exception):
        cn.n_to_bits_checked_dev(d, acc=torch.zeros(1, dtype=torch.int32, device="cuda"))


def test_every_byte_is_counted_by_exactly_one_tile_or_edge_item_at_every_alignment(cn, oracle, torch_cuda, tuning):
    """stream kernel (input on a line), window kernel (every other byte phase: the tile's own bytes start `phase` into its first
    row and end `phase` into the read-ahead row), head words in front, ragged end behind, below and above the generic-only
    threshold: bit-split inputs, guard words around the output, all three flag modes"""
    torch = torch_cuda
    tuning.set_tuning("small_nt", 0)
    sizes = [4096 * 2 + 144 + 5, 512 + 4096 + 143, 512 + 4096 + 145, 2048 * 5 + 17, 40000 + 13]
    ibuf = torch.zeros(max(sizes) + 256, dtype=torch.uint8, device="cuda")
    obuf = torch.empty(max(sizes) // 32 + 64, dtype=torch.int64, device="cuda")
    for n_len in sizes:
        cases = _bit_split_inputs(n_len, n_len)
        words = (n_len + 31) // 32
        acc = torch.zeros(len(cases), dtype=torch.int64, device="cuda")
        for io in (0, 1, 5, 15, 16, 17, 48, 64, 100, 127):
            for oo in (0, 3, 15) if io in (0, 5, 127) else (1,):
                for mode in MODES if io in (0, 17) else MODES[:1]:
                    acc.zero_()
                    view, out = ibuf[io : io + n_len], obuf[8 + oo : 8 + oo + words]
                    for i, (n, _) in enumerate(cases):
                        view.copy_(torch.from_numpy(n))
                        obuf.fill_(-1)
                        cn.n_to_bits_checked_dev(view, out=out, acc=acc[i : i + 1], **mode)
                        if i < 2 or i == len(cases) - 1:  # words and guards on a few of them (the unchecked matrix covers the rest)
                            got = obuf.cpu().numpy()
                            assert (got[: 8 + oo] == -1).all() and (got[8 + oo + words :] == -1).all(), (n_len, io, oo)
                            ref = cn.n_to_bits_dev(view, **mode).cpu().numpy()
                            assert np.array_equal(got[8 + oo : 8 + oo + words], ref), (n_len, io, oo, mode)
                    assert acc.cpu().tolist() == [c for _, c in cases], (n_len, io, oo, mode)


def test_checked_encode_on_the_product_build_off_the_grid(cn, oracle, torch_cuda):
    """the same paths as shipped (no knob): sizes past the small-input path, input byte phase x output word phase"""
    from cute_nucleotides_amd import _lib

    assert not _lib.is_lab_build()
    torch = torch_cuda
    for n_len in ((1 << 17) + 2048 * 3 + 149, (1 << 18) + 100003):
        cases = _bit_split_inputs(n_len, 3 * n_len)
        ibuf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
        acc = torch.zeros(len(cases), dtype=torch.int64, device="cuda")
        for io, oo in ((0, 0), (1, 0), (16, 1), (33, 7), (127, 15)):
            acc.zero_()
            view = ibuf[io : io + n_len]
            obuf = torch.empty(n_len // 32 + 32, dtype=torch.int64, device="cuda")
            for i, (n, _) in enumerate(cases):
                view.copy_(torch.from_numpy(n))
                cn.n_to_bits_checked_dev(view, out=obuf[oo : oo + (n_len + 31) // 32], acc=acc[i : i + 1])
            assert acc.cpu().tolist() == [c for _, c in cases], (n_len, io, oo)
            assert np.array_equal(obuf[oo : oo + (n_len + 31) // 32].cpu().numpy().view(np.uint64), cn.n_to_bits_dev(view).cpu().numpy().view(np.uint64))


def test_fused_round_trip_checked_at_any_alignment_of_its_three_pointers(cn, oracle, torch_cuda, tuning):
    torch = torch_cuda
    tuning.set_tuning("small_nt", 0)
    for n_len in (4096 * 3 + 400 + 7, 4096 + 25 * 16 + 1, 3000, 65536 + 31):
        cases = _bit_split_inputs(n_len, 7 * n_len)
        words = (n_len + 31) // 32
        ibuf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
        pbuf = torch.empty(words + 64, dtype=torch.int64, device="cuda")
        bbuf = torch.empty(n_len + 512, dtype=torch.uint8, device="cuda")
        acc = torch.zeros(len(cases), dtype=torch.int64, device="cuda")
        for io, po, bo in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (5, 3, 77), (127, 15, 127), (16, 8, 64), (100, 7, 33)):
            for mode in MODES if (io, po, bo) in ((0, 0, 0), (5, 3, 77)) else MODES[:1]:
                acc.zero_()
                view = ibuf[io : io + n_len]
                pk, back = pbuf[16 + po : 16 + po + words], bbuf[128 + bo : 128 + bo + n_len]
                for i, (n, _) in enumerate(cases):
                    view.copy_(torch.from_numpy(n))
                    pbuf.fill_(-1)
                    bbuf.fill_(0x2A)
                    cn.round_trip_checked_dev(view, out_bits=pk, out_n=back, acc=acc[i : i + 1], **mode)
                    if i < 2 or i == len(cases) - 1:
                        rb, rn = cn.round_trip_dev(view, **mode)
                        gp, gb = pbuf.cpu().numpy(), bbuf.cpu().numpy()
                        assert (gp[: 16 + po] == -1).all() and (gp[16 + po + words :] == -1).all() and (gb[: 128 + bo] == 0x2A).all() and (gb[128 + bo + n_len :] == 0x2A).all()
                        assert np.array_equal(gp[16 + po : 16 + po + words], rb.cpu().numpy()) and np.array_equal(gb[128 + bo : 128 + bo + n_len], rn.cpu().numpy()), (n_len, io, po, bo)
                assert acc.cpu().tolist() == [c for _, c in cases], (n_len, io, po, bo, mode)


def test_5letter_checked_encode_counts_against_its_own_alphabet(cn, oracle, torch_cuda, tuning):
    """N and n are letters here (n_to_bits2.rs:8-23): count == validate(allow_n), words == the unchecked encoder's, wave tiles
    (3456 B: the fourth 16-B row is partial), the window kernel at every kind of byte phase, edge words"""
    torch = torch_cuda
    tuning.set_tuning("small_nt", 0)
    for n_len in (3456 * 3 + 128 + 11, 3456 + 127, 27 * 5 + 4, 100003):
        cases = _bit_split_inputs(n_len, 11 * n_len, five=True)
        words = (n_len + 26) // 27
        ibuf = torch.zeros(n_len + 256, dtype=torch.uint8, device="cuda")
        obuf = torch.empty(words + 64, dtype=torch.int64, device="cuda")
        acc = torch.zeros(len(cases), dtype=torch.int64, device="cuda")
        for io, oo in ((0, 0), (1, 0), (5, 3), (16, 1), (17, 15), (100, 7), (127, 0)):
            for mode in MODES if io in (0, 5) else MODES[:1]:
                acc.zero_()
                view, out = ibuf[io : io + n_len], obuf[8 + oo : 8 + oo + words]
                for i, (n, c) in enumerate(cases):
                    view.copy_(torch.from_numpy(n))
                    obuf.fill_(-1)
                    cn.n_to_bits2_checked_dev(view, out=out, acc=acc[i : i + 1], **mode)
                    if i < 2 or i == len(cases) - 1:
                        assert oracle.validate(n, allow_n=True) == c
                        got = obuf.cpu().numpy()
                        assert (got[: 8 + oo] == -1).all() and (got[8 + oo + words :] == -1).all()
                        assert np.array_equal(got[8 + oo : 8 + oo + words], cn.n_to_bits2_dev(view, **mode).cpu().numpy()), (n_len, io, oo, mode)
                assert acc.cpu().tolist() == [c for _, c in cases], (n_len, io, oo, mode)
    n = _mixed(1 << 20, 9, five=True)
    d = torch.from_numpy(n).cuda()
    assert int(cn.n_to_bits2_checked_dev(d, spread=True)[1].sum().item()) == oracle.validate(n, allow_n=True)
    w, acc = cn.n_to_bits2_checked_dev(d, strict_lut=True)
    assert np.array_equal(w.cpu().numpy().view(np.uint64), oracle.n_to_bits2_lut(n)) and int(acc.item()) == oracle.validate(n, allow_n=True)
    assert int(cn.n_to_bits_checked_dev(d)[1].item()) == oracle.validate(n)  # the 2-bit encoder counts the N's too


def test_checked_calls_walk_the_several_launch_loop(cn, oracle, torch_cuda, tuning):
    """every launch of a call adds to the same counter, the edges ride in the last one"""
    torch = torch_cuda
    tuning.set_tuning("small_nt", 0)
    tuning.set_tuning("launch_tiles", 64)
    n_len = 2048 * 64 * 3 + 2048 * 5 + 77
    n = _mixed(n_len, 4)
    d = torch.from_numpy(n).cuda()
    for view in (d, torch.cat([torch.zeros(3, dtype=torch.uint8, device="cuda"), d])[3:]):
        w, acc = cn.n_to_bits_checked_dev(view, tail_lut=True)  # the SIMD encoders to the letter: what the oracle's bit-extract port computes
        assert np.array_equal(w.cpu().numpy().view(np.uint64), oracle.n_to_bits_bitextract(n)) and int(acc.item()) == oracle.validate(n)
        b, back, acc2 = cn.round_trip_checked_dev(view, tail_lut=True)
        assert torch.equal(b, w) and int(acc2.item()) == oracle.validate(n)
        assert torch.equal(cn.n_to_bits_checked_dev(view)[0], cn.n_to_bits_dev(view))
    n5 = _mixed(3456 * 64 * 2 + 3456 * 3 + 5, 6, five=True)
    w5, acc5 = cn.n_to_bits2_checked_dev(torch.from_numpy(n5).cuda())
    assert torch.equal(w5, cn.n_to_bits2_dev(torch.from_numpy(n5).cuda())) and int(acc5.item()) == oracle.validate(n5, allow_n=True)


def test_checked_calls_are_one_launch_and_graph_capturable(cn, oracle, torch_cuda):
    """enqueue-only like the unchecked entry points: no allocation, no synchronisation -- the counter is the caller's -- so a
    zeroing + checked encode + checked fused pass records into a HIP graph and replays on new contents"""
    torch = torch_cuda
    n_len = (1 << 18) + 40000 + 13
    d_in = torch.zeros(n_len + 3, dtype=torch.uint8, device="cuda")[3:]
    d_pk = torch.zeros((n_len + 31) // 32, dtype=torch.int64, device="cuda")
    d_pk2, d_back = torch.zeros_like(d_pk), torch.zeros(n_len, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(2, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cn.n_to_bits_checked_dev(d_in, out=d_pk, acc=acc[0:1])
        cn.round_trip_checked_dev(d_in, out_bits=d_pk2, out_n=d_back, acc=acc[1:2])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        acc.zero_()
        cn.n_to_bits_checked_dev(d_in, out=d_pk, acc=acc[0:1], tail_lut=True)
        cn.round_trip_checked_dev(d_in, out_bits=d_pk2, out_n=d_back, acc=acc[1:2], tail_lut=True)
    for seed in (1, 2, 3):
        n = _mixed(n_len, seed)
        d_in.copy_(torch.from_numpy(n))
        g.replay()
        torch.cuda.synchronize()
        assert acc.cpu().tolist() == [oracle.validate(n)] * 2
        assert np.array_equal(d_pk.cpu().numpy().view(np.uint64), oracle.n_to_bits_bitextract(n)) and torch.equal(d_pk, d_pk2)


@pytest.mark.parametrize("n_len", [1, 31, 40000, (1 << 20) - 5, (1 << 20) + 1, (1 << 22) + 77, (1 << 26) + 12345])
def test_host_tier_checked_encode(cn, oracle, n_len):
    """cnt_n_to_bits_checked / cnt_n_to_bits2_checked: the zero-copy small path (its own staged kernel, final word padded with
    'A'), the slot pipeline (one device counter per slot), every flag mode"""
    n = _mixed(n_len, n_len)
    for mode in MODES:
        w, bad = cn.n_to_bits_hip_checked(n, **mode)
        assert bad == oracle.validate(n) and np.array_equal(w, cn.n_to_bits_hip(n, **mode)), (n_len, mode)
    assert np.array_equal(cn.n_to_bits_hip_checked(n, strict_lut=True)[0], oracle.n_to_bits_lut(n))
    clean = VALID[np.random.default_rng(1).integers(0, 10, n_len)]
    assert cn.n_to_bits_hip_checked(clean)[1] == 0
    if n_len <= (1 << 22) + 77:
        n5 = _mixed(n_len, n_len + 1, five=True)
        w5, bad5 = cn.n_to_bits2_hip_checked(n5)
        assert bad5 == oracle.validate(n5, allow_n=True) and np.array_equal(w5, cn.n_to_bits2_hip(n5))
        assert np.array_equal(cn.n_to_bits2_hip_checked(n5, strict_lut=True)[0], oracle.n_to_bits2_lut(n5))


def test_checked_argument_errors(cn, torch_cuda):
    from cute_nucleotides_amd import _lib

    torch = torch_cuda
    L = _lib.lib()
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(2, dtype=torch.int64, device="cuda")
    c = torch.zeros(2, dtype=torch.int64, device="cuda")
    p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    assert L.cnt_n_to_bits_checked_dev(p(d), 64, p(o), 2, 0, None, None) == _lib.CNT_EINVAL  # the counter is never optional
    assert L.cnt_n_to_bits_checked_dev(p(d), 64, p(o), 2, 0, p(c, 4), None) == _lib.CNT_EINVAL  # ... and 8-byte aligned
    assert L.cnt_n_to_bits_checked_dev(p(d), 64, p(o), 1, 0, p(c), None) == _lib.CNT_ECAP
    assert L.cnt_n_to_bits_checked_dev(p(d), 64, p(o), 2, 0x2, p(c), None) == _lib.CNT_EINVAL  # CNT_ALLOW_N is cnt_validate's flag
    assert L.cnt_n_to_bits_dev(p(d), 64, p(o), 2, _lib.CNT_SPREAD_COUNT, None) == _lib.CNT_EINVAL  # ... and CNT_SPREAD_COUNT the checked calls'
    hb = ctypes.c_uint64(0)
    assert L.cnt_n_to_bits_checked(None, 0, None, 0, _lib.CNT_SPREAD_COUNT, ctypes.byref(hb)) == _lib.CNT_EINVAL  # the host tier keeps its own counters
    assert L.cnt_n_to_bits2_checked_dev(p(d), 54, p(o), 2, 0, None, None) == _lib.CNT_EINVAL
    assert L.cnt_round_trip_checked_dev(p(d), 64, p(o), 2, p(d), 0, p(c), None) == _lib.CNT_EINVAL  # back overlaps n
    assert L.cnt_n_to_bits_checked_dev(None, 0, None, 0, 0, p(c), None) == _lib.CNT_OK  # empty in, nothing added
    bad = ctypes.c_uint64(7)
    assert L.cnt_n_to_bits_checked(None, 0, None, 0, 0, ctypes.byref(bad)) == _lib.CNT_OK and bad.value == 0
    assert L.cnt_n_to_bits_checked(None, 0, None, 0, 0, None) == _lib.CNT_EINVAL
    torch.cuda.synchronize()
    assert c.cpu().tolist() == [0, 0]


def test_sharded_queue_checked_forms_on_an_explicit_device_list(cn, oracle, torch_cuda):
    """cnt_*_checked_sharded_dev_enqueue through a queue opened on devices [0, 0, 0] -- the PRODUCT library: an explicit device
    list may repeat a device, no test hook is involved -- with a ragged and an empty shard, three ops queued ahead, one wait"""
    from cute_nucleotides_amd import sharding

    torch = torch_cuda
    lens = [(1 << 18) + 4096 + 33, 0, 100003]
    ns = [_mixed(m, 40 + k) if m else np.zeros(0, dtype=np.uint8) for k, m in enumerate(lens)]
    shards = [torch.from_numpy(n).cuda() for n in ns]
    outs = [torch.empty((m + 31) // 32, dtype=torch.int64, device="cuda") for m in lens]
    outs2 = [torch.empty_like(o) for o in outs]
    backs = [torch.empty(m, dtype=torch.uint8, device="cuda") for m in lens]
    bad = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in lens]
    bad2 = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in lens]
    with sharding.DevQueue(devices=[0, 0, 0], timed=True) as q:
        assert q.ndev == 3 and q.devices == [0, 0, 0]
        q.n_to_bits(shards, outs, invalid=bad, tail_lut=True)
        q.round_trip(shards, outs2, backs, invalid=bad2, strict_lut=True)
        q.n_to_bits(shards, outs, invalid=bad, tail_lut=True)  # adds again
        ms = q.wait()
        assert ms[0] > 0 and ms[2] > 0
    for k, n in enumerate(ns):
        want = oracle.validate(n) if n.size else 0
        assert int(bad[k].item()) == 2 * want and int(bad2[k].item()) == want, k
        assert np.array_equal(outs[k].cpu().numpy().view(np.uint64), oracle.n_to_bits_bitextract(n) if n.size else np.zeros(0, dtype=np.uint64))
        assert np.array_equal(outs2[k].cpu().numpy().view(np.uint64), oracle.n_to_bits_lut(n) if n.size else np.zeros(0, dtype=np.uint64))
    n5 = [_mixed(3456 * 5 + 9, 50, five=True)]
    with sharding.DevQueue(devices=[0]) as q:
        o5, b5 = [torch.empty((n5[0].size + 26) // 27, dtype=torch.int64, device="cuda")], [torch.zeros(1, dtype=torch.int64, device="cuda")]
        q.n_to_bits([torch.from_numpy(n5[0]).cuda()], o5, five_letter=True, invalid=b5)
        q.wait()
    assert int(b5[0].item()) == oracle.validate(n5[0], allow_n=True)
    assert torch.equal(o5[0], cn.n_to_bits2_dev(torch.from_numpy(n5[0]).cuda()))


def test_metric_size_checked_encode_is_a_single_pass_at_the_plain_encoders_speed(cn, oracle, torch_cuda, fullsize):
    """2^34 nt of device-generated random ACGT (the metric's buffer): the checked encode finds nothing, writes the plain encoder's
    words (checksum of the packed stream), and -- the point of the exercise -- costs what the plain encode costs, where
    cnt_validate_dev + cnt_n_to_bits_dev is 2.25 B/nt.  Then K strays are planted at known places (first byte, last byte, tile
    borders, the middle of nowhere) and counted exactly, aligned and 5 bytes off the grid."""
    from cute_nucleotides_amd import devutil, packed_ops

    torch = torch_cuda
    need_free_hbm(26)
    n_len = 1 << 34
    buf = torch.empty(n_len + 128, dtype=torch.uint8, device="cuda")
    d = buf[:n_len]
    devutil.fill_random_acgt(d, 0x5EED)
    out = torch.empty(n_len // 32 + 1, dtype=torch.int64, device="cuda")
    acc = torch.zeros(1, dtype=torch.int64, device="cuda")

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    plain_ms = timed(lambda: cn.n_to_bits_dev(d, out=out))
    want_sum = devutil.checksum_words(out[: n_len // 32])
    checked_ms = timed(lambda: cn.n_to_bits_checked_dev(d, out=out, acc=acc))
    assert int(acc.item()) == 0 and devutil.checksum_words(out[: n_len // 32]) == want_sum
    two_pass_ms = timed(lambda: (packed_ops.validate_dev(d, acc=acc), cn.n_to_bits_dev(d, out=out)))
    fullsize(34, checked_ms, plain_ms=round(plain_ms, 3), two_pass_ms=round(two_pass_ms, 3), checked_over_plain=round(checked_ms / plain_ms, 4),
             frac_of_8TBps=round(1.25 * n_len / (checked_ms * 1e-3) / 8e12, 4))
    assert checked_ms < 1.05 * plain_ms, (checked_ms, plain_ms)  # the bench line holds it to 1.5 %; this is the tripwire
    assert checked_ms < 0.75 * two_pass_ms, (checked_ms, two_pass_ms)
    spots = [0, 1, 2047, 2048, 4095, 4096, (1 << 20) + 17, (1 << 33) - 1, 1 << 33, n_len - 2049, n_len - 33, n_len - 1] + [int(x) for x in np.random.default_rng(3).integers(0, n_len, 200)]
    spots = sorted(set(spots))
    d[torch.tensor(spots, device="cuda")] = ord("N")
    acc.zero_()
    cn.n_to_bits_checked_dev(d, out=out, acc=acc)
    assert int(acc.item()) == len(spots)
    off = buf[5 : 5 + n_len]  # the window kernel at the metric size: the buffer entered five bytes further on
    buf[n_len : n_len + 5] = ord("x")  # ... so the last five bytes are strays too, and the first five spots' neighbours change hands
    want_off = sum(1 for s in spots if s >= 5) + 5
    acc.zero_()
    off_ms = timed(lambda: cn.n_to_bits_checked_dev(off, out=out, acc=acc), reps=1)
    assert int(acc.item()) == 2 * want_off  # timed() ran it twice
    fullsize(34, off_ms, input_offset=5)
    b, back, acc3 = cn.round_trip_checked_dev(d[: 1 << 32])
    assert int(acc3.item()) == sum(1 for s in spots if s < (1 << 32))
