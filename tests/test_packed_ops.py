"""Packed-domain operations: Hamming distance, complement, reverse complement,
validation.  These are NOT in the reference (parity unpinned): the CPU part pins the oracle's
scalar definitions against plain-Python/ASCII-level definitions, the GPU part compares the
HIP kernels with the oracle bit for bit and checks algebraic properties at large sizes."""
import numpy as np
import pytest

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _ascii(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]


@pytest.mark.parametrize("n_len", [0, 1, 5, 31, 32, 33, 64, 100, 1000, 4099])
def test_oracle_definitions_against_ascii_level_semantics(oracle, n_len):
    rng = np.random.default_rng(n_len)
    a, b = _ascii(rng, n_len), _ascii(rng, n_len)
    pa, pb = oracle.n_to_bits_lut(a), oracle.n_to_bits_lut(b)
    assert oracle.hamming(pa, pb, n_len) == int((a != b).sum())
    comp = oracle.complement(pa, n_len)
    assert bytes(oracle.bits_to_n_lut(comp, n_len)) == bytes(a).translate(COMP)
    rc = oracle.reverse_complement(pa, n_len)
    assert bytes(oracle.bits_to_n_lut(rc, n_len)) == bytes(a).translate(COMP)[::-1]
    # unused high bits stay zero, like every encoder's output
    if n_len & 31:
        assert int(comp[-1]) >> (2 * (n_len & 31)) == 0 and int(rc[-1]) >> (2 * (n_len & 31)) == 0
    # garbage beyond len in the inputs is ignored
    if n_len & 31:
        pa2 = pa.copy()
        pa2[-1] |= np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(2 * (n_len & 31))
        assert oracle.hamming(pa2, pb, n_len) == oracle.hamming(pa, pb, n_len)
        assert np.array_equal(oracle.complement(pa2, n_len), comp)
        assert np.array_equal(oracle.reverse_complement(pa2, n_len), rc)


def test_oracle_validate():
    from oracle import cnt_oracle as orc

    allb = np.arange(256, dtype=np.uint8)
    assert orc.validate(allb) == 256 - 10
    assert orc.validate(allb, allow_n=True) == 256 - 12
    assert orc.validate(b"ACGTUacgtu" * 100) == 0
    assert orc.validate(b"ACGTN") == 1 and orc.validate(b"ACGTN", allow_n=True) == 0
    assert orc.validate(b"") == 0


# ---------------------------------------------------------------------------- GPU part
gpu = pytest.mark.gpu

SIZES = [0, 1, 5, 31, 32, 33, 63, 64, 65, 1000, 4099, 32 * 8192, 32 * 8192 + 7, (1 << 22) + 13, 3 * (1 << 21) + 64]


@pytest.fixture(params=[1, 0], ids=["persistent", "tiles+second-pass"])
def reduce_form(request):
    """both forms of the reductions: one launch of persistent waves (shipped: the product build, no knob) and round 1's
    tiles + stream-ordered scratch + second pass (lab build only, tuning key reduce_persistent)"""
    if request.param == 1:
        yield 1
        return
    request.getfixturevalue("lab_build")
    from cute_nucleotides_amd import devutil

    devutil.set_tuning("reduce_persistent", 0)
    yield 0
    devutil.set_tuning("reduce_persistent", 1)


@gpu
@pytest.mark.parametrize("n_len", SIZES)
def test_gpu_ops_match_oracle(oracle, n_len, reduce_form):
    import torch

    from cute_nucleotides_amd import packed_ops as po

    rng = np.random.default_rng(n_len + 3)
    words = (n_len + 31) // 32
    a = rng.integers(0, 2**64, words, dtype=np.uint64)  # includes garbage beyond len in the last word
    b = a.copy()
    flip = rng.integers(0, 2**64, words, dtype=np.uint64) & rng.integers(0, 2**64, words, dtype=np.uint64)
    b ^= flip
    assert po.hamming_hip(a, b, n_len) == oracle.hamming(a, b, n_len)
    assert np.array_equal(po.complement_hip(a, n_len), oracle.complement(a, n_len))
    assert np.array_equal(po.reverse_complement_hip(a, n_len), oracle.reverse_complement(a, n_len))
    if words:
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        assert int(po.hamming_dev(da, db, n_len).item()) == oracle.hamming(a, b, n_len)
        assert np.array_equal(po.complement_dev(da, n_len).cpu().numpy().view(np.uint64), oracle.complement(a, n_len))
        assert np.array_equal(po.reverse_complement_dev(da, n_len).cpu().numpy().view(np.uint64), oracle.reverse_complement(a, n_len))
        # unaligned (8-B but not 16-B aligned) device pointers take the generic path
        if words > 3:
            m = (words - 1) * 32 if n_len == words * 32 else n_len - 32
            assert int(po.hamming_dev(da[1:], db[1:], m).item()) == oracle.hamming(a[1:], b[1:], m)
            assert np.array_equal(po.complement_dev(da[1:], m).cpu().numpy().view(np.uint64), oracle.complement(a[1:], m))


@gpu
def test_gpu_validate(oracle, reduce_form):
    import torch

    from cute_nucleotides_amd import packed_ops as po

    rng = np.random.default_rng(1)
    for n_len in (0, 1, 15, 16, 17, 65536, 65536 * 3 + 5, (1 << 22) + 3):
        n = rng.integers(0, 256, n_len, dtype=np.uint8)
        for allow in (False, True):
            assert po.validate_hip(n, allow_n=allow) == oracle.validate(n, allow_n=allow)
            if n_len:
                d = torch.from_numpy(n).cuda()
                assert int(po.validate_dev(d, allow_n=allow).item()) == oracle.validate(n, allow_n=allow)
                if n_len > 1:
                    assert int(po.validate_dev(d[1:], allow_n=allow).item()) == oracle.validate(n[1:], allow_n=allow)
        good = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)[rng.integers(0, 10, n_len)]
        assert po.validate_hip(good) == 0


@gpu
def test_gpu_ops_pointer_phases(oracle, reduce_form):
    """Word pointers at every 8-B phase of a 128-B line (equal and unequal phases of the two
    streams), ASCII pointers at odd bytes: heads are peeled, results and guard words unchanged."""
    import torch

    from cute_nucleotides_amd import packed_ops as po

    rng = np.random.default_rng(9)
    words = 3 * 8192 + 40
    a = rng.integers(0, 2**64, words + 40, dtype=np.uint64)
    b = a ^ (rng.integers(0, 2**64, words + 40, dtype=np.uint64) & rng.integers(0, 2**64, words + 40, dtype=np.uint64))
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    obuf = torch.empty(words + 64, dtype=torch.int64, device="cuda")
    for pa in (0, 1, 2, 3, 7, 8, 15, 16, 17):
        for pb in (pa, pa + 1, pa + 2, 0):
            for n_len in (words * 32, words * 32 - 45, 64 * 32 + 3):
                w = (n_len + 31) // 32
                got = int(po.hamming_dev(da[pa : pa + w], db[pb : pb + w], n_len).item())
                assert got == oracle.hamming(a[pa : pa + w], b[pb : pb + w], n_len), (pa, pb, n_len)
                obuf.fill_(-1)
                po.complement_dev(da[pa : pa + w], n_len, out=obuf[8 + pb : 8 + pb + w])
                o = obuf.cpu().numpy()
                assert (o[: 8 + pb] == -1).all() and (o[8 + pb + w :] == -1).all()
                assert np.array_equal(o[8 + pb : 8 + pb + w].view(np.uint64), oracle.complement(a[pa : pa + w], n_len)), (pa, pb, n_len)
    # reverse complement: output phase x every window phase (len % 32) x tile-count boundaries
    for po_ in (0, 1, 2, 5, 15, 16):
        for pi_ in (0, 1):
            for n_len in [512 * 32 * 2 + k for k in (0, 1, 5, 31, 32, 33)] + [512 * 32 + 16 * 32 - 1, 512 * 32 + 15 * 32 + 1, words * 32 - 7, 40 * 32]:
                w = (n_len + 31) // 32
                obuf.fill_(-1)
                po.reverse_complement_dev(da[pi_ : pi_ + w], n_len, out=obuf[8 + po_ : 8 + po_ + w])
                o = obuf.cpu().numpy()
                assert (o[: 8 + po_] == -1).all() and (o[8 + po_ + w :] == -1).all()
                assert np.array_equal(o[8 + po_ : 8 + po_ + w].view(np.uint64), oracle.reverse_complement(a[pi_ : pi_ + w], n_len)), (po_, pi_, n_len)
    n = rng.integers(0, 256, 5 * 65536 + 300, dtype=np.uint8)
    d = torch.from_numpy(n).cuda()
    for off in (0, 1, 7, 16, 100, 127, 128, 129):
        for n_len in (n.size - off, 65536 + 127, 65536 + 128, 200):
            for allow in (False, True):
                assert int(po.validate_dev(d[off : off + n_len], allow_n=allow).item()) == oracle.validate(n[off : off + n_len], allow_n=allow)


@gpu
def test_gpu_large_properties(oracle):
    """2^32 nt: rc(rc(x)) == x, complement is an involution, hamming(x, complement(x)) == len,
    hamming(x, x) == 0, and encode(valid ASCII) validates clean."""
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil, packed_ops as po

    n_len = (1 << 32) + 77
    d = torch.empty(n_len, dtype=torch.uint8, device="cuda")
    devutil.fill_random_acgt(d, 5)
    assert int(po.validate_dev(d).item()) == 0
    d[12345] = ord("N")
    d[n_len - 1] = 0
    assert int(po.validate_dev(d).item()) == 2 and int(po.validate_dev(d, allow_n=True).item()) == 1
    devutil.fill_random_acgt(d, 5)
    x = cn.n_to_bits_dev(d)
    c = po.complement_dev(x, n_len)
    assert int(po.hamming_dev(x, x, n_len).item()) == 0
    assert int(po.hamming_dev(x, c, n_len).item()) == n_len
    assert devutil.count_mismatch(po.complement_dev(c, n_len), x) == 0
    rc = po.reverse_complement_dev(x, n_len)
    assert devutil.count_mismatch(po.reverse_complement_dev(rc, n_len), x) == 0
    # the head of rc is the complement of the reversed tail of the ASCII
    tail = d[n_len - 1000 :].cpu().numpy()
    head = cn.bits_to_n_dev(rc, 1000).cpu().numpy()
    assert bytes(head) == bytes(tail).translate(COMP)[::-1]
    # the reductions above took their scratch from the stream-ordered allocator; none fell back to the slow path
    assert devutil.get_tuning("reduce_fallbacks") == 0


@gpu
def test_reductions_are_graph_capturable_and_accumulate(oracle):
    """Since round 2 hamming / validate neither synchronise nor allocate (one launch of persistent waves): a whole
    encode -> hamming -> validate chain records into a HIP graph and replays on new buffer contents; the counters are
    the caller's and the calls ADD to them (two calls into one counter = the sum)."""
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import packed_ops as po

    n_len = (1 << 21) + 4099  # tiles' worth of persistent pieces plus a ragged tail and a word tail
    rng = np.random.default_rng(9)
    d_a = torch.zeros(n_len, dtype=torch.uint8, device="cuda")
    d_b = torch.zeros(n_len, dtype=torch.uint8, device="cuda")
    p_a = torch.zeros((n_len + 31) // 32, dtype=torch.int64, device="cuda")
    p_b = torch.zeros_like(p_a)
    ham = torch.zeros(1, dtype=torch.int64, device="cuda")
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")

    def chain():
        ham.zero_()
        bad.zero_()
        cn.n_to_bits_dev(d_a, out=p_a)
        cn.n_to_bits_dev(d_b, out=p_b)
        po.hamming_dev(p_a, p_b, n_len, acc=ham)
        po.validate_dev(d_a, acc=bad)
        po.validate_dev(d_b, acc=bad)  # accumulates into the same counter

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        chain()  # warm-up outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    for seed in (1, 2, 3):
        a = _ascii(rng, n_len)
        b = _ascii(rng, n_len)
        b[rng.integers(0, n_len, 50)] = ord("N")
        a[7] = 0
        d_a.copy_(torch.from_numpy(a))
        d_b.copy_(torch.from_numpy(b))
        g.replay()
        torch.cuda.synchronize()
        # expected values from the oracle's definitions on what the encoder makes of the bytes
        pa, pb = p_a.cpu().numpy().view(np.uint64), p_b.cpu().numpy().view(np.uint64)
        assert int(ham.item()) == oracle.hamming(pa, pb, n_len), seed
        assert int(bad.item()) == oracle.validate(a) + oracle.validate(b), seed
        assert oracle.validate(a) >= 1 and oracle.validate(b) >= 1
    with pytest.raises(ValueError):
        po.hamming_dev(p_a, p_b, n_len, acc=torch.zeros(1, dtype=torch.int32, device="cuda"))
