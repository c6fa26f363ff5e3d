"""The several-launch loops of every tiled launcher, at sizes of a few MiB.

A kernel launch takes at most 2^31-1 threads, so the launchers cut very large buffers into several launches
(hip/codec2_launch.hpp: max_tiles_per_launch) -- with the default shapes that starts at 2^36 nucleotides for the 2-bit
encoder and never for most others, so the loops, and the rule that a call's edge work (head words, ragged end) rides in the
LAST launch only, were exercised by two 64-GiB tests at best.  The tuning key "launch_tiles" lowers the limit: here every
tier runs with 64 and 128 tiles per launch against the oracle, at aligned and misaligned pointers, and a captured graph
confirms that the calls really were cut."""
import numpy as np
import pytest

from test_gpu_codec2 import _kernel_nodes_of, _rand_valid

pytestmark = pytest.mark.gpu

ALPHA5 = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)


@pytest.fixture(params=[64, 128])
def launch_tiles(request, lab_build):
    from cute_nucleotides_amd import devutil

    saved = {k: devutil.get_tuning(k) for k in ("encode", "decode", "encode2", "decode2", "small_nt", "reduce_persistent")}
    devutil.set_tuning("small_nt", 0)
    devutil.set_tuning("launch_tiles", request.param)
    assert devutil.get_tuning("launch_tiles") == request.param
    yield request.param
    devutil.set_tuning("launch_tiles", 0)
    for k, v in saved.items():
        devutil.set_tuning(k, v)


def test_launch_tiles_key_is_validated(lab_build):
    from cute_nucleotides_amd import devutil

    for bad in (-64, 1, 63, 100):
        with pytest.raises(Exception):
            devutil.set_tuning("launch_tiles", bad)
    assert devutil.get_tuning("launch_tiles") == 0


@pytest.mark.parametrize("enc_variant,dec_variant", [(0, 0), (1, 35), (17, 1), (4, 5)])
def test_2bit_codec_in_several_launches(oracle, launch_tiles, enc_variant, dec_variant):
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil

    devutil.set_tuning("encode", enc_variant)
    devutil.set_tuning("decode", dec_variant)
    ibuf = torch.zeros((1 << 21) + 512, dtype=torch.uint8, device="cuda")
    pbuf = torch.zeros((1 << 16) + 64, dtype=torch.int64, device="cuda")
    obuf = torch.zeros((1 << 21) + 8192, dtype=torch.uint8, device="cuda")
    for n_len in (launch_tiles * 4096 * 3, launch_tiles * 4096 * 3 + 4096 * 17 + 1234, launch_tiles * 4096 + 5, launch_tiles * 2048 * 2 - 1):
        host = _rand_valid(n_len, n_len)
        want = oracle.n_to_bits_lut(host)
        back = oracle.bits_to_n_lut(want, n_len)
        words = (n_len + 31) // 32
        for io, po, oo in ((0, 0, 0), (5, 3, 77), (127, 1, 4095)):
            view = ibuf[io : io + n_len]
            view.copy_(torch.from_numpy(host))
            packed, out = pbuf[po : po + words], obuf[oo : oo + n_len]
            for strict, tail in ((False, False), (True, False), (False, True)):
                pbuf.fill_(-1)
                cn.n_to_bits_dev(view, out=packed, strict_lut=strict, tail_lut=tail)
                got = pbuf.cpu().numpy()
                assert np.array_equal(got[po : po + words].view(np.uint64), want), (n_len, io, po, strict, tail)
                assert (got[:po] == -1).all() and (got[po + words :] == -1).all()
            obuf.fill_(0x2A)
            cn.bits_to_n_dev(packed, n_len, out=out)
            got = obuf.cpu().numpy()
            assert np.array_equal(got[oo : oo + n_len], back), (n_len, po, oo)
            assert (got[:oo] == 0x2A).all() and (got[oo + n_len :] == 0x2A).all()
    # the override took effect: a call over 3 x launch_tiles (+ a few) tiles is four kernel nodes in a captured graph
    n_len = launch_tiles * 4096 * 3 + 4096 * 17 + 1234
    view, packed, out = ibuf[:n_len], pbuf[: (n_len + 31) // 32], obuf[:n_len]
    if dec_variant == 0:
        assert _kernel_nodes_of(torch, lambda: cn.bits_to_n_dev(packed, n_len, out=out)) == 4
    if enc_variant == 0:
        assert _kernel_nodes_of(torch, lambda: cn.n_to_bits_dev(view, out=packed)) == 7  # 2-KiB tiles: twice as many


@pytest.mark.parametrize("strict", [False, True])
def test_fused_round_trip_in_several_launches(oracle, launch_tiles, strict):
    import torch

    import cute_nucleotides_amd as cn

    for n_len in (launch_tiles * 4096 * 2, launch_tiles * 4096 * 2 + 4096 * 9 + 4077, launch_tiles * 2048 + 31):
        host = _rand_valid(n_len, 7 + n_len) if not strict else np.random.default_rng(n_len).integers(0, 256, n_len, dtype=np.uint8)
        want = oracle.n_to_bits_lut(host)
        back = oracle.bits_to_n_lut(want, n_len)
        d = torch.from_numpy(host).cuda()
        bits, out = cn.round_trip_dev(d, strict_lut=strict)
        assert np.array_equal(bits.cpu().numpy().view(np.uint64), want), n_len
        assert np.array_equal(out.cpu().numpy(), back), n_len


@pytest.mark.parametrize("enc_variant,dec_variant", [(0, 0), (1, 1), (2, 2), (3, 3)])
def test_5letter_codec_in_several_launches(oracle, launch_tiles, enc_variant, dec_variant):
    import torch

    import cute_nucleotides_amd as cn
    from cute_nucleotides_amd import devutil

    devutil.set_tuning("encode2", enc_variant)
    devutil.set_tuning("decode2", dec_variant)
    tile = 27 * 128  # one wave's tile at two words per lane (the four-word shapes take two of them)
    ibuf = torch.zeros((1 << 21) + 512, dtype=torch.uint8, device="cuda")
    pbuf = torch.zeros((1 << 17) + 64, dtype=torch.int64, device="cuda")
    obuf = torch.zeros((1 << 21) + 8192, dtype=torch.uint8, device="cuda")
    for n_len in (launch_tiles * tile * 3, launch_tiles * tile * 3 + tile * 5 + 1000, launch_tiles * tile + 1):
        host = ALPHA5[np.random.default_rng(n_len).integers(0, ALPHA5.size, n_len)]
        want = oracle.n_to_bits2_lut(host)
        back = oracle.bits_to_n2_lut(want, n_len)
        words = (n_len + 26) // 27
        for io, po, oo in ((0, 0, 0), (5, 3, 77), (127, 1, 4095)):
            view = ibuf[io : io + n_len]
            view.copy_(torch.from_numpy(host))
            packed, out = pbuf[po : po + words], obuf[oo : oo + n_len]
            for strict, tail in ((False, False), (True, False), (False, True)):
                pbuf.fill_(-1)
                cn.n_to_bits2_dev(view, out=packed, strict_lut=strict, tail_lut=tail)
                got = pbuf.cpu().numpy()
                assert np.array_equal(got[po : po + words].view(np.uint64), want), (n_len, io, po, strict, tail)
                assert (got[:po] == -1).all() and (got[po + words :] == -1).all()
            obuf.fill_(0x2A)
            cn.bits_to_n2_dev(packed, n_len, out=out)
            got = obuf.cpu().numpy()
            assert np.array_equal(got[oo : oo + n_len], back), (n_len, po, oo)
            assert (got[:oo] == 0x2A).all() and (got[oo + n_len :] == 0x2A).all()


@pytest.mark.parametrize("persistent", [1, 0])
def test_packed_ops_in_several_launches(oracle, launch_tiles, persistent):
    import torch

    from cute_nucleotides_amd import devutil, packed_ops as po

    devutil.set_tuning("reduce_persistent", persistent)
    rng = np.random.default_rng(launch_tiles + persistent)
    words = launch_tiles * 512 * 2 + 512 * 5 + 77  # 4-KiB tiles of 512 words
    a = rng.integers(0, 2**64, words + 16, dtype=np.uint64)
    b = a ^ (rng.integers(0, 2**64, words + 16, dtype=np.uint64) & rng.integers(0, 2**64, words + 16, dtype=np.uint64))
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    obuf = torch.empty(words + 64, dtype=torch.int64, device="cuda")
    for pa, pb in ((0, 0), (1, 3), (2, 2)):
        for n_len in (words * 32, words * 32 - 45, launch_tiles * 512 * 32):
            w = (n_len + 31) // 32
            assert int(po.hamming_dev(da[pa : pa + w], db[pb : pb + w], n_len).item()) == oracle.hamming(a[pa : pa + w], b[pb : pb + w], n_len), (pa, pb, n_len)
            for fn, ref in ((po.complement_dev, oracle.complement), (po.reverse_complement_dev, oracle.reverse_complement)):
                obuf.fill_(-1)
                fn(da[pa : pa + w], n_len, out=obuf[8 + pb : 8 + pb + w])
                o = obuf.cpu().numpy()
                assert (o[: 8 + pb] == -1).all() and (o[8 + pb + w :] == -1).all()
                assert np.array_equal(o[8 + pb : 8 + pb + w].view(np.uint64), ref(a[pa : pa + w], n_len)), (fn.__name__, pa, pb, n_len)
    n = rng.integers(0, 256, launch_tiles * 16384 * 2 + 70000, dtype=np.uint8)
    d = torch.from_numpy(n).cuda()
    for off in (0, 1, 127):
        for allow in (False, True):
            assert int(po.validate_dev(d[off:], allow_n=allow).item()) == oracle.validate(n[off:], allow_n=allow)
