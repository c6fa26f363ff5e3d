"""Randomised differential test of the HIP path against the oracle: random lengths (biased to
tile edges), random 16-B/8-B/1-B pointer offsets, both codecs, both encode modes, device tier.
Deterministic seeds; a few hundred cases in a few seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGES = [0, 1, 2, 3, 26, 27, 31, 32, 33, 63, 64, 2047, 2048, 2049, 3455, 3456, 3457, 4095, 4096, 4097,
         32767, 32768, 32769, 65535, 65536]


def _lengths(rng, k):
    out = []
    for _ in range(k):
        base = int(rng.choice(EDGES)) * int(rng.integers(1, 40))
        out.append(max(0, base + int(rng.integers(-3, 4))))
    return out


@pytest.fixture(params=[0, 1 << 17], ids=["tiles", "small-path"])
def small_nt(request):
    """with and without the single-launch path for small ragged inputs (tuning key small_nt)"""
    from cute_nucleotides_amd import devutil

    saved = devutil.get_tuning("small_nt")
    devutil.set_tuning("small_nt", request.param)
    yield request.param
    devutil.set_tuning("small_nt", saved)


@pytest.mark.parametrize("seed", range(4))
def test_two_bit_codec_fuzz(oracle, small_nt, seed):
    import torch

    import cute_nucleotides_amd as cn

    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTUacgtu", dtype=np.uint8)
    for n_len in _lengths(rng, 40):
        off_in, off_out = int(rng.choice([0, 0, 16, 48, 8, 1, 5])), int(rng.choice([0, 0, 1, 2]))  # out offset in words
        anybytes = bool(rng.integers(0, 3) == 0)
        n = rng.integers(0, 256, n_len, dtype=np.uint8) if anybytes else alpha[rng.integers(0, 10, n_len)]
        buf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
        view = buf[off_in : off_in + n_len]
        view.copy_(torch.from_numpy(n))
        words = (n_len + 31) // 32
        obuf = torch.full((words + 4,), -1, dtype=torch.int64, device="cuda")
        strict = bool(rng.integers(0, 2))
        got = cn.n_to_bits_dev(view, out=obuf[off_out:], strict_lut=strict).cpu().numpy().view(np.uint64)
        if strict or not anybytes:
            want = oracle.n_to_bits_lut(n)
        else:  # default mode on arbitrary bytes: (byte>>1)&3 everywhere, tail included
            codes = ((n >> 1) & 3).astype(np.uint64)
            pad = np.zeros((-n_len) % 32, dtype=np.uint64)
            want = np.bitwise_or.reduce(np.concatenate([codes, pad]).reshape(-1, 32) << (np.arange(32, dtype=np.uint64) * np.uint64(2)), axis=1) if words else np.empty(0, np.uint64)
        assert np.array_equal(got, want), (seed, n_len, off_in, off_out, strict, anybytes)
        o = obuf.cpu().numpy()
        assert (o[:off_out] == -1).all() and (o[off_out + words :] == -1).all()
        # decode a random prefix of what was just packed, into an offset output
        length = int(rng.integers(0, n_len + 1))
        dout = torch.full((n_len + 80,), 0x5A, dtype=torch.uint8, device="cuda")
        off_d = int(rng.choice([0, 0, 16, 7]))
        cn.bits_to_n_dev(obuf[off_out : off_out + words], length, out=dout[off_d:])
        d = dout.cpu().numpy()
        assert np.array_equal(d[off_d : off_d + length], oracle.bits_to_n_lut(want, length)), (seed, n_len, length)
        assert (d[:off_d] == 0x5A).all() and (d[off_d + length :] == 0x5A).all()


@pytest.mark.parametrize("seed", range(3))
def test_five_letter_codec_fuzz(oracle, small_nt, seed):
    import torch

    import cute_nucleotides_amd as cn

    rng = np.random.default_rng(100 + seed)
    alpha = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)
    for n_len in _lengths(rng, 30):
        off_in = int(rng.choice([0, 0, 16, 27, 1]))
        n = alpha[rng.integers(0, 12, n_len)]
        buf = torch.zeros(n_len + 64, dtype=torch.uint8, device="cuda")
        view = buf[off_in : off_in + n_len]
        view.copy_(torch.from_numpy(n))
        want = oracle.n_to_bits2_lut(n)
        got = cn.n_to_bits2_dev(view, strict_lut=bool(rng.integers(0, 2))).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), (seed, n_len, off_in)
        length = int(rng.integers(0, n_len + 1))
        dout = torch.full((n_len + 80,), 0x5A, dtype=torch.uint8, device="cuda")
        off_d = int(rng.choice([0, 0, 16, 3]))
        cn.bits_to_n2_dev(torch.from_numpy(want.view(np.int64)).cuda(), length, out=dout[off_d:])
        d = dout.cpu().numpy()
        assert np.array_equal(d[off_d : off_d + length], oracle.bits_to_n2_lut(want, length)), (seed, n_len, length)
        assert (d[:off_d] == 0x5A).all() and (d[off_d + length :] == 0x5A).all()
