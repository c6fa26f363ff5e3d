"""The `_into` forms of the host mirrors: n_to_bits_hip_into / bits_to_n_hip_into and the 5-letter
pair write into an array the CALLER owns -- same words, same letters, same errors as the returning forms, the buffer is
written in place and nothing past the result is touched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cn():
    import cute_nucleotides_amd as cn

    return cn


@pytest.mark.parametrize("n_len", [0, 1, 31, 32, 40000, (1 << 20) + 13, (1 << 24) + 5])
def test_into_forms_match_the_oracle_and_write_in_place(cn, oracle, n_len):
    n = oracle.fill_random_acgt(n_len, 77) if n_len else np.empty(0, dtype=np.uint8)
    words = (n_len + 31) // 32
    buf = np.full(words + 4, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
    got = cn.n_to_bits_hip_into(n, buf)
    assert got.base is buf or got is buf or got.size == 0
    assert np.array_equal(got, oracle.n_to_bits_lut(n)) and (buf[words:] == 0xA5A5A5A5A5A5A5A5).all()
    back = np.full(n_len + 64, 0x2A, dtype=np.uint8)
    out = cn.bits_to_n_hip_into(got, n_len, back)
    assert np.array_equal(out, n) and (back[n_len:] == 0x2A).all()
    # 5-letter pair
    n5 = oracle.fill_random_acgtn(n_len, 78) if n_len else np.empty(0, dtype=np.uint8)
    w5 = (n_len + 26) // 27
    buf5 = np.full(w5 + 4, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
    got5 = cn.n_to_bits2_hip_into(n5, buf5)
    assert np.array_equal(got5, oracle.n_to_bits2_lut(n5)) and (buf5[w5:] == 0xA5A5A5A5A5A5A5A5).all()
    back5 = np.full(n_len + 64, 0x2A, dtype=np.uint8)
    assert np.array_equal(cn.bits_to_n2_hip_into(got5, n_len, back5), n5) and (back5[n_len:] == 0x2A).all()


def test_into_forms_refuse_unusable_outputs(cn):
    n = np.frombuffer(b"ATCG" * 16, dtype=np.uint8)
    with pytest.raises(ValueError):
        cn.n_to_bits_hip_into(n, np.empty(1, dtype=np.uint64))  # two words needed
    with pytest.raises(ValueError):
        cn.n_to_bits_hip_into(n, np.empty(8, dtype=np.uint8))  # wrong element type
    with pytest.raises(ValueError):
        cn.n_to_bits_hip_into(n, np.empty(8, dtype=np.uint64)[::2])  # not contiguous
    ro = np.empty(8, dtype=np.uint64)
    ro.flags.writeable = False
    with pytest.raises(ValueError):
        cn.n_to_bits_hip_into(n, ro)
    bits = cn.n_to_bits_hip(n)
    with pytest.raises(ValueError, match="The length is greater than the number of nucleotides!"):
        cn.bits_to_n_hip_into(bits, 65, np.empty(128, dtype=np.uint8))  # the reference's panic, n_to_bits.rs:52-54
    with pytest.raises(ValueError):
        cn.bits_to_n_hip_into(bits, 64, np.empty(63, dtype=np.uint8))
    assert cn.n_to_bits_hip_into(b"ATCG" * 8, np.zeros(1, dtype=np.uint64)).tolist() == [0xD8D8D8D8D8D8D8D8]  # n_to_bits.rs:414-415
