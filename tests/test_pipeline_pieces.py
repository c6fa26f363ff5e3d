"""How the host tier's pipeline cuts a call into pieces (hip/host_tier.inc Pieces / pipeline_chunk, reached through the test hook
cnt_test_pipeline_pieces -- no device needed): whatever the size and whichever schedule,

  * the pieces add up to the call, none is empty, none exceeds the slot (the full chunk),
  * every piece but the last is a whole number of 8192-word granules -- so every piece starts on a packed word and the
    staged kernels' 16-byte loads stay aligned,
  * equal pieces (ramp off) are what pipeline_chunk always gave,
  * a ramped call opens with a quarter and a half piece and closes with a half and a quarter, never adds more than two
    pieces + the middle's rounding to the equal schedule, and calls below the ramp's threshold are untouched."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def L():
    from cute_nucleotides_amd import _lib, build

    build.build_hooks()
    prev = _lib.use_build("hooks")
    lib = _lib.lib()
    _lib.use_build(prev)
    return lib


def pieces(L, total, unit, ramp_log2):
    out = (ctypes.c_uint64 * 70000)()
    n = L.cnt_test_pipeline_pieces(total, unit, ramp_log2, out, len(out))
    assert 0 <= n <= len(out)
    return [int(out[i]) for i in range(n)]


SIZES = [1, 31, 32, 33, (1 << 20) + 1, 1 << 21, (1 << 22) - 5, 1 << 22, 3 << 21, 1 << 24, (1 << 24) + 27, 5 << 22, 1 << 25, (1 << 25) - 1, (1 << 25) + 8192 * 32,
         3 << 24, 1 << 26, (1 << 26) + 1, (1 << 26) - 1, 7 << 24, 1 << 27, (1 << 28) + 12345, 1 << 30, (1 << 30) + (1 << 23) + 3, (1 << 33) + 77]


@pytest.mark.parametrize("unit", [32, 27])
@pytest.mark.parametrize("ramp_log2", [0, 1, 22, 25, 28])
def test_pieces_cover_the_call_on_the_word_grid(L, unit, ramp_log2):
    gran = unit * 8192
    chunk = (8 << 20) if unit == 32 else 27 * 256 * 1024
    for total in SIZES:
        p = pieces(L, total, unit, ramp_log2)
        assert sum(p) == total, (total, p[:8])
        assert all(m > 0 for m in p) and max(p) <= chunk, (total, p[:8])
        assert all(m % gran == 0 for m in p[:-1]), (total, p[:8])
        equal = pieces(L, total, unit, 0)
        if ramp_log2 == 0:
            assert len(set(p[:-1])) <= 1 and p[-1] <= p[0], (total, p[:8])  # equal pieces, a shorter last one
            if total >= 4 * chunk:
                assert p[0] == chunk
            continue
        ramped = total >= (1 << ramp_log2) and len(equal) >= 3 and equal[0] >= 4 * gran
        if not ramped:
            assert p == equal, (total, p[:8])
            continue
        c = equal[0]
        q, h = c // 4 // gran * gran, c // 2 // gran * gran
        assert p[0] == q and p[1] == h, (total, p[:8])
        assert p[-1] <= q and p[-2] == h and p[-1] > q - gran, (total, p[-4:])
        assert len(p) <= len(equal) + 3, (total, len(p), len(equal))  # 1.5 chunks' worth became four pieces
        mid = p[2:-2]
        assert mid and max(mid) - min(mid) <= gran and 2 * min(mid) >= c, (total, mid[:6])  # near-equal middle pieces, none small


def test_bad_arguments(L):
    out = (ctypes.c_uint64 * 4)()
    assert L.cnt_test_pipeline_pieces(100, 16, 0, out, 4) == -1
    assert L.cnt_test_pipeline_pieces(100, 32, 0, None, 4) == -1
    assert L.cnt_test_pipeline_pieces(1 << 30, 32, 0, out, 4) == 128  # counts past cap, writes only cap
