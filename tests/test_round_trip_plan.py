"""The launch plan of the any-alignment fused round trip (hip/device_tier.inc round_trip_plan, reached through the test
hook cnt_test_round_trip_plan -- no device needed), walked over EVERY combination of the three pointers' phases on the CPU
box: 128 input byte phases x 8 packed-word phases x 128 (4096 for large buffers: sampled) output byte phases, at lengths
around every boundary the plan has.  Checked for each: the aligned windows of all tiles lie inside the caller's buffer
(nothing is read in front of d_n or behind d_n + n_len), both funnel reads stay inside the wave's 320-dword slab and
inside the vectors the lanes actually load, the tiles' decoded stream starts on a 128-B line (4 KiB for large buffers)
and their packed stream on a 64-B segment, every letter and every packed dword is owned by exactly one of {tiles, edge
items}, and under CNT_TAIL_LUT no tile reaches the final partial word (BYTE_LUT's, n_to_bits.rs:109-111).  The GPU tests
run a few hundred of these phase combinations; the bug fixed in round 4 (a tile's last letters inside that word) lived in
370 000 of them and in none of the ones the GPU tests had picked."""
import ctypes

import pytest

CNT_STRICT_LUT, CNT_TAIL_LUT = 1, 4
TILE = 4096


@pytest.fixture(scope="module")
def L():
    """the TEST-HOOKS build (tests/libcute_nt_hip_hooks.so, -DCNT_TEST_HOOKS): the product library exports no cnt_test_* symbol"""
    from cute_nucleotides_amd import _lib, build

    build.build_hooks()
    prev = _lib.use_build("hooks")
    lib = _lib.lib()
    _lib.use_build(prev)
    return lib


def _plan(L, a_n, a_bits, a_back, n_len, flags):
    out = (ctypes.c_uint64 * 8)()
    assert L.cnt_test_round_trip_plan(a_n, a_bits, a_back, n_len, flags, out) == 0
    fast, t0, p0, tiles, w0, phase, phase2, slack_vecs = [int(x) for x in out]
    return fast, t0, p0, tiles, w0 - (1 << 64) if w0 >> 63 else w0, phase, phase2, slack_vecs


def _check(L, a_n, a_bits, a_back, n_len, flags):
    fast, t0, p0, tiles, w0, phase, phase2, slack_vecs = _plan(L, a_n, a_bits, a_back, n_len, flags)
    where = (a_n, a_bits, a_back, n_len, flags)
    if fast:
        assert a_n % 128 == 0 and a_bits % 128 == 0 and a_back % 128 == 0 and n_len >= TILE, where
        return 0
    if tiles == 0:
        return 0
    t1, p1 = t0 + TILE * tiles, p0 + (TILE // 16) * tiles
    words = (n_len + 31) // 32
    dwords = 2 * words
    # stores: decoded stream on its grain, packed stream on a 64-B segment
    assert (a_back + t0) % 128 == 0 and (a_bits + 4 * p0) % 64 == 0, where
    # the head left to the edge items: under two pages (large buffers: the plan looks that far for a cheap start), one line otherwise
    assert t0 < (8192 + 128 if n_len >= (1 << 20) else 256 + 128), where
    # window: 128-B aligned, starts inside the buffer, holds both first nucleotides
    assert w0 >= 0 and (a_n + w0) % 128 == 0 and phase == t0 - w0 and phase2 == 16 * p0 - w0, where
    assert 0 <= phase < 128 + 256 and 0 <= phase2 < 128 + 256, where
    q, q2 = phase >> 4, phase2 >> 4
    # the fifth load's lanes 0..ceil(max(phase, phase2) / 16)-1 fetch vectors 256..: inside the descriptor's slack; the funnels'
    # highest slab index 255 + max(q, q2) + 1 inside the slab (5 rows of 64)
    extra = (max(phase, phase2) + 15) >> 4
    assert extra <= slack_vecs and extra <= 64 and 255 + max(q, q2) + 1 < 320, where
    # the last tile's window ends inside the buffer (its highest vector is loaded whole)
    assert w0 + TILE * tiles + 16 * extra <= n_len, where
    # ownership: letters [t0, t1) and packed dwords [p0, p1) are the tiles'; the edge dword set [0, h) U [f, dwords) owns the rest
    h, f = max(p0, (t0 + 15) >> 4), min(p1, t1 >> 4)
    assert t1 <= n_len and p1 <= dwords and h <= f, where  # h <= f: the two edge ranges do not overlap
    assert (t0 + 15) >> 4 <= h and p0 <= h and f <= p1 and f <= t1 >> 4, where  # every letter < t0 / >= t1 and every dword < p0 / >= p1 has an edge item
    if (flags & CNT_TAIL_LUT) and not (flags & CNT_STRICT_LUT) and n_len % 32:
        w = n_len & ~31  # the final partial word starts here
        assert t1 <= w and 16 * p1 <= w, where
    return tiles


def test_every_phase_small_buffers(L):
    """grain 128: all 128 x 8 x 128 phases, lengths at the first boundaries where tiles appear and around a later tile end"""
    checked = with_tiles = 0
    for a_n in range(128):
        for a_bits in range(0, 64, 8):
            for a_back in range(128):
                base = 0x7F0000000000
                an, ab, ak = base + a_n, base + (1 << 30) + a_bits, base + (2 << 30) + a_back
                for flags in (0, CNT_TAIL_LUT):
                    for n_len in (4096 + 127 + 143 + 32, 3 * 4096 + 411, 3 * 4096 + 439):
                        with_tiles += 1 if _check(L, an, ab, ak, n_len, flags) else 0
                        checked += 1
    assert checked == 128 * 8 * 128 * 6 and with_tiles > checked // 2


def test_lengths_sweep_a_whole_period_for_sampled_phases(L):
    """every length of a 4-KiB period (+ 64) for 200 sampled phase triples, with and without CNT_TAIL_LUT: the distance
    between the last tile and the end of the input takes every value"""
    import random

    rnd = random.Random(4)
    for _ in range(200):
        a_n, a_bits, a_back = 0x7E0000000000 + rnd.randrange(128), 0x7E1000000000 + 8 * rnd.randrange(8), 0x7E2000000000 + rnd.randrange(128)
        for n_len in range(3 * 4096, 4 * 4096 + 64):
            _check(L, a_n, a_bits, a_back, n_len, CNT_TAIL_LUT)
            _check(L, a_n, a_bits, a_back, n_len, 0)


def test_large_buffers_price_two_pages_of_starts(L):
    """n >= 2^20: every start on a line of d_back within two pages x both segment choices is priced (read-ahead lines, window on
    a page of d_n); sampled output phases mod 4096 x input phases x packed phases.  Whenever SOME start reads one line ahead from a
    page-aligned window, the plan's does."""
    import random

    rnd = random.Random(5)
    for _ in range(4000):
        a_n, a_bits, a_back = 0x7D0000000000 + rnd.randrange(4096), 0x7D1000000000 + 8 * rnd.randrange(16), 0x7D2000000000 + rnd.randrange(4096)
        n_len = (1 << 20) + rnd.randrange(1 << 16)
        assert _check(L, a_n, a_bits, a_back, n_len, rnd.choice((0, CNT_TAIL_LUT, CNT_STRICT_LUT))) or (a_n % 128 == 0 and a_bits % 128 == 0 and a_back % 128 == 0)
        fast, t0, p0, tiles, w0, phase, phase2, _ = _plan(L, a_n, a_bits, a_back, n_len, 0)
        if fast:
            continue
        # the cheapest kind of start exists iff phi = (d_n - d_back) mod 128 and psi = the packed stream's offset from a page-aligned
        # window (mod 256) both leave the read-ahead inside one line: ceil(max / 16) <= 8 vectors
        phi = (a_n - a_back) % 128
        psi = (16 * ((-a_bits) % 64 // 4) + a_n) % 256
        if max(phi, psi) <= 128:
            assert ((a_n + w0) % 4096 == 0 and max(phase, phase2) <= 128) or max(phase, phase2) == 0, (a_n, a_bits, a_back, n_len, phase, phase2, phi, psi)


def test_tiny_and_degenerate_inputs(L):
    for n_len in (1, 31, 32, 100, 127, 128, 400, 4095, 4096, 4097, 4096 + 143, 4096 + 144):
        for a_back in (0, 1, 127):
            for a_n in (0, 1, 127):
                _check(L, 0x7C0000000000 + a_n, 0x7C1000000008, 0x7C2000000000 + a_back, n_len, CNT_TAIL_LUT)
    out = (ctypes.c_uint64 * 8)()
    assert L.cnt_test_round_trip_plan(0, 4, 0, 100, 0, out) != 0  # packed pointers are 8-byte aligned
    assert L.cnt_test_round_trip_plan(0, 0, 0, 100, 0x80, out) != 0  # unknown flag
