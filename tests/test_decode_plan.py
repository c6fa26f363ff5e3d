"""Where cnt_bits_to_n_dev lays its tiles (hip/device_tier.inc decode_plan / decode_turn_pages, reached through the test hook
cnt_test_decode_plan -- no device needed), for EVERY packed offset and output phase:

  * the output is peeled to its 128-B line, from 2^20 nt on to its 4-KiB page;
  * calls of more than 2^30 nt -- whose packed stream no longer fits the Infinity Cache; below that the turns' position measures
    as nothing (profiles/r05_decode_off_grid.md) -- peel up to three further output pages so that the XCD turns of four tiles
    start within 512 bytes of a page boundary of the PACKED buffer (round 5);
  * there, a packed stream off its 128-B lines or off its dwords goes to bits_to_n_window, whose first window lies inside the
    caller's buffer (the head grows by 128 B worth of packed words when it would not) and whose last window ends inside it;
    inside the cache the stream kernel (any dword phase) and the shifted kernel (a bit phase) stay;
  * the aligned call is untouched: nothing peeled, the stream kernel, every tile."""
import ctypes

import pytest

BIG, MID, SMALL = (1 << 32) + 12345, 1 << 24, 1 << 16


@pytest.fixture(scope="module")
def L():
    from cute_nucleotides_amd import _lib, build

    build.build_hooks()
    prev = _lib.use_build("hooks")
    lib = _lib.lib()
    _lib.use_build(prev)
    return lib


SPX = 1 << 30  # cnt_chip_cache_nt of an SPX MI355X: 8 XCDs x 2^27 nt (32 MiB of packed words each)


def _plan(L, a_bits, a_out, n_len, cache_nt=SPX):
    out = (ctypes.c_uint64 * 7)()
    assert L.cnt_test_decode_plan(a_bits, a_out, n_len, cache_nt, out) == 0
    return dict(zip(("head", "out_phase", "r", "sh", "q", "window", "tiles"), (int(x) for x in out)))


def _check_common(p, a_bits, a_out, n_len, grain, cache_nt=SPX):
    """what every plan satisfies, whatever the size"""
    where = (hex(a_bits), hex(a_out), n_len, p)
    assert (a_out + p["head"]) % grain == 0 and p["out_phase"] == (a_out + p["head"]) % 4096, where
    assert p["sh"] == 2 * (p["head"] % 16) and p["q"] == ((a_bits + 4 * (p["head"] // 16)) % 128) // 4, where
    assert p["window"] == (1 if (p["q"] or p["sh"]) and n_len > cache_nt else 0), where
    assert p["tiles"] == 0 or p["head"] + 4096 * p["tiles"] <= n_len, where  # no tiles: the generic kernel alone takes the call
    if p["window"]:
        if p["tiles"]:
            first = a_bits + 4 * (p["head"] // 16)          # the dword of the first tile's first nucleotide
            window = first - 4 * p["q"]                     # 128-B aligned, and INSIDE the buffer
            assert window % 128 == 0 and window >= a_bits, where
            extra = (p["q"] + (1 if p["sh"] else 0) + 3) // 4
            end = a_bits + 8 * ((n_len + 31) // 32)
            assert window + 1024 * p["tiles"] + 16 * extra <= end, where  # the last tile's window ends inside the packed words
        # ... and no more than one tile is given up for that (it joins the edge items)
        assert max(0, n_len - p["head"]) // 4096 - p["tiles"] <= 1, where
    else:
        assert p["tiles"] == max(0, n_len - p["head"]) // 4096, where


def test_turns_start_within_512_bytes_of_a_packed_page_for_calls_past_the_infinity_cache(L):
    base_bits, base_out = 0x7F0000000000, 0x7E0000000000
    seen_k = set()
    for p_off in range(0, 4096, 8):  # word pointers are 8-byte aligned
        for a_off in (0, 1, 5, 16, 77, 128, 1000, 2048, 4095):
            p = _plan(L, base_bits + p_off, base_out + a_off, BIG)
            _check_common(p, base_bits + p_off, base_out + a_off, BIG, 4096)
            peel = (4096 - a_off) % 4096
            pages = (p["head"] - peel) // 4096
            assert p["head"] >= peel and (p["head"] - peel) % 4096 == 0 and pages <= 7, (p_off, a_off, p)
            assert min(p["r"], 4096 - p["r"]) <= 512, (p_off, a_off, p)  # the rule: nearest page boundary of the packed buffer, either side
            seen_k.add(pages % 4)
            if pages > 3:  # four pages more: only to keep the first window inside the buffer
                assert p["window"] and (p["head"] - 4 * 4096) // 16 < p["q"], (p_off, a_off, p)
    assert seen_k == {0, 1, 2, 3}
    # a packed pointer a whole number of KiB off its page is put back on it, lines and all: the stream kernel
    for kib in (1, 2, 3):
        p = _plan(L, base_bits + 1024 * kib, base_out, BIG)
        assert p["r"] == 0 and p["head"] == 4096 * (4 - kib) and p["window"] == 0
    # on the grid nothing is peeled: the aligned call launches exactly what it always did
    assert _plan(L, base_bits, base_out, 1 << 34) == {"head": 0, "out_phase": 0, "r": 0, "sh": 0, "q": 0, "window": 0, "tiles": 1 << 22}


def test_calls_inside_the_infinity_cache_place_no_turns(L):
    base_bits, base_out = 0x7F0000000000, 0x7E0000000000
    for n_len in (MID, 1 << 30):
        for p_off in range(0, 4096, 8):
            for a_off in (0, 5, 16, 4095):
                p = _plan(L, base_bits + p_off, base_out + a_off, n_len)
                _check_common(p, base_bits + p_off, base_out + a_off, n_len, 4096)
                peel = (4096 - a_off) % 4096
                assert p["head"] == peel and p["window"] == 0, (p_off, a_off, p)  # the page peel and nothing else: round 4's launch
                assert p["r"] == (p_off + 4 * (peel // 16)) % 4096
    assert _plan(L, base_bits, base_out, MID)["head"] == 0


def test_small_calls_peel_to_a_line_only(L):
    base_bits, base_out = 0x7F0000000000, 0x7E0000000000
    for n_len in (1 << 12, SMALL, (1 << 20) - 1):
        for p_off in (0, 8, 64, 120, 128, 1000):
            for a_off in (0, 3, 16, 130, 127):
                p = _plan(L, base_bits + p_off, base_out + a_off, n_len)
                _check_common(p, base_bits + p_off, base_out + a_off, n_len, 128)
                peel = (128 - a_off % 128) % 128
                assert p["head"] == peel and p["window"] == 0, (n_len, p_off, a_off, p)
    out = (ctypes.c_uint64 * 7)()
    assert L.cnt_test_decode_plan(0x7F0000000004, 0, 1 << 20, SPX, out) != 0  # packed pointers are 8-byte aligned
    assert L.cnt_test_decode_plan(0, 0, 1 << 20, SPX, None) != 0
    assert L.cnt_test_decode_plan(0, 0, 1 << 20, 0, out) != 0  # a device always has a share of the cache


def test_short_and_ragged_lengths_never_plan_past_either_buffer(L):
    base_bits, base_out = 0x7F0000000000, 0x7E0000000000
    for n_len in (1, 31, 4095, 4096, 4097, 8191, 3 * 4096 + 77, (1 << 20) + 1, (1 << 20) + 4096 * 5 + 1234, (1 << 30) + 4097):
        for p_off in (0, 8, 24, 264, 1016, 2312, 4088):
            for a_off in (0, 5, 16, 77, 2048, 4095):
                grain = 4096 if n_len >= (1 << 20) else 128
                _check_common(_plan(L, base_bits + p_off, base_out + a_off, n_len), base_bits + p_off, base_out + a_off, n_len, grain)


def test_the_gate_follows_the_devices_share_of_the_infinity_cache(L):
    """the one MI355X constant that did not come from the device -- `len > 2^30` = "256 MiB of Infinity
    Cache" -- is now chip_info().cache_nt = XCDs x 2^27 nt, so a partitioned-mode device (1 / 2 / 4 XCDs, sharing the cache with
    its siblings) takes the past-the-cache plan from 2^27 / 2^28 / 2^29 nt on instead of running the cached plan on buffers
    that left its share long ago.  Walked for log2(XCDs) = 0..3 on both sides of each threshold."""
    base_bits, base_out = 0x7F0000000000, 0x7E0000000000
    for xs in range(4):
        cache_nt = (1 << xs) << 27
        for n_len, past in ((cache_nt, False), (cache_nt + 4096, True), (cache_nt // 2 + 77, False), (2 * cache_nt + 12345, True)):
            placed = 0
            for p_off in range(0, 4096, 136):
                for a_off in (0, 5, 16, 2048, 4095):
                    p = _plan(L, base_bits + p_off, base_out + a_off, n_len, cache_nt)
                    _check_common(p, base_bits + p_off, base_out + a_off, n_len, 4096, cache_nt)
                    peel = (4096 - a_off) % 4096
                    if past:
                        assert min(p["r"], 4096 - p["r"]) <= 512, (xs, n_len, p_off, a_off, p)  # the turns are placed ...
                        placed += p["head"] != peel
                    else:
                        assert p["head"] == peel and not p["window"], (xs, n_len, p_off, a_off, p)  # ... or nothing is, inside the cache
            assert (placed > 0) == past, (xs, n_len)
    # the same call on the other side of the gate depending on the device: 2^29 + 1 nt is cached on SPX and streams on a CPX partition
    a = _plan(L, base_bits + 264, base_out + 5, (1 << 29) + 1, 1 << 30)
    b = _plan(L, base_bits + 264, base_out + 5, (1 << 29) + 1, 1 << 27)
    assert not a["window"] and b["window"] and a["head"] != b["head"]
