"""Instruction selection of the shipped kernels, pinned on the CPU box: DESIGN.md argues from the
disassembly -- streaming `nt` loads, write-through `sc0 sc1 nt` stores, `v_perm_b32` letter tables, `v_dot4_u32_u8`
in the 5-letter packer, no waterfall loops, no scratch, register counts far below the residency caps -- and nothing
checked it: a compiler bump could reintroduce a waterfall loop and only show up as a few percent on the GPU box.
hipcc cross-compiles gfx950 without a GPU, so the assembly of the library's one translation unit is regenerated here
(~5 s), compared with the committed digest (profiles/r06_isa_digest.txt) and checked property by property.  Round 4 also
checks what the PRODUCT code object contains: the default kernels and the any-alignment kernels, none of the lab's."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bench"))


@pytest.fixture(scope="module")
def isa():
    import isa_digest

    found = isa_digest.kernels(isa_digest.assembly())
    return isa_digest, found


def _tile(isa_digest, found, name):
    assert name in found, "kernel not in the code object: " + name
    return isa_digest.summarise(found[name], True), isa_digest.summarise(found[name], False), found[name]["meta"]


def test_committed_digest_is_current(isa):
    isa_digest, found = isa
    want = open(isa_digest.DIGEST).read()
    got = isa_digest.digest(found)
    assert "MISSING" not in got
    assert got == want, "the compiler's output changed: review the diff, then `python bench/isa_digest.py --write`"


def test_defaults_have_not_moved_since_round_3(isa):
    """what a round changes in the shipped kernels is changed on purpose and named here --
    every other line of round 3's committed digest is still a line of round 4's, and every other line of round 4's a line of
    round 5's.  Round 5 touched NO tile: the decoders' edge items became 16-letter vectors (the `whole` instruction count and
    the register counts of `decode` and `decode, any output phase` moved, their tile lines did not), and bits_to_n_window joined
    them for calls past the Infinity Cache."""
    isa_digest, _ = isa

    def carried_over(old_path, new_path, whole_blocks=(), edge_only=()):
        now = open(new_path).read().split("\n")
        skip, edge = False, False
        for line in open(old_path).read().split("\n"):
            if line and not line.startswith((" ", "#")):
                skip, edge = line.startswith(whole_blocks), line.startswith(edge_only)
            if not line or skip or (edge and line.startswith(("  vgpr", "  whole:"))):
                continue
            assert line in now, "%s's digest line is gone from %s: %s" % (os.path.basename(old_path), os.path.basename(new_path), line)

    # round 4 on purpose: reverse complement's second load no longer waits for the first (branch-free funnel); the encode window
    # kernel takes 4-KiB tiles (its read-ahead line is 3 % of the tile's reads instead of 6 %)
    carried_over(isa_digest.DIGEST_R03, isa_digest.DIGEST_R04, whole_blocks=("reverse complement:", "encode, any input phase:"))
    carried_over(isa_digest.DIGEST_R04, isa_digest.DIGEST_R05, edge_only=("decode:", "decode, any output phase:"))
    # round 6 ADDED kernels (the validated twins) and split the encoders' bodies into shared templates: not one line of round 5's
    # digest moved -- the unflagged kernels are instruction for instruction what they were (VERDICT r05 next-1 "Done")
    carried_over(isa_digest.DIGEST_R05, isa_digest.DIGEST)


def test_validated_twins_are_the_same_tiles_with_a_tail_behind_the_stores(isa):
    """cnt_*_checked_dev: up to its last store a validated kernel is its unchecked twin (same loads in flight, same policies, same
    arithmetic); behind the stores sits the suspicion sum -- v_sad_u8, one per input dword -- ONE wave-uniform branch, and only
    behind that the exact recount and the wave's single atomic."""
    isa_digest, found = isa
    for plain, checked, dwords in (("void cnt::n_to_bits_stream<64, 2, 1, 2, 19, false>", "void cnt::n_to_bits_stream_checked<64, 2, 1, 2, 19, false>", 8),
                                   ("void cnt::n_to_bits_window<4, 1, 2, 19, false>", "void cnt::n_to_bits_window_checked<4, 1, 2, 19, false>", 20),
                                   ("void cnt::round_trip_stream<64, 4, 1, 2, 19, false>", "void cnt::round_trip_stream_checked<64, 4, 1, 2, 19, false>", 16),
                                   ("void cnt::round_trip_window<1, 2, 19, false>", "void cnt::round_trip_window_checked<1, 2, 19, false>", 20),
                                   ("void cnt::n_to_bits2_wave<1, 2, 2, 16, false, 1>", "void cnt::n_to_bits2_wave_checked<2, 2, 16, false, 1>", 16),
                                   ("void cnt::n_to_bits2_window<2, 16, false, 1>", "void cnt::n_to_bits2_window_checked<2, 16, false, 1>", 16)):
        tp, _, mp = _tile(isa_digest, found, plain)
        tc, wc, mc = _tile(isa_digest, found, checked)
        for key in ("buffer_load_dwordx4", "buffer_store_dword", "buffer_store_dwordx4", "buffer_store_dwordx2", "v_mul_lo_u32", "v_dot4_u32_u8", "v_alignbit_b32", "ds_read"):
            assert tc["counts"].get(key, 0) == tp["counts"].get(key, 0), (checked, key)
        assert tc["load_policies"] == tp["load_policies"] and tc["store_policies"] == tp["store_policies"]
        assert tc["loads_before_first_wait"] == tc["loads"] == tp["loads"]
        assert abs(tc["instructions"] - tp["instructions"]) <= 6, (checked, tc["instructions"], tp["instructions"])  # the tile is not longer for being checked
        assert "v_sad_u8" not in tc["counts"] and wc["counts"]["v_sad_u8"] == dwords, (checked, wc["counts"].get("v_sad_u8"))  # behind the stores
        body = found[checked]["body"]
        last_store = max(k for k, ins in enumerate(body) if ins.startswith("buffer_store"))
        first_sad = min(k for k, ins in enumerate(body) if ins.startswith("v_sad_u8"))
        branch = next(k for k, ins in enumerate(body) if k > first_sad and ins.startswith("s_cbranch"))
        first_atomic = min(k for k, ins in enumerate(body) if "atomic_add" in ins)
        assert last_store < first_sad < branch < first_atomic, (checked, last_store, first_sad, branch, first_atomic)
        assert mc["private_segment_fixed_size"] == 0 and mc["next_free_vgpr"] <= 84
        assert mc["group_segment_fixed_size"] == mp["group_segment_fixed_size"]


def test_product_code_object_holds_no_lab_kernels(isa):
    """the product library = defaults + the any-alignment kernels + generic / staged / utility kernels; every measured
    alternative (LDS-widened, pipelined, multi-wave, round 1's reductions, other shapes and policies) is lab-only"""
    isa_digest, found = isa
    names = sorted(found)
    for needle in ("n_to_bits_lds", "bits_to_n_lds", "n_to_bits2_pipe", "bits_to_n2_pipe", "hamming_tiles", "validate_tiles", "sum_partials"):
        assert not [n for n in names if needle in n], (needle, [n for n in names if needle in n])
    assert len([n for n in names if "cnt::n_to_bits_stream<" in n]) == 2  # default, strict and default
    assert len([n for n in names if "cnt::bits_to_n_stream<" in n]) == 1
    assert len([n for n in names if "cnt::n_to_bits2_wave<" in n]) == 2 and len([n for n in names if "cnt::bits_to_n2_wave<" in n]) == 1
    assert len([n for n in names if "cnt::round_trip_stream<" in n]) == 2 and len([n for n in names if "cnt::round_trip_window<" in n]) == 2
    assert len(names) < 60, len(names)


def test_test_hooks_build_has_the_products_device_code():
    """tests/libcute_nt_hip_hooks.so (-DCNT_TEST_HOOKS: cnt_test_alias_devices / _advise_output /
    _round_trip_plan) is what the N > 1 sharded tests load instead of the product -- its DEVICE code is the product's, byte for
    byte: the macro touches host code only, so what those tests exercise on the GPU is what ships."""
    import isa_digest

    import re

    # what may differ: source paths, the compiler banner, and __hip_cuid_<hash> (a one-byte symbol named after a hash of the
    # compiler's command line, which contains the -D); every instruction, kernel descriptor and metadata line may not
    strip = lambda asm: re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid", "\n".join(l for l in asm.splitlines() if not l.lstrip().startswith((".file", ".ident", "; "))))
    product, hooks = strip(isa_digest.assembly()), strip(isa_digest.assembly(hooks=True))
    assert product.count("s_endpgm") > 30 and hooks == product


def test_every_shipped_kernel_is_lean(isa):
    isa_digest, found = isa
    for label, name in isa_digest.SHIPPED:
        t, w, m = _tile(isa_digest, found, name)
        assert m["private_segment_fixed_size"] == 0 and "scratch_" not in w["counts"], (label, "scratch")
        # the residency caps are set with dummy LDS (<= 6 waves per SIMD): anything under 85 VGPRs changes nothing
        assert m["next_free_vgpr"] <= 84, (label, m)
        # a waterfall loop (non-uniform buffer descriptor) is `s_xor_b64 exec, exec` + `s_cbranch_execnz` around the access
        assert "s_xor_b64 exec, exec" not in w["counts"], (label, "waterfall loop")
        assert "global_load" not in t["counts"] and "global_store" not in t["counts"], (label, "tile accesses go through raw buffer ops")


def test_2bit_codec_instruction_selection(isa):
    isa_digest, found = isa
    t, w, m = _tile(isa_digest, found, "void cnt::n_to_bits_stream<64, 2, 1, 2, 19, false>")
    assert t["load_policies"] == ["nt"] and t["store_policies"] == ["sc0 nt sc1"]
    assert t["counts"]["buffer_load_dwordx4"] == 2 and t["counts"]["buffer_store_dword"] == 2
    assert "s_and_saveexec_b64" not in t["counts"] and "v_readfirstlane_b32" not in t["counts"]  # nothing divergent in front of the stores
    assert t["counts"]["v_mul_lo_u32"] == 8  # y*0x41041: the reference's n_to_bits_mul identity, found by the compiler
    assert m["group_segment_fixed_size"] == 0 and t["instructions"] <= 90
    t, w, m = _tile(isa_digest, found, "void cnt::n_to_bits_window<4, 1, 2, 19, false>")
    assert t["load_policies"] == ["nt"] and t["store_policies"] == ["sc0 nt sc1"] and t["counts"]["v_alignbit_b32"] == 4
    assert t["counts"]["buffer_load_dwordx4"] == 5 and "s_and_saveexec_b64" not in t["counts"]  # four of its own + the read-ahead
    for name in ("void cnt::bits_to_n_stream<64, 4, 4, 0, 19>", "void cnt::bits_to_n_shifted<64, 4, 4, 0, 19>", "void cnt::bits_to_n_window<4, 0, 19>"):
        t, w, m = _tile(isa_digest, found, name)
        assert t["store_policies"] == ["sc0 nt sc1"] and t["counts"]["buffer_store_dwordx4"] == 4
        assert t["counts"]["v_perm_b32"] == 16  # the 4-entry "ACTG" table, one per packed byte (four packed dwords per lane)
        assert "s_and_saveexec_b64" not in t["counts"] and m["next_free_vgpr"] <= 32
    # the window decoder: ONE line-aligned 16-B load per lane over the tile's 1 KiB + the vectors behind it (most lanes aim
    # past the descriptor: no branch), four funnel reads of the slab, the stream kernel's stores
    t, w, m = _tile(isa_digest, found, "void cnt::bits_to_n_window<4, 0, 19>")
    assert t["counts"]["buffer_load_dwordx4"] == 2 and t["loads"] == 2 and t["counts"]["v_alignbit_b32"] == 4 and t["counts"]["ds_read"] == 4

    t, w, m = _tile(isa_digest, found, "void cnt::round_trip_stream<64, 4, 1, 2, 19, false>")
    assert t["load_policies"] == ["nt"] and t["store_policies"] == ["sc0 nt sc1"]
    assert t["counts"]["buffer_load_dwordx4"] == 4 and t["counts"]["buffer_store_dword"] == 4 and t["counts"]["buffer_store_dwordx4"] == 4
    assert "s_and_saveexec_b64" not in t["counts"]
    # the any-alignment twin: five window loads (the fifth reaches past the descriptor for most lanes: no branch), two funnel
    # reads per output pair, the same stores
    t, w, m = _tile(isa_digest, found, "void cnt::round_trip_window<1, 2, 19, false>")
    assert t["load_policies"] == ["nt"] and t["store_policies"] == ["sc0 nt sc1"]
    assert t["counts"]["buffer_load_dwordx4"] == 5 and t["counts"]["buffer_store_dword"] == 4 and t["counts"]["buffer_store_dwordx4"] == 4
    assert t["counts"]["v_alignbit_b32"] == 8 and "s_and_saveexec_b64" not in t["counts"] and m["next_free_vgpr"] <= 64


def test_5letter_codec_instruction_selection(isa):
    isa_digest, found = isa
    t, w, m = _tile(isa_digest, found, "void cnt::n_to_bits2_wave<1, 2, 2, 16, false, 1>")
    assert t["counts"]["v_dot4_u32_u8"] == 26  # 13 byte dot products per word, two words per lane
    assert t["counts"]["v_perm_b32"] >= 14 and t["load_policies"] == ["nt"] and t["counts"]["buffer_store_dwordx2"] == 2
    assert 3400 <= m["group_segment_fixed_size"] <= 3584  # the wave's slab; the launcher's cap arithmetic assumes <= 3584
    t, w, m = _tile(isa_digest, found, "void cnt::bits_to_n2_wave<1, 2, 0, 19, 4>")
    assert t["counts"]["v_pk_"] >= 40 and t["counts"]["v_perm_b32"] >= 40  # packed 16-bit /25, /5 and the weave
    assert t["store_policies"] == ["sc0 nt sc1"] and t["counts"]["buffer_store_dwordx4"] == 4
    assert 3400 <= m["group_segment_fixed_size"] <= 3584


def test_every_tile_load_is_in_flight_before_the_first_wait(isa):
    """A lane-masked partial load can end up behind an exec branch AND behind the s_waitcnt of the loads in front of it (the
    5-letter window encoder did, round 4: its fourth load left only after half the wave had waited for the first three, two
    dependent trips to memory per tile, 4.0 ms instead of 3.6 at 2^34 nt).  Every tile kernel the default paths launch must
    have ALL its global loads issued when it first waits for memory."""
    isa_digest, found = isa
    for label, name in isa_digest.SHIPPED:
        if label in ("hamming", "validate"):  # persistent loops: loads and waits alternate by design
            continue
        t, _, _ = _tile(isa_digest, found, name)
        assert t["loads"] >= 1 and t["loads_before_first_wait"] == t["loads"], (label, t["loads_before_first_wait"], t["loads"])


def test_packed_ops_instruction_selection(isa):
    isa_digest, found = isa
    for name, loads in (("void cnt::hamming_persist<8>", 32), ("void cnt::validate_persist<16, false>", 32)):
        t, w, m = _tile(isa_digest, found, name)
        assert w["counts"]["buffer_load_dwordx4"] == loads and w["load_policies"] == ["nt"]  # prologue + steady-state loads of the software pipeline
        assert "buffer_store_dword" not in w["counts"] and m["next_free_vgpr"] >= 64  # the pipeline lives in registers: 16 vectors in flight
    for name in ("void cnt::complement_tiles<256, 1>", "void cnt::reverse_complement_tiles<256>"):
        t, w, m = _tile(isa_digest, found, name)
        assert t["store_policies"] == ["sc0 nt sc1"] and t["load_policies"] == ["nt"]
