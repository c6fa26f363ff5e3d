#!/usr/bin/env python3
"""What the compiler made of the shipped kernels: registers, LDS, scratch and the instructions the design rests on,
per kernel, from the gfx950 assembly of the library's one translation unit (no GPU needed: hipcc cross-compiles).

    python bench/isa_digest.py            # print the digest
    python bench/isa_digest.py --write    # refresh profiles/r06_isa_digest.txt

tests/test_isa_digest.py (CPU box, -m "not gpu") regenerates the digest, compares it with the committed file and
asserts the properties DESIGN.md argues from: streaming loads with `nt`, write-through stores `sc0 sc1 nt`,
`v_perm_b32` as the letter table, `v_dot4_u32_u8` in the 5-letter packer, no waterfall loops (the
`s_xor_b64 exec, exec` / `s_cbranch_execnz` pair around a buffer access that a non-uniform descriptor produces),
no scratch, and register counts that keep the residency caps meaningful.  A compiler bump that changes any of it
fails on the CPU box instead of showing up as a few percent on the GPU."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hip", "cute_nt.hip")
DIGEST = os.path.join(ROOT, "profiles", "r06_isa_digest.txt")
DIGEST_R05 = os.path.join(ROOT, "profiles", "r05_isa_digest.txt")  # round 6 added kernels and moved none: every line of round 5's is a line of round 6's
DIGEST_R04 = os.path.join(ROOT, "profiles", "r04_isa_digest.txt")  # round 4's: the kernels both rounds ship must not have moved
DIGEST_R03 = os.path.join(ROOT, "profiles", "r03_isa_digest.txt")  # ... nor since round 3

# the kernels the default paths launch (demangled prefix up to the template arguments' closing bracket)
SHIPPED = [
    ("encode", "void cnt::n_to_bits_stream<64, 2, 1, 2, 19, false>"),
    ("encode, any input phase", "void cnt::n_to_bits_window<4, 1, 2, 19, false>"),
    ("decode", "void cnt::bits_to_n_stream<64, 4, 4, 0, 19>"),
    ("decode, any output phase", "void cnt::bits_to_n_shifted<64, 4, 4, 0, 19>"),
    ("decode, any packed phase past the Infinity Cache", "void cnt::bits_to_n_window<4, 0, 19>"),
    ("fused round trip", "void cnt::round_trip_stream<64, 4, 1, 2, 19, false>"),
    ("fused round trip, any alignment", "void cnt::round_trip_window<1, 2, 19, false>"),
    ("5-letter encode", "void cnt::n_to_bits2_wave<1, 2, 2, 16, false, 1>"),
    ("5-letter encode, any input phase", "void cnt::n_to_bits2_window<2, 16, false, 1>"),
    ("5-letter decode", "void cnt::bits_to_n2_wave<1, 2, 0, 19, 4>"),
    ("hamming", "void cnt::hamming_persist<8>"),
    ("validate", "void cnt::validate_persist<16, false>"),
    ("complement", "void cnt::complement_tiles<256, 1>"),
    ("reverse complement", "void cnt::reverse_complement_tiles<256>"),
    # round 6: the validated twins -- the same tile bodies with the count behind the stores
    ("validated encode", "void cnt::n_to_bits_stream_checked<64, 2, 1, 2, 19, false>"),
    ("validated encode, any input phase", "void cnt::n_to_bits_window_checked<4, 1, 2, 19, false>"),
    ("validated fused round trip", "void cnt::round_trip_stream_checked<64, 4, 1, 2, 19, false>"),
    ("validated fused round trip, any alignment", "void cnt::round_trip_window_checked<1, 2, 19, false>"),
    ("validated 5-letter encode", "void cnt::n_to_bits2_wave_checked<2, 2, 16, false, 1>"),
    ("validated 5-letter encode, any input phase", "void cnt::n_to_bits2_window_checked<2, 16, false, 1>"),
]
COUNTED = ("buffer_load_dwordx4", "buffer_load_dwordx2", "buffer_load_dword ", "buffer_store_dwordx4", "buffer_store_dwordx2", "buffer_store_dword ",
           "global_load", "global_store", "v_perm_b32", "v_mul_lo_u32", "v_dot4_u32_u8", "v_alignbit_b32", "v_pk_", "v_bitop3_b32", "ds_read", "ds_write",
           "ds_bpermute", "v_readfirstlane_b32", "s_and_saveexec_b64", "s_xor_b64 exec, exec", "s_cbranch_execnz", "s_load_dword", "s_waitcnt", "scratch_")


EXTRA = ("v_sad_u8", "v_bcnt_u32_b32", "global_atomic")  # counted for the tests, not printed (round 5's digest lines stay comparable)


def hipcc():
    import shutil

    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def assembly(lab=False, hooks=False):
    """demangled gfx950 assembly of the library's device code, same flags as cute_nucleotides_amd/build.py (the PRODUCT
    build; lab=True adds -DCNT_LAB_VARIANTS, hooks=True -DCNT_TEST_HOOKS)"""
    with tempfile.TemporaryDirectory(prefix="cnt_isa_") as tmp:
        out = os.path.join(tmp, "cute_nt.s")
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S"] + (["-DCNT_LAB_VARIANTS"] if lab else []) + (["-DCNT_TEST_HOOKS"] if hooks else []) +
                              ["-o", out, SRC], stderr=subprocess.DEVNULL)
        cxxfilt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
        if not os.path.exists(cxxfilt):
            cxxfilt = "c++filt"
        return subprocess.run([cxxfilt], stdin=open(out), capture_output=True, text=True, check=True).stdout


def kernels(asm):
    """{demangled name (without the parameter list): {"body": [instructions], "meta": {...}}}"""
    found = {}
    lines = asm.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(void cnt::[^\n]*?>|void cnt::\w+)\((.*)\): +; @", lines[i])
        if m:
            name = m.group(1)
            body = []
            i += 1
            while i < len(lines) and not lines[i].startswith(".Lfunc_end") and ".amdhsa_kernel" not in lines[i]:
                t = lines[i].strip()
                if t and not t.startswith((";", ".")) and not t.endswith(":"):
                    body.append(t.split(";")[0].strip())
                i += 1
            meta = {}
            while i < len(lines) and ".end_amdhsa_kernel" not in lines[i]:
                mm = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|kernarg_size)\s+(\d+)", lines[i])
                if mm:
                    meta[mm.group(1)] = int(mm.group(2))
                i += 1
            found[name] = {"body": body, "meta": meta}
        i += 1
    return found


def main_path(body):
    """the instructions of a tile up to and including its last buffer store (what follows in the codec kernels is the
    edge-workgroup branch, byte-granular and rare by construction)"""
    last = max((k for k, ins in enumerate(body) if ins.startswith("buffer_store")), default=len(body) - 1)
    return body[: last + 1]


def summarise(entry, tile_only):
    body = main_path(entry["body"]) if tile_only else entry["body"]
    counts = {key.strip(): sum(1 for ins in body if key in ins + " ") for key in COUNTED + EXTRA}
    pol_loads = sorted({" ".join(w for w in ins.split() if w in ("nt", "sc0", "sc1")) or "plain" for ins in body if ins.startswith("buffer_load")})
    pol_stores = sorted({" ".join(w for w in ins.split() if w in ("nt", "sc0", "sc1")) or "plain" for ins in body if ins.startswith("buffer_store")})
    # how many of the tile's global loads are in flight when the wave first waits for memory: every one of them should be
    # (a lane-masked partial load that the compiler moves behind an exec branch AND behind the other loads' s_waitcnt
    # makes a tile pay two dependent trips to memory: round 4's n_to_bits2_window, 4.0 ms instead of 3.6)
    first_wait = next((k for k, ins in enumerate(body) if ins.startswith("s_waitcnt") and "vmcnt" in ins), len(body))
    ahead = sum(1 for ins in body[:first_wait] if ins.startswith("buffer_load"))
    return {"instructions": len(body), "counts": {k: v for k, v in counts.items() if v}, "load_policies": pol_loads, "store_policies": pol_stores,
            "loads_before_first_wait": ahead, "loads": sum(1 for ins in body if ins.startswith("buffer_load"))}


def digest(found=None):
    found = found or kernels(assembly())
    out = ["# gfx950 ISA digest of the shipped kernels (bench/isa_digest.py; hipcc -O3 --offload-arch=gfx950, flags of build.py)",
           "# `tile` = instructions up to the tile's last buffer store; the codec kernels' edge branch behind it is listed as `whole`"]
    for label, name in SHIPPED:
        if name not in found:
            out.append("%s: %s  MISSING" % (label, name))
            continue
        e = found[name]
        m = e["meta"]
        t, w = summarise(e, True), summarise(e, False)
        out.append("%s: %s" % (label, name))
        out.append("  vgpr %d  sgpr %d  lds_static %d B  scratch %d B  kernarg %d B" % (m.get("next_free_vgpr", -1), m.get("next_free_sgpr", -1),
                   m.get("group_segment_fixed_size", -1), m.get("private_segment_fixed_size", -1), m.get("kernarg_size", -1)))
        out.append("  tile:  %d instructions; loads [%s]; stores [%s]" % (t["instructions"], ", ".join(t["load_policies"]), ", ".join(t["store_policies"])))
        out.append("         %d of %d global loads issued before the first wait for memory" % (t["loads_before_first_wait"], t["loads"]))
        out.append("         " + "  ".join("%s=%d" % kv for kv in sorted(t["counts"].items()) if kv[0] not in EXTRA))
        out.append("  whole: %d instructions  %s" % (w["instructions"], "  ".join("%s=%d" % (k, w["counts"][k]) for k in ("s_and_saveexec_b64", "s_xor_b64 exec, exec", "s_cbranch_execnz", "scratch_") if k in w["counts"])))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = digest()
    if "--write" in sys.argv:
        open(DIGEST, "w").write(text)
    sys.stdout.write(text)
