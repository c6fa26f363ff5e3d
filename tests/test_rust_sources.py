"""The Rust side is shipped as source (no Rust toolchain in the image): these checks keep it honest without
a compiler -- the bench harness is real code (not comments), names the reference's groups and functions, uses
only std, and every `hip::` item it calls is defined in rust/src/hip.rs."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _code(path):
    lines = open(path).read().splitlines()
    return [l for l in lines if l.strip() and not l.strip().startswith("//")]


def test_rust_bench_harness_is_code_and_matches_the_reference_rows():
    path = os.path.join(ROOT, "rust", "benches", "bench_n_to_bits.rs")
    code = _code(path)
    assert len(code) > 150, "the harness must be code, not commented rows"
    text = "\n".join(code)
    # reference rows, benches/bench_n_to_bits.rs:15-20,31-32,44-47,59-60 -- same ids, plus one *_hip row per group
    for fn in ("n_to_bits_lut", "n_to_bits_pext", "n_to_bits_shift", "n_to_bits_movemask", "n_to_bits_mul", "memcpy",
               "n_to_bits2_lut", "n_to_bits2_pext", "bits_to_n_lut", "bits_to_n_shuffle", "bits_to_n_pdep", "bits_to_n_clmul",
               "bits_to_n2_lut", "bits_to_n2_pdep", "n_to_bits_hip", "bits_to_n_hip", "n_to_bits2_hip", "bits_to_n2_hip"):
        assert 'bench_function("%s"' % fn in text, fn
    for group in ('"n_to_bits"', '"n_to_bits2"', '"bits_to_n"', '"bits_to_n2"'):
        assert "Group::new(%s, 40000)" % group in text
    assert 'b"ATCG".repeat(repeat)' in text and 'b"ATCGN".repeat(repeat)' in text  # :68-74
    assert "Instant::now()" in text and "fn main()" in text
    uses = re.findall(r"^use (\S+?)[:;{]", text, re.M)
    assert set(uses) <= {"std", "cute_nucleotides", "cute_nucleotides_hip"}, uses  # std-only: no criterion


def test_every_hip_item_the_harness_calls_exists_in_the_binding():
    bench = "\n".join(_code(os.path.join(ROOT, "rust", "benches", "bench_n_to_bits.rs")))
    hip = open(os.path.join(ROOT, "rust", "src", "hip.rs")).read()
    for item in ("n_to_bits_hip", "bits_to_n_hip", "n_to_bits2_hip", "bits_to_n2_hip", "n_to_bits_hip_dev", "bits_to_n_hip_dev",
                 "device_sync", "shutdown", "words_for"):
        assert item + "(" in bench
        assert re.search(r"pub fn %s\b" % item, hip), item
    assert "DeviceBuffer::from_slice" in bench and "pub struct DeviceBuffer" in hip
    for method in ("new", "from_slice", "to_vec"):
        assert re.search(r"pub fn %s\b" % method, hip), method
    cargo = open(os.path.join(ROOT, "rust", "Cargo.toml")).read()
    assert 'name = "bench_n_to_bits"' in cargo and "harness = false" in cargo and "criterion" not in cargo
