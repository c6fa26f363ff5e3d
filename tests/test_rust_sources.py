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


def test_binding_keeps_the_reference_signatures_and_is_well_formed():
    """The four mirrored functions carry the reference's signatures to the letter (n_to_bits.rs:34,51; n_to_bits2.rs:37,78),
    the `_into` forms and the queue type added in round 4 are there, every `unsafe` call site checks the status, and both
    Rust files are at least lexically well formed (balanced delimiters outside strings, chars and comments) -- what can be
    said without a compiler."""
    hip = open(os.path.join(ROOT, "rust", "src", "hip.rs")).read()
    for sig in (r"pub fn n_to_bits_hip\(n: &\[u8\]\) -> Vec<u64> \{", r"pub fn bits_to_n_hip\(bits: &\[u64\], len: usize\) -> Vec<u8> \{",
                r"pub fn n_to_bits2_hip\(n: &\[u8\]\) -> Vec<u64> \{", r"pub fn bits_to_n2_hip\(bits: &\[u64\], len: usize\) -> Vec<u8> \{",
                r"pub fn n_to_bits_hip_into\(n: &\[u8\], res: &mut Vec<u64>\) \{", r"pub fn bits_to_n_hip_into\(bits: &\[u64\], len: usize, res: &mut Vec<u8>\) \{",
                r"pub fn n_to_bits2_hip_into\(n: &\[u8\], res: &mut Vec<u64>\) \{", r"pub fn bits_to_n2_hip_into\(bits: &\[u64\], len: usize, res: &mut Vec<u8>\) \{",
                r"pub struct ShardedDevQueue \{", r"impl Drop for ShardedDevQueue \{", r"pub fn enqueue_n_to_bits\(&mut self,", r"pub fn enqueue_bits_to_n\(&mut self,",
                r"pub fn wait\(&mut self\) -> Vec<f32> \{"):
        assert re.search(sig, hip), sig
    # the queue's shard count is what the LIBRARY resolved (cnt_sharded_dev_shards), never the caller's
    # `ndev` (0 = all devices would have sized wait()'s and op_ms()'s buffers with zero floats for the C side to overrun)
    q = hip.split("impl ShardedDevQueue {", 1)[1].split("impl Drop for ShardedDevQueue", 1)[0]
    assert "cnt_sharded_dev_shards(handle, &mut n)" in q and "ndev: n as usize" in q and "ShardedDevQueue { handle, ndev }" not in q
    assert q.count("ShardedDevQueue::adopt(handle)") == 3 and "vec![0f32; self.ndev]" in q  # new / on_streams / on_devices
    for sig in (r"pub fn on_streams\(streams: &\[\*mut c_void\], timed: bool\) -> ShardedDevQueue \{", r"pub fn wait_event\(&mut self, k: usize, event: \*mut c_void\) \{",
                r"pub fn record_event\(&mut self, k: usize, event: \*mut c_void\) \{"):
        assert re.search(sig, hip), sig
    # round 6: the validated encode (one pass), the explicit device list, and where the queue runs a shard
    for sig in (r"pub fn n_to_bits_hip_checked\(n: &\[u8\]\) -> \(Vec<u64>, u64\) \{", r"pub fn try_n_to_bits_hip\(n: &\[u8\]\) -> Result<Vec<u64>, u64> \{",
                r"pub fn n_to_bits2_hip_checked\(n: &\[u8\]\) -> \(Vec<u64>, u64\) \{", r"pub fn n_to_bits_hip_checked_dev\(d_n: &DeviceBuffer, n_len: usize, d_out: &DeviceBuffer, d_invalid: &DeviceBuffer\) \{",
                r"pub fn on_devices\(devices: &\[i32\], timed: bool\) -> ShardedDevQueue \{", r"pub fn device\(&self, k: usize\) -> i32 \{",
                r"pub fn enqueue_n_to_bits_checked\(&mut self,"):
        assert re.search(sig, hip), sig
    # round 6: slices the caller owns (the forms that can take pinned memory), the pinned allocator and the in-place pin
    for sig in (r"pub fn n_to_bits_hip_slice\(n: &\[u8\], out: &mut \[u64\]\) -> usize \{", r"pub fn bits_to_n_hip_slice\(bits: &\[u64\], len: usize, out: &mut \[u8\]\) \{",
                r"pub struct PinnedBuf<T: Copy> \{", r"impl<T: Copy> Drop for PinnedBuf<T> \{", r"pub struct HostPin<'a, T> \{", r"impl<'a, T> Drop for HostPin<'a, T> \{",
                r"pub fn is_pinned<T>\(s: &\[T\]\) -> bool \{"):
        assert re.search(sig, hip), sig
    # the reference's panic text wherever a decoder checks `len` (n_to_bits.rs:52-54)
    assert hip.count('panic!("The length is greater than the number of nucleotides!")') >= 6
    # every call of a status-returning C symbol goes through check(...) (the three Drop impls and the bool probe excepted)
    calls = re.findall(r"\b(cnt_\w+)\(", hip.split('extern "C" {', 1)[1].split("}", 1)[1])
    body = hip.split("fn check(status: c_int)", 1)[1]
    for name in set(calls) - {"cnt_strerror", "cnt_words_for", "cnt_words2_for", "cnt_host_is_pinned"}:  # the last answers 1 / 0, not a status
        sites = [m.start() for m in re.finditer(r"\b%s\(" % name, body)]
        for s in sites:
            line = body[body.rfind("\n", 0, s) + 1 : body.find("\n", s)]
            assert "check(" in line or "fn drop" in body[max(0, s - 120) : s] or "== 0" in line, (name, line.strip())
    for rel in (("rust", "src", "hip.rs"), ("rust", "benches", "bench_n_to_bits.rs"), ("rust", "src", "lib.rs"), ("rust", "build.rs")):
        text = open(os.path.join(ROOT, *rel)).read()
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r'b?"(?:\\.|[^"\\])*"', '""', text)
        text = re.sub(r"b?'(?:\\.|[^'\\])'", "' '", text)
        stack = []
        for ch in text:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and "([{".index(stack.pop()) == ")]}".index(ch), rel
        assert not stack, rel


def _split_top_level(args):
    """split an argument list on commas that are not inside (), [], {} or <>"""
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_every_ffi_call_site_passes_as_many_arguments_as_the_extern_declares():
    """No compiler has seen rust/src/hip.rs: the one class of error a text check CAN rule out in the FFI layer is arity -- every
    call of a `cnt_*` symbol in the wrappers hands over exactly as many arguments as its `extern "C"` declaration (which
    tests/test_host_mirrors.py holds to the header) lists.  A wrapper that drifted when an entry point grew a parameter would
    otherwise only be found by the first user with a toolchain."""
    hip = open(os.path.join(ROOT, "rust", "src", "hip.rs")).read()
    hip = re.sub(r"//[^\n]*", "", hip)
    block = re.search(r'extern "C" \{(.*?)\n\}', hip, re.S).group(1)
    arity = {name: len(_split_top_level(args)) for name, args in re.findall(r"fn (cnt_\w+)\((.*?)\)", block, re.S)}
    assert len(arity) >= 42
    body = hip.split('extern "C" {', 1)[1].split("\n}", 1)[1]
    calls = 0
    for m in re.finditer(r"\b(cnt_\w+)\(", body):
        name, i = m.group(1), m.end()
        depth, j = 1, i
        while depth:  # the matching parenthesis
            depth += {"(": 1, ")": -1}.get(body[j], 0)
            j += 1
        got = len(_split_top_level(body[i : j - 1]))
        assert name in arity, "%s is called but not declared in the extern block" % name
        assert got == arity[name], "%s: %d arguments at a call site, %d parameters declared: %s" % (name, got, arity[name], body[i : j - 1][:120])
        calls += 1
    assert calls >= 53 and set(arity) - {m.group(1) for m in re.finditer(r"\b(cnt_\w+)\(", body)} == set(), "every declared symbol is used"


def test_the_back_end_sits_behind_a_hip_cargo_feature_and_builds_through_hip_makefile():
    """north_star's crate shape -- `src/n_to_bits*.rs` gain HIP-backed variants, `a new
    hip/ directory holds the kernels and C-ABI shim` -- as far as text can show it without cargo: the binding module exists only
    with the `hip` feature, build.rs does nothing without it, builds ../hip through its Makefile into OUT_DIR when no prebuilt
    library is named, and links what it built; the bench needs the feature."""
    cargo = open(os.path.join(ROOT, "rust", "Cargo.toml")).read()
    feats = cargo.split("[features]", 1)[1].split("[[bench]]", 1)[0]
    assert re.search(r"^hip = \[\]$", feats, re.M) and re.search(r'^standalone = \["hip"\]$', feats, re.M) and re.search(r'^default = \["hip"\]$', feats, re.M)
    lib = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    assert re.search(r'#\[cfg\(feature = "hip"\)\]\s*\npub mod hip;', lib) and lib.count("pub mod") == 1
    build = open(os.path.join(ROOT, "rust", "build.rs")).read()
    gate = build.index('env::var_os("CARGO_FEATURE_HIP").is_none()')
    assert build.index("return;", gate) < build.index("rustc-link-lib")  # nothing is linked without the feature
    assert 'Command::new("make")' in build and '.arg("-C")' in build and 'format!("OUT={}"' in build and '.arg("product")' in build
    assert 'env::var("CUTE_NT_LIB_DIR")' in build and 'env::var("OUT_DIR")' in build and 'join("hip")' in build
    assert os.path.exists(os.path.join(ROOT, "hip", "Makefile")) and os.path.exists(os.path.join(ROOT, "hip", "cute_nt.hip"))
    assert not os.path.exists(os.path.join(ROOT, "cute_nucleotides_amd", "csrc"))  # one home for the kernels
