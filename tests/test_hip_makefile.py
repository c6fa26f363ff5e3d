"""`hip/` is the directory BASELINE.json's north_star names ("a new `hip/` directory holds the kernels and C-ABI shim"): the
kernels, the C-ABI shim and a Makefile that builds them WITHOUT Python -- what a Rust or C++ consumer (rust/build.rs) runs.
The Python package builds the same translation unit with its own hipcc line (cute_nucleotides_amd/build.py).  Two recipes
must not drift: this test builds the product through `make -C hip`, pulls the gfx950 code object out of both shared
libraries and compares them instruction for instruction, and holds both dynamic symbol tables to the header."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _device_disassembly(lib, tmp, tag):
    """gfx950 disassembly of the code object embedded in a HIP shared library (its .hip_fatbin section, unbundled)"""
    fat, co = os.path.join(tmp, tag + ".fatbin"), os.path.join(tmp, tag + ".co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, tag + ".copy")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co])
    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    # what may differ: the file name in the banner, and __hip_cuid_<hash> (named after a hash of the compiler's command line)
    return re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid", "\n".join(l for l in text.splitlines() if "file format" not in l))


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("cnt_")})


@pytest.fixture(scope="module")
def made():
    if not shutil.which("make") or not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")):
        pytest.skip("needs make and the ROCm LLVM tools")
    with tempfile.TemporaryDirectory(prefix="cnt_make_") as tmp:
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "hip"), "OUT=" + tmp, "product"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        yield tmp, os.path.join(tmp, "libcute_nt_hip.so")


def test_make_builds_the_product_with_build_pys_device_code(made):
    from cute_nucleotides_amd import build

    tmp, lib = made
    ours = build.build()
    a, b = _device_disassembly(lib, tmp, "make"), _device_disassembly(ours, tmp, "py")
    assert a.count("s_endpgm") > 30
    assert a == b, "hip/Makefile and cute_nucleotides_amd/build.py no longer compile the same device code"
    assert _exported(lib) == _exported(ours)  # ... and the same C ABI: exactly the header's product section (tests/test_abi.py)
    assert not [n for n in _exported(lib) if n.startswith("cnt_test_")]


def test_make_is_incremental_and_knows_its_inputs(made):
    tmp, lib = made
    t0 = os.path.getmtime(lib)
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "hip"), "OUT=" + tmp, "product"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.getmtime(lib) == t0  # nothing to do
    # every file of the translation unit is a prerequisite (an edit to any .hpp / .inc / the header rebuilds)
    deps = subprocess.run(["make", "-C", os.path.join(ROOT, "hip"), "-pn", "OUT=" + tmp, "product"], capture_output=True, text=True).stdout
    line = next(l for l in deps.splitlines() if l.startswith(tmp + "/libcute_nt_hip.so:"))
    for f in sorted(os.listdir(os.path.join(ROOT, "hip"))):
        if f.endswith((".hip", ".hpp", ".inc")):
            assert f in line.split(), f
    assert "../include/cute_nt.h" in line.split()


def test_both_recipes_use_the_same_flags():
    mk = open(os.path.join(ROOT, "hip", "Makefile")).read()
    py = open(os.path.join(ROOT, "cute_nucleotides_amd", "build.py")).read()
    assert "HIPCCFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17" in mk and "ARCH ?= gfx950" in mk
    assert '"--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC"' in py and 'ARCH = "gfx950"' in py
    for define, target in (("-DCNT_LAB_VARIANTS -DCNT_TEST_HOOKS", "lab"), ("-DCNT_TEST_HOOKS", "hooks")):
        assert define in mk and (target + ":") in mk
    # the Rust crate builds through the Makefile when it is not pointed at a prebuilt library
    rs = open(os.path.join(ROOT, "rust", "build.rs")).read()
    assert "CUTE_NT_LIB_DIR" in rs and '"make"' in rs and "OUT_DIR" in rs and "CARGO_FEATURE_HIP" in rs
