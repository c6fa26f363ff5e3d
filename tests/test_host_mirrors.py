"""The non-Python host mirrors stay buildable: the C++ header-only mirror of the reference API
and the C++ twin of the criterion harness must compile against include/cute_nt.h (syntax +
types; no GPU needed), and the shipped Rust binding must declare exactly the C symbols it uses
with the header's argument lists."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_and_bench_twin_compile():
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    for src in ("cute_nucleotides_amd/cute_nucleotides.hpp", "bench/bench_n_to_bits.cpp"):
        r = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-x", "c++", os.path.join(ROOT, src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


def test_c_header_is_valid_c():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    r = subprocess.run([gcc, "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-pedantic", "-x", "c",
                        os.path.join(ROOT, "include", "cute_nt.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_rust_binding_matches_the_header():
    header = open(os.path.join(ROOT, "include", "cute_nt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    rust = open(os.path.join(ROOT, "rust", "src", "hip.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', rust, re.S).group(1)
    decls = re.findall(r"fn (cnt_\w+)\((.*?)\)", block, re.S)
    assert len(decls) >= 9
    for name, args in decls:
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, header, re.S)
        assert m, "%s is not declared in include/cute_nt.h" % name
        c_args = [a for a in (x.strip() for x in m.group(1).split(",")) if a and a != "void"]
        r_args = [a for a in (x.strip() for x in args.split(",")) if a]
        assert len(c_args) == len(r_args), (name, c_args, r_args)
        for ca, ra in zip(c_args, r_args):
            is_ptr_c, is_ptr_r = "*" in ca, "*" in ra
            assert is_ptr_c == is_ptr_r, (name, ca, ra)
            if is_ptr_c:
                assert ("const" in ca) == ("*const" in ra), (name, ca, ra)


def test_plain_c_program_links_and_runs():
    """the boundary is a C ABI: tests/c_link_check.c (C11, gcc) links libcute_nt_hip.so and runs the device-free
    entry points; on a box without a GPU the compute call answers CNT_ENODEV"""
    import sys

    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    sys.path.insert(0, ROOT)
    from cute_nucleotides_amd import build

    lib = build.build()
    exe = os.path.join(ROOT, "tests", "c_link_check")
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-pedantic", "-o", exe, os.path.join(ROOT, "tests", "c_link_check.c"),
                        "-L" + os.path.dirname(lib), "-lcute_nt_hip", "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    try:
        run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert run.returncode == 0, (run.returncode, run.stdout, run.stderr[-500:])
        assert "c link ok: abi 1" in run.stdout
    finally:
        os.remove(exe)
