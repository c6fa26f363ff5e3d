"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`python -m pytest tests/ -m "not gpu"` runs on a CPU-only box (oracle vs golden
vectors, host logic, C-ABI symbol/arg-check coverage, gloo world_size-2 sharding).
`python -m pytest tests/ -m gpu` needs one MI355X and calls the HIP path through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/libcnt_oracle.so via ctypes."""
    from oracle import cnt_oracle

    cnt_oracle.build()
    return cnt_oracle


@pytest.fixture()
def lab_build():
    """Run this test against bench/libcute_nt_hip_lab.so (-DCNT_LAB_VARIANTS: every kernel variant + cnt_set_tuning) instead of
    the product library, which has no run-time kernel selection.  Parity tests that need no knob run on the product build;
    the variant sweeps, the tile-map / several-launch-loop tests and everything that forces a path through a tuning key
    take this fixture (directly or through `tuning` / `no_small_path` / `launch_tiles` / `reduce_form`)."""
    from cute_nucleotides_amd import _lib

    prev = _lib.use_lab_build(True)
    assert _lib.lib().cnt_set_tuning(b"small_nt", 1 << 17) == 0  # really the lab build
    yield
    _lib.use_lab_build(prev)


@pytest.fixture()
def hooks_build():
    """Run this test against tests/libcute_nt_hip_hooks.so (-DCNT_TEST_HOOKS: the product's sources and kernels plus the
    cnt_test_* hooks).  The product library exports no hook and has no switch that could fold shards onto another device
    (VERDICT r04 next-4), so every test that needs N > 1 shards on the 1-GPU box, the huge-page advice alone or the fused
    launch plan takes this fixture; everything else runs on the product."""
    from cute_nucleotides_amd import _lib

    prev = _lib.use_build("hooks")
    L = _lib.lib()
    assert L.cnt_test_alias_devices(0) == 0  # really the hooks build, and the switch is off
    yield L
    L.cnt_test_alias_devices(0)
    _lib.use_build(prev)


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


# ---- full-size runs: what actually executed, at which size ------------------------------------
# The full-size GPU tests (1 GiB, the 16 GiB metric size, 64 GiB) never change size silently: they free
# torch's cached blocks first, then either run at the size BASELINE.json names or call pytest.skip with
# the reason (visible in the summary).  Every run appends {test, log2_nt, ms, ...} to
# gpurun_out/fullsize_tests.jsonl and the terminal summary prints the list, so a GPUTEST record shows
# which sizes ran.
FULLSIZE_RUNS = []
FULLSIZE_LOG = os.path.join(ROOT, "gpurun_out", "fullsize_tests.jsonl")


@pytest.fixture()
def fullsize(request):
    import json

    def record(log2_nt, ms=None, **extra):
        row = {"test": request.node.name, "log2_nt": log2_nt, "ms": None if ms is None else round(ms, 3)}
        row.update(extra)
        FULLSIZE_RUNS.append(row)
        try:
            os.makedirs(os.path.dirname(FULLSIZE_LOG), exist_ok=True)
            with open(FULLSIZE_LOG, "a") as f:
                f.write(json.dumps(row) + "\n")
        except OSError:
            pass
        return row

    return record


def need_free_hbm(gib):
    """Release torch's cached device blocks, then skip VISIBLY unless `gib` GiB of HBM are free."""
    import gc

    import torch

    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < gib * (1 << 30):
        pytest.skip("needs %d GiB of free HBM, %.1f of %.1f GiB are free" % (gib, free / 2**30, total / 2**30))


def pytest_terminal_summary(terminalreporter):
    if FULLSIZE_RUNS:
        terminalreporter.write_line("full-size runs (also in gpurun_out/fullsize_tests.jsonl):")
        for r in FULLSIZE_RUNS:
            extra = " ".join("%s=%s" % (k, v) for k, v in r.items() if k not in ("test", "log2_nt", "ms"))
            terminalreporter.write_line("  %-58s 2^%-2d nt  %s ms  %s" % (r["test"], r["log2_nt"], r["ms"], extra))
    skipped = terminalreporter.stats.get("skipped", [])
    for rep in skipped:
        terminalreporter.write_line("SKIPPED %s: %s" % (rep.nodeid, rep.longrepr[2] if isinstance(rep.longrepr, tuple) else rep.longrepr))
