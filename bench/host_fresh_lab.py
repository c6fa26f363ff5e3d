#!/usr/bin/env python3
"""Host tier into FRESH outputs (the literal `-> Vec<u8>` drop-in: the reference allocates inside the timed call,
benches/bench_n_to_bits.rs:6-7) against reused outputs, under the library's environment knobs -- one child process
per setting (the knobs are read once per process).  (profiles/r03_host_fresh_lab.jsonl and r03_host_pipeline_slots.jsonl
were produced by an earlier version of this script, which also had rows for the fault-in helper team that round 3 built,
measured and removed: CNT_HOST_PREFAULT / CNT_HOST_PREFAULT_THREADS in their `env` columns; r03_host_copy_blocks.jsonl and
r03_host_numa_placement.jsonl by versions with lab-only knobs of the copy pool, see their first lines.)  CNT_LAB_PIN = gpu | other
confines the CHILD (caller, its data and the library's helpers) to the CPUs of the GPU's NUMA node / of another node;
CNT_LAB_DATA_NODE does so only while the inputs are created (first touch); with CNT_LAB_CALLER_NODE the caller is a second
thread confined to that node while the process keeps its full mask.

    python bench/host_fresh_lab.py [--log2-nt 30] [--reps 8]

Rows: ms per call and GiB/s of nucleotides for n_to_bits_hip / bits_to_n_hip with the output (a) reused, (b) a fresh
numpy array per call (np.empty -> untouched pages, freed again after the call), (c) fresh and munmap'ed outside the
timed region (the allocation's share alone)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = [
    ("default (huge-page advice, 3 pipeline slots, 4 copy threads / 8 into fresh pages), caller wherever the scheduler puts it", {}),
    ("default, run 2", {}),
    ("caller and its data on the GPU's NUMA node", {"CNT_LAB_PIN": "gpu"}),
    ("caller and its data on the other node", {"CNT_LAB_PIN": "other"}),
    ("data first touched on the other node, caller unpinned", {"CNT_LAB_DATA_NODE": "other"}),
    ("data on the other node, caller thread on the other node", {"CNT_LAB_DATA_NODE": "other", "CNT_LAB_CALLER_NODE": "other"}),
    ("data on the GPU's node, caller thread on the other node", {"CNT_LAB_DATA_NODE": "gpu", "CNT_LAB_CALLER_NODE": "other"}),
    ("no huge-page advice (numpy still advises its own allocations)", {"CNT_HOST_HUGEPAGE": "0"}),
    ("2 pipeline slots (rounds 1-2)", {"CNT_HOST_SLOTS": "2"}),
    ("4 pipeline slots", {"CNT_HOST_SLOTS": "4"}),
    ("2 copy threads (4 into fresh pages)", {"CNT_HOST_COPY_THREADS": "2"}),
    ("6 copy threads (12 into fresh pages)", {"CNT_HOST_COPY_THREADS": "6"}),
    ("8 copy threads (16 into fresh pages)", {"CNT_HOST_COPY_THREADS": "8"}),
]


def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin(where):
    """restrict this process (before any thread exists) to the CPUs of the GPU's NUMA node, or of another node"""
    import glob

    nodes = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        nodes[int(d.rsplit("node", 1)[1])] = parse_cpulist(open(d + "/cpulist").read())
    gpu_node = None
    try:
        gpu_node = int(open("/sys/bus/pci/devices/%s/numa_node" % os.environ["CNT_LAB_GPU_BDF"]).read())
    except (OSError, KeyError, ValueError):
        pass
    if gpu_node is None or len(nodes) < 2:
        return {"gpu_node": gpu_node, "nodes": len(nodes), "pinned": None}
    target = gpu_node if where == "gpu" else sorted(k for k in nodes if k != gpu_node)[-1]
    os.sched_setaffinity(0, nodes[target] & os.sched_getaffinity(0))
    return {"gpu_node": gpu_node, "nodes": len(nodes), "pinned": target, "cpus": len(os.sched_getaffinity(0))}


def child(log2_nt, reps):
    pinned = pin(os.environ["CNT_LAB_PIN"]) if os.environ.get("CNT_LAB_PIN") else None
    everything = os.sched_getaffinity(0)
    data_pin = pin(os.environ["CNT_LAB_DATA_NODE"]) if os.environ.get("CNT_LAB_DATA_NODE") else None  # only while the inputs are created
    import numpy as np

    from cute_nucleotides_amd import _lib

    L = _lib.lib()
    m = 1 << log2_nt
    words = m // 32
    rng = np.random.default_rng(1)
    n = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, m, dtype=np.uint8)]
    bits = np.empty(words, dtype=np.uint64)
    back = np.empty(m, dtype=np.uint8)
    bits[:] = 0
    back[:] = 0
    if data_pin:
        os.sched_setaffinity(0, everything)  # the data stay where they were first touched; the process may run anywhere again
        pinned = {"data": data_pin}
    if os.environ.get("CNT_LAB_CALLER_NODE"):
        # the CALLER is a second thread confined to one node; the process (its main thread) keeps the full mask, which is
        # what the library looks at when it places its helpers
        import threading

        result = {}

        def body():
            result["caller"] = pin(os.environ["CNT_LAB_CALLER_NODE"])
            result["rows"] = measure(np, L, n, m, words, bits, back, reps)

        t = threading.Thread(target=body)
        t.start()
        t.join()
        rows = result["rows"]
        rows["pin"] = dict(pinned or {}, caller=result["caller"])
        print(json.dumps(rows))
        return
    rows = measure(np, L, n, m, words, bits, back, reps)
    rows["pin"] = pinned
    print(json.dumps(rows))


def measure(np, L, n, m, words, bits, back, reps):
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert L.cnt_n_to_bits(p(n), m, p(bits), words) == 0 and L.cnt_bits_to_n(p(bits), words, m, p(back)) == 0
    assert np.array_equal(back, n)
    keep = []

    def timed(fn, reps):
        fn()
        keep.clear()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    def enc_fresh():
        out = np.empty(words, dtype=np.uint64)
        assert L.cnt_n_to_bits(p(n), m, p(out), words) == 0

    def dec_fresh():
        out = np.empty(m, dtype=np.uint8)
        assert L.cnt_bits_to_n(p(bits), words, m, p(out)) == 0

    def enc_fresh_kept():
        out = np.empty(words, dtype=np.uint64)
        assert L.cnt_n_to_bits(p(n), m, p(out), words) == 0
        keep.append(out)

    def dec_fresh_kept():
        out = np.empty(m, dtype=np.uint8)
        assert L.cnt_bits_to_n(p(bits), words, m, p(out)) == 0
        keep.append(out)

    rows = {}
    for name, fn, r in (("n_to_bits_hip reused", lambda: L.cnt_n_to_bits(p(n), m, p(bits), words), reps),
                        ("bits_to_n_hip reused", lambda: L.cnt_bits_to_n(p(bits), words, m, p(back)), reps),
                        ("n_to_bits_hip fresh", enc_fresh, reps), ("bits_to_n_hip fresh", dec_fresh, reps),
                        ("n_to_bits_hip fresh, freed later", enc_fresh_kept, min(reps, 4)),
                        ("bits_to_n_hip fresh, freed later", dec_fresh_kept, min(reps, 4))):
        dt = timed(fn, r)
        rows[name] = {"ms": round(dt * 1e3, 3), "GiBs": round(m / dt / 2**30, 2)}
    keep.clear()
    out = np.empty(m, dtype=np.uint8)
    assert L.cnt_bits_to_n(p(bits), words, m, p(out)) == 0 and np.array_equal(out, n)
    rows["fresh_over_reused"] = {"n_to_bits_hip": round(rows["n_to_bits_hip fresh"]["ms"] / rows["n_to_bits_hip reused"]["ms"], 3),
                                 "bits_to_n_hip": round(rows["bits_to_n_hip fresh"]["ms"] / rows["bits_to_n_hip reused"]["ms"], 3)}
    try:
        cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else ctypes.CDLL(None).sched_getcpu()
        node = [d for d in os.listdir("/sys/devices/system/cpu/cpu%d" % cpu) if d.startswith("node")]
        rows["caller_cpu_at_end"] = {"cpu": cpu, "node": node[0] if node else None}
    except OSError:
        pass
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-nt", type=int, default=30)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--settings", default="", help="comma list of indices into SETTINGS (default: all)")
    ap.add_argument("--repeat", type=int, default=1, help="run the picked settings this many times (how often does a process land badly?)")
    a = ap.parse_args()
    if a.child:
        child(a.log2_nt, a.reps)
        sys.exit(0)
    picked = [SETTINGS[int(i)] for i in a.settings.split(",") if i] or SETTINGS
    bdf = subprocess.run([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))"],
                         capture_output=True, text=True).stdout.strip()
    os.environ["CNT_LAB_GPU_BDF"] = bdf
    print(json.dumps({"gpu_bdf": bdf}), flush=True)
    for name, env in picked * a.repeat:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--log2-nt", str(a.log2_nt), "--reps", str(a.reps)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"setting": name, "env": env, "log2_nt": a.log2_nt, "cpus": len(os.sched_getaffinity(0)), "rows": json.loads(line[-1]) if line else None,
                          "error": None if line else r.stderr[-400:]}), flush=True)
