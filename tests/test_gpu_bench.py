"""bench.py contract checks on the GPU box: the default single-process line, and the N>1 launch
path (`python -m torch.distributed.run ... bench.py --gpus 2`) exercised with two ranks sharing
cuda:0 over gloo (RCCL refuses two ranks on one device; the driver's real multi-GPU run uses
nccl = RCCL with one GPU per rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), text  # stdout is ONE JSON line and nothing else
    return json.loads(lines[0])


def test_single_gpu_line_small():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log2-nt", "30",
                          "--shard-log2-nt", "31", "--cpu-seconds", "0", "--no-live-traffic"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["verified"] is True
    assert j["unit"] == "Gnt/s" and j["higher_is_better"] is True and j["dtype"] == "u8"
    assert j["config"]["nt_per_step"] == 2 * (1 << 30)
    for key in ("roofline", "roofline_decode"):
        r = j[key]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["algorithmic_bytes_per_launch"] == int(1.25 * (1 << 30))
        k = r["kernel_ms"]  # mean, median and min of the per-launch HIP-event times
        assert k["n"] == 3 and k["min"] <= k["median"] <= k["max"] and abs(k["mean"] - r["avg_kernel_ms"]) < 1e-3
        assert r["frac_at_min"] >= r["frac_at_median"] > 0
    # same-run ceilings from the shipped-shape probes, and both views of the encode kernel against them
    c = j["ceilings"]["rank0"]
    for name in ("read_only", "write_only", "copy_1to1", "read4_write1_encode_shape", "read1_write4_decode_shape"):
        assert 1000.0 < c[name]["GBs"] < 8000.0, (name, c[name])
    ev = j["ceilings"]["encode_vs"]
    assert abs(ev["of_spec_8000"] - j["roofline"]["frac"]) < 1e-3 and 0.5 < ev["of_read4_write1_ceiling"] < 1.3
    assert abs(ev["read_only_view_of_spec_8000"] - j["roofline"]["read_only_view"]["frac"]) < 1e-3
    fv = j["ceilings"]["fused_vs"]  # the fused kernel against the arithmetic-free 1:1 copy of the same run
    assert abs(fv["of_spec_8000"] - j["fused_round_trip"]["frac"]) < 1e-3 and 0.6 < fv["of_copy_1to1_ceiling"] < 1.3
    # the 1 GiB configs, measured on the same buffers; the fused pass verified against the two-pass outputs
    c1 = j["configs"]["configs[1] n_to_bits encode, 1 GiB (2^30 nt)"]
    assert c1["frac"] > 0.3 and c1["timing"].startswith("10 launches queued") and c1["isolated_single_launch"]["frac"] > 0.3
    assert j["configs"]["configs[2] bits_to_n decode, 1 GiB (2^30 nt)"]["round_trip_verified"] is True
    assert j["fused_round_trip"]["ms_stats"]["verified"] is True
    # ... and the same call with all three pointers off the 128-B grid, one launch too: within 10 % of the aligned one at 2^30 nt
    og = j["fused_round_trip"]["ms_stats"]["off_grid"]
    assert og["verified"] is True and og["nt"] == (1 << 30) - 4096 and 0.9 < j["fused_round_trip"]["off_grid_vs_aligned"] < 1.1
    rag = j["configs"]["ragged: 2^30 - 19 nt (13 nt in the last word, zero-padded)"]
    assert rag["round_trip_verified"] is True and rag["encode_frac"] > 0.3 and rag["decode_frac"] > 0.3
    # the ragged size is ONE launch per call now: within a few percent of the aligned size in the same run
    assert rag["launches_per_call"] == 1 and 0.9 < rag["encode_vs_aligned_2p30"] < 1.05 and 0.9 < rag["decode_vs_aligned_2p30"] < 1.05
    # SURVEY 8 f-1 / f-4 on the driver line: 5-letter codec and packed-domain ops at the line's size, verified in-run
    c5, po = j["codec5"], j["packed_ops"]
    assert c5["verified"] is True and c5["nt"] == 1 << 30 and abs(c5["algorithmic_bytes_per_nt"] - (1 + 8 / 27)) < 1e-3
    for d in ("encode", "decode"):
        assert 0.2 < c5[d]["frac"] < 1.0 and c5[d]["ms"]["n"] == 4 and abs(c5[d]["frac"] - c5[d]["achieved_GBs"] / 8000.0) < 1e-3
    assert po["verified"] is True and po["parity"].startswith("unpinned")
    for op, bpn in (("hamming", 0.5), ("complement", 0.5), ("reverse_complement", 0.5), ("validate", 1.0)):
        assert po[op]["bytes_per_nt"] == bpn and 0.2 < po[op]["frac"] < 1.0, (op, po[op])
    # configs[4]'s per-GPU shard (reduced to 2^31 nt here), and this rank's device identity
    sh = j["configs4_sharded_encode"]
    assert sh["nt_per_gpu"] == 1 << 31 and sh["ranks_measured"] == 1 and sh["per_gpu_gnts"]["min"] > 1000
    (r0,) = j["ranks"]
    assert r0["rank"] == 0 and r0["device_index"] == 0 and r0["visible_devices"] >= 1
    assert len(r0["pci_bus_id"].split(":")) == 3 and r0["configs4_shard"]["round_trip_verified"] is True
    assert j["devices"]["distinct"] == 1 and j["devices"]["data_path_collective"] is None
    assert j["roofline"]["traffic"] is None or "static" in j["roofline"]["traffic_source"]
    assert abs(j["value"] - j["config"]["nt_per_step"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"]) / 1e9) < 0.01 * j["value"]


@pytest.mark.parametrize("n", [4, 8])
def test_both_bench_forms_folded_onto_one_gpu(n):
    """the two N > 1 forms of bench.py -- one rank per GPU under torch.distributed.run (the driver's) and
    ONE process over N devices through the enqueue-only sharded tier -- with N = 4 and 8 shards folded onto this box's GPU:
    same JSON contract, N rows, verified, and for the single-process form the fan-out's cost on the line
    (`scaling_overhead_us`: wall per step over N shards minus N x the wall per step of shard 0 alone; <= 1 % of a 3.1-ms
    kernel) with the host's enqueue time per step beside it."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CNT_BENCH_SHARE_GPU="1")
    # every N > 1 line is a full contract line -- `cpu_baseline` (the CPU leg runs in one of the two forms per
    # N here, 1 s of it: the single-process form at N = 4, the rank form at N = 8) and a non-null, labelled `roofline.traffic`
    cpu_single, cpu_ranks = ("1", "0") if n == 4 else ("0", "1")
    common = ["--gpus", str(n), "--steps", "20", "--warmup", "2", "--log2-nt", "26", "--shard-log2-nt", "27", "--cpu-seconds", cpu_ranks]
    # (a) one process, N shards, everything queued, one wait (100 steps: the overhead figure multiplies a one-shard wall time by N)
    single = ["--gpus", str(n), "--steps", "100", "--warmup", "5", "--log2-nt", "26", "--shard-log2-nt", "27", "--cpu-seconds", cpu_single]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + single, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == n and j["verified"] is True and len(j["ranks"]) == n and j["devices"]["processes"] == 1
    assert "enqueue-only" in j["config"]["launch"] and "ONE cnt_sharded_dev_wait" in j["config"]["launch"] and "cnt_sharded_dev_wait_event" in j["config"]["launch"]
    _full_contract_line(j, n, cpu_single == "1")
    d = j["devices"]  # first contact (next-5): N shards folded onto this box's one device -- printed, and labelled as folded
    assert d["distinct"] == 1 and d["expected_distinct"] == n and d["shared_gpu_test_hook"] is True and d["library_build"] == "hooks" and d["visible"] == 1
    assert all(r["chip"] == j["ranks"][0]["chip"] and r["chip"]["compute_units"] > 0 and r["chip"]["xcds"] >= 1 and r["hbm_before"]["free_GiB"] > 1 for r in j["ranks"])
    f = j["fused_round_trip"]  # the fused pass of every shard through the same queue, verified against the two-pass outputs
    assert f["ms_stats"]["verified"] is True and 0 < f["frac_over_ranks"]["min"] <= f["frac_over_ranks"]["max"] < 1.2 and f["bytes_per_nt"] == 2.25
    assert 1000.0 < j["ceilings"]["rank0"]["read_only"]["GBs"] < 8000.0 and j["ceilings"]["encode_vs"]["of_read4_write1_ceiling"] > 0.5 / n  # N folded shards share this box's bandwidth
    so = j["scaling_overhead"]
    assert so["shards_per_device"] == n and so["per_step_us"] == j["scaling_overhead_us"]
    assert so["per_step_us"] <= 31.0, so  # 1 % of a 3.1-ms kernel; folded shards overlap, so the number is usually negative
    assert 0 < so["host_enqueue_us_per_step"] < 40.0 * n, so  # a few microseconds per launch, 2 n launches per step
    assert all(r["encode_ms"]["n"] == 100 and r["encode_ms"]["p90"] >= r["encode_ms"]["p10"] > 0 for r in j["ranks"])
    assert [r["first_nt"] for r in j["ranks"]] == [k << 26 for k in range(n)]
    assert abs(j["value"] - j["config"]["nt_per_step"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"]) / 1e9) < 0.01 * j["value"]
    # (b) the driver's form: N ranks (sharing cuda:0 over gloo here), barrier + max over ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + common
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == n and j["verified"] is True and [r["rank"] for r in j["ranks"]] == list(range(n))
    assert len({r["pid"] for r in j["ranks"]}) == n and j["devices"]["control_plane"] == "gloo"
    assert j["config"]["nt_per_step"] == 2 * n * (1 << 26) and j["configs4_sharded_encode"]["ranks_measured"] == n
    _full_contract_line(j, n, cpu_ranks == "1")
    assert j["devices"]["distinct"] == 1 and j["devices"]["expected_distinct"] == n and j["devices"]["shared_gpu_test_hook"] is True
    assert all(r["chip"]["compute_units"] > 0 and r["hbm_before"]["free_GiB"] > 1 for r in j["ranks"])


def _full_contract_line(j, n, with_cpu):
    """what a SCALE line must carry at N > 1 (VERDICT r04 missing-1): the contract keys, both rooflines with a non-null traffic and
    a source that says how it was obtained, and -- when the CPU leg ran -- cpu_baseline with value, unit, cores, kind, sample"""
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in j, key
    assert j["scaling"] == "weak" and j["dtype"] == "u8" and j["vs_baseline"] is None and "workload" in j["config"]
    for key in ("roofline", "roofline_decode"):
        r = j[key]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is not None and r["traffic"] > 0, (key, j.get("traffic_live"))
        assert r["traffic_source"].startswith(("measured by this run", "static")), r["traffic_source"]
        if r["traffic_source"].startswith("measured"):  # 2^26-nt launches: the calibrated counters land within a few percent
            assert "rank 0's device" in r["traffic_source"] and abs(r["traffic"] / r["algorithmic_bytes_per_launch"] - 1.0) < 0.05, (key, r["traffic"])
    if with_cpu:
        c = j["cpu_baseline"]
        assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["unit"] == "Gnt/s" and c["sample"] and "after the timed region" in c["when"]
        assert c["encode_gnts"] > 0 and c["decode_gnts"] > 0 and c["one_thread"]["value"] > 0
    else:
        assert "cpu_baseline" not in j  # --cpu-seconds 0


def test_two_rank_launch_path_shares_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CNT_BENCH_SHARE_GPU="1")  # default control-plane backend (gloo)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--log2-nt", "28", "--shard-log2-nt", "29", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["verified"] is True
    assert j["config"]["nt_per_gpu"] == 1 << 28 and j["config"]["nt_per_step"] == 2 * 2 * (1 << 28)
    assert "cpu_baseline" not in j  # --cpu-seconds 0 (with a budget the leg runs at every N: test_both_bench_forms_folded_onto_one_gpu)
    # one row per rank through the control plane: who ran where, each rank's own kernel times and fractions
    rows = j["ranks"]
    assert [r["rank"] for r in rows] == [0, 1] and rows[0]["pid"] != rows[1]["pid"]
    assert rows[0]["first_nt"] == 0 and rows[1]["first_nt"] == 1 << 28 and all(r["nt"] == 1 << 28 for r in rows)
    for r in rows:
        assert r["device_index"] == 0 and r["visible_devices"] >= 1 and r["pci_bus_id"]  # both on cuda:0 under the test hook
        assert r["encode_ms"]["n"] == 2 and 0 < r["encode_frac"] < 1 and 0 < r["decode_frac"] < 1
        assert r["configs4_shard"]["nt"] == 1 << 29 and r["configs4_shard"]["round_trip_verified"] is True
    assert rows[1]["configs4_shard"]["first_nt"] == 1 << 29
    d = j["devices"]
    assert (d["distinct"], d["expected_distinct"], d["shared_gpu_test_hook"], d["data_path_collective"], d["control_plane"], d["processes"]) == (1, 2, True, None, "gloo", 2)
    assert d["not_an_spx_mi355x"] == [] or all(x["chip"] for x in d["not_an_spx_mi355x"])  # an SPX MI355X here; anything else is named
    o = j["roofline_over_ranks"]
    assert o["encode_frac"]["min"] == min(r["encode_frac"] for r in rows) and o["encode_frac"]["max"] == max(r["encode_frac"] for r in rows)
    assert o["encode_read_view_frac"]["min"] <= o["encode_read_view_frac"]["max"]
    sh = j["configs4_sharded_encode"]
    assert sh["ranks_measured"] == 2 and sh["nt_per_gpu"] == 1 << 29 and sh["total_GiB"] == 1.0 and sh["aggregate_gnts"] > 0


def test_gpus_n_without_a_launcher_is_one_process_over_n_devices():
    """`python bench.py --gpus N` launched plainly used to exit ("needs a torch.distributed.run
    launch").  It now drives the N devices from ONE process through the enqueue-only sharded tier (cnt_*_sharded_dev_enqueue,
    one cnt_sharded_dev_wait for all steps) and prints the same JSON line.  On the 1-GPU box the N shards are folded onto cuda:0 (CNT_BENCH_SHARE_GPU=1 ->
    cnt_test_alias_devices); without that the shortage of devices is a clear error, not a wrong run."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2-nt", "28", "--shard-log2-nt", "29"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, CNT_BENCH_SHARE_GPU="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["verified"] is True and j["steps"] == 3
    assert j["config"]["nt_per_gpu"] == 1 << 28 and j["config"]["nt_per_step"] == 2 * 2 * (1 << 28) and "single process" in j["config"]["launch"]
    assert j["devices"]["processes"] == 1 and j["devices"]["data_path_collective"] is None and j["devices"]["control_plane"] is None
    assert j["scaling_overhead"]["shards_per_device"] == 2 and j["scaling_overhead"]["wall_ms_per_step_shard0_alone"] > 0
    rows = j["ranks"]
    assert [r["rank"] for r in rows] == [0, 1] and rows[0]["pid"] == rows[1]["pid"]
    assert rows[0]["first_nt"] == 0 and rows[1]["first_nt"] == 1 << 28 and all(r["nt"] == 1 << 28 for r in rows)
    for r in rows:
        assert r["encode_ms"]["n"] == 3 and 0 < r["encode_frac"] < 1 and 0 < r["decode_frac"] < 1 and r["pci_bus_id"]
        assert r["configs4_shard"]["nt"] == 1 << 29 and r["configs4_shard"]["round_trip_verified"] is True
    assert rows[1]["configs4_shard"]["first_nt"] == 1 << 29
    sh = j["configs4_sharded_encode"]
    assert sh["ranks_measured"] == 2 and sh["nt_per_gpu"] == 1 << 29 and sh["aggregate_gnts"] > 0
    assert abs(j["value"] - j["config"]["nt_per_step"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"]) / 1e9) < 0.01 * j["value"]
    for key in ("roofline", "roofline_decode"):
        assert j[key]["bound"] == "hbm" and abs(j[key]["frac"] - j[key]["achieved"] / j[key]["peak"]) < 1e-3
    # more devices than exist and no test hook: a clear refusal
    import torch

    too_many = str(torch.cuda.device_count() + 1)
    env = {k: v for k, v in os.environ.items() if k != "CNT_BENCH_SHARE_GPU"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", too_many, "--steps", "1", "--warmup", "0", "--log2-nt", "24"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "visible" in out.stderr and out.stdout.strip() == ""


def test_criterion_twin_runs_and_self_checks():
    """bench/bench_n_to_bits (the executable C++ twin of the reference's criterion harness,
    benches/bench_n_to_bits.rs:9-82; the std-only Rust original is rust/benches/bench_n_to_bits.rs) runs on the
    GPU box: rc 0, one `*_hip` row in each of the reference's four groups + the memcpy comparator, the
    device-resident rows, and its built-in C-ABI self-checks (every row's output compared with its input or with
    the reference's 0xD8... vector)."""
    import re

    exe = os.path.join(ROOT, "bench", "bench_n_to_bits")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe, "20"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"^(\S+)\s+(.+?)\s+time:\s+([0-9.]+) us\s+thrpt:\s+([0-9.]+) GiB/s", line)
        if m:
            rows[(m.group(1), m.group(2).strip())] = (float(m.group(3)), float(m.group(4)))
    for key in (("n_to_bits", "n_to_bits_hip"), ("n_to_bits", "memcpy"), ("bits_to_n", "bits_to_n_hip"),
                ("n_to_bits2", "n_to_bits2_hip"), ("bits_to_n2", "bits_to_n2_hip"),
                ("n_to_bits", "n_to_bits_hip_dev (resident)"), ("bits_to_n", "bits_to_n_hip_dev (resident)"),
                ("host-tier", "n_to_bits_hip/2^20"), ("host-tier", "cnt_bits_to_n/2^20 (reused out)"),
                ("device-tier", "n_to_bits_hip_dev/2^20 (resident)"), ("device-tier", "bits_to_n_hip_dev/2^20 (resident)"),
                ("n_to_bits", "n_to_bits_hip_into"), ("bits_to_n", "bits_to_n_hip_into"), ("host-tier", "n_to_bits_hip_into/2^20"),
                ("host-tier", "bits_to_n_hip_into/2^20"), ("queue", "3 x (encode + decode)/2^22, one wait"),
                ("queue", "adopted stream + events/2^22")):  # round 5: the queue ordered against the caller's streams through the C++ mirror
        assert key in rows, (key, sorted(rows))
        us, gib = rows[key]
        assert us > 0 and gib > 0
    assert rows[("n_to_bits", "n_to_bits_hip")][0] < 200.0  # a 40 000-nt host-slice call is tens of microseconds, not ms
    assert rows[("device-tier", "n_to_bits_hip_dev/2^20 (resident)")][1] > 20.0  # GiB/s of ASCII, launch-latency-bound at 1 MiB
    assert "self-check ok" in out.stdout
    # the Rust original names the same groups and functions (kept in step with the twin by this check)
    rust = open(os.path.join(ROOT, "rust", "benches", "bench_n_to_bits.rs")).read()
    for name in ("n_to_bits_lut", "n_to_bits_pext", "n_to_bits_shift", "n_to_bits_movemask", "n_to_bits_mul", "memcpy", "n_to_bits_hip",
                 "bits_to_n_lut", "bits_to_n_shuffle", "bits_to_n_pdep", "bits_to_n_clmul", "bits_to_n_hip",
                 "n_to_bits2_lut", "n_to_bits2_pext", "n_to_bits2_hip", "bits_to_n2_lut", "bits_to_n2_pdep", "bits_to_n2_hip"):
        assert 'bench_function("%s"' % name in rust, name
    assert "std::time" in rust and "use criterion" not in rust


def test_cpu_baseline_and_host_tier_blocks():
    """the N = 1 line's CPU leg: the port timed on this box's cores (threads, logical and physical cores stated
    separately), the reference-faithful allocation-inclusive rows incl. the 5-letter functions and the 1 MiB / 1 GiB
    sizes, and the host-tier crossover table against one CPU thread"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log2-nt", "28",
                          "--shard-log2-nt", "28", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Gnt/s" and c["value"] > 1.0
    assert c["reference_toolchain"]["cargo"] is None  # if this ever fails the reference can be built: switch kind to "reference"
    d = c["cores_detail"]
    assert c["cores"] == d["threads_timed"] == d["logical_cpus"] and 1 <= d["physical_cores"] <= d["logical_cpus"] and d["sockets"] >= 1
    f = c["reference_faithful_40k_GiBs"]
    for name in ("n_to_bits_lut", "n_to_bits_pext", "n_to_bits_shift", "n_to_bits_movemask", "n_to_bits_mul", "memcpy", "bits_to_n_lut",
                 "bits_to_n_shuffle", "bits_to_n_pdep", "bits_to_n_clmul", "n_to_bits2_lut", "n_to_bits2_pext", "bits_to_n2_lut", "bits_to_n2_pdep"):
        assert f[name] > 0, name  # every row of the reference's criterion harness (bench_n_to_bits.rs:15-20,31-32,44-47,59-60)
    big = c["reference_faithful_GiBs_1thread_alloc_inclusive"]
    assert len(big["2^20"]) == 10 and len(big["2^30"]) == 10 and big["2^30"]["n_to_bits_movemask"] > 0
    h = j["host_tier"]
    assert h["log2_nt"] == [12, 14, 16, 18, 20, 22, 24, 26, 28, 30]
    assert all(len(v) == 10 and min(v) > 1.0 for v in h["us_per_call"].values()) and len(h["us_per_call"]) == 6  # reused, pinned, fresh x 2 directions
    x = h["crossover_vs_one_cpu_thread"]["n_to_bits_hip vs n_to_bits_movemask"]
    assert x["host_tier_ahead_from"] is None or x["host_tier_ahead_from"] in x["table_GiBs"]
    assert len(x["table_GiBs"]) == 10
    # same-run PCIe ceilings (pinned hipMemcpy) and the host tier's fraction of them at 1 GiB, reused and fresh outputs
    assert 10.0 < h["pcie_ceiling"]["h2d_GiBs"] < 70.0 and 10.0 < h["pcie_ceiling"]["d2h_GiBs"] < 70.0
    fr = h["frac_of_pcie_ceiling_at_2^30"]
    assert len(fr) == 6 and all(0.02 < v < 1.2 for v in fr.values()), fr
    assert fr["n_to_bits_hip pinned in + out"] > 0.9 * fr["n_to_bits_hip reused out"], fr  # no staging copies: never materially slower
    fo = h["fresh_over_reused_at_2^30"]
    for fn in ("n_to_bits_hip", "bits_to_n_hip"):
        assert 0.8 < fo[fn]["drop_outside"] <= fo[fn]["drop_inside"] * 1.25 and fo[fn]["drop_inside"] < 8.0, fo
        assert 0.8 < fo[fn]["into"] < 1.3, fo  # the `_into` form IS the call into a reused output, through the mirror
    tc = h["timing_conventions"]
    assert set(tc) == {"what", "2^26", "2^28", "2^30"}
    for fn in ("n_to_bits_hip", "bits_to_n_hip"):  # all three conventions on the line
        row = tc["2^30"][fn]
        assert row["into_us"] > 0 and row["drop_outside_us"] > 0 and row["drop_inside_us"] >= 0.8 * row["into_us"], row
    # roofline.traffic: HBM bytes per launch measured by THIS run (two rocprofv3 --pmc child passes, calibrated on
    # known-size probes) -- equal to the algorithmic bytes to well under 1 %: nothing is re-read
    for key in ("roofline", "roofline_decode"):
        r = j[key]
        assert r["traffic_source"].startswith("measured by this run"), j.get("traffic_live")
        assert abs(r["traffic"] / r["algorithmic_bytes_per_launch"] - 1.0) < 0.01, (key, r["traffic"], r["algorithmic_bytes_per_launch"])
    cal = j["traffic_live"]["calibration"]
    assert 1.5 < cal["fetch_scale"] < 2.5 and 0.8 < cal["write_scale"] < 1.25
