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
                          "--shard-log2-nt", "31", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
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
    # the 1 GiB configs, measured on the same buffers; the fused pass verified against the two-pass outputs
    assert j["configs"]["configs[1] n_to_bits encode, 1 GiB (2^30 nt)"]["frac"] > 0.3
    assert j["configs"]["configs[2] bits_to_n decode, 1 GiB (2^30 nt)"]["round_trip_verified"] is True
    assert j["fused_round_trip"]["ms_stats"]["verified"] is True
    # configs[4]'s per-GPU shard (reduced to 2^31 nt here), and this rank's device identity
    sh = j["configs4_sharded_encode"]
    assert sh["nt_per_gpu"] == 1 << 31 and sh["ranks_measured"] == 1 and sh["per_gpu_gnts"]["min"] > 1000
    (r0,) = j["ranks"]
    assert r0["rank"] == 0 and r0["device_index"] == 0 and r0["visible_devices"] >= 1
    assert len(r0["pci_bus_id"].split(":")) == 3 and r0["configs4_shard"]["round_trip_verified"] is True
    assert j["devices"]["distinct"] == 1 and j["devices"]["data_path_collective"] is None
    assert j["roofline"]["traffic"] is None or "static" in j["roofline"]["traffic_source"]
    assert abs(j["value"] - j["config"]["nt_per_step"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"]) / 1e9) < 0.01 * j["value"]


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
    assert "cpu_baseline" not in j  # rank 0 at N=1 only
    # one row per rank through the control plane: who ran where, each rank's own kernel times and fractions
    rows = j["ranks"]
    assert [r["rank"] for r in rows] == [0, 1] and rows[0]["pid"] != rows[1]["pid"]
    assert rows[0]["first_nt"] == 0 and rows[1]["first_nt"] == 1 << 28 and all(r["nt"] == 1 << 28 for r in rows)
    for r in rows:
        assert r["device_index"] == 0 and r["visible_devices"] >= 1 and r["pci_bus_id"]  # both on cuda:0 under the test hook
        assert r["encode_ms"]["n"] == 2 and 0 < r["encode_frac"] < 1 and 0 < r["decode_frac"] < 1
        assert r["configs4_shard"]["nt"] == 1 << 29 and r["configs4_shard"]["round_trip_verified"] is True
    assert rows[1]["configs4_shard"]["first_nt"] == 1 << 29
    assert j["devices"] == {"distinct": 1, "shared_gpu_test_hook": True, "data_path_collective": None, "control_plane": "gloo"}
    o = j["roofline_over_ranks"]
    assert o["encode_frac"]["min"] == min(r["encode_frac"] for r in rows) and o["encode_frac"]["max"] == max(r["encode_frac"] for r in rows)
    assert o["encode_read_view_frac"]["min"] <= o["encode_read_view_frac"]["max"]
    sh = j["configs4_sharded_encode"]
    assert sh["ranks_measured"] == 2 and sh["nt_per_gpu"] == 1 << 29 and sh["total_GiB"] == 1.0 and sh["aggregate_gnts"] > 0
