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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log2-nt", "28",
                          "--cpu-seconds", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["verified"] is True
    assert j["unit"] == "Gnt/s" and j["higher_is_better"] is True and j["dtype"] == "u8"
    assert j["config"]["nt_per_step"] == 2 * (1 << 28)
    for key in ("roofline", "roofline_decode"):
        r = j[key]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["algorithmic_bytes_per_launch"] == int(1.25 * (1 << 28))
    assert abs(j["value"] - j["config"]["nt_per_step"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"]) / 1e9) < 0.01 * j["value"]


def test_two_rank_launch_path_shares_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CNT_BENCH_SHARE_GPU="1")  # default control-plane backend (gloo)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--log2-nt", "28", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["verified"] is True
    assert j["config"]["nt_per_gpu"] == 1 << 28 and j["config"]["nt_per_step"] == 2 * 2 * (1 << 28)
    assert "cpu_baseline" not in j  # rank 0 at N=1 only
