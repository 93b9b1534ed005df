"""RCCL has to have run once before an 8-GPU node sees this code (the reference has no distributed code,
F8): ONE fresh child process per test -- launched before any GPU call in it -- builds an `nccl` process group of world size 1 on
cuda:0 and (a) pushes the cfg4 weight bucket (C1, ~90.7 MB) and the flat gradient bucket (C2) through parallel/dist.py, (b) runs
bench.py with its N > 1 branches forced (process group, weight broadcast, barriers, max-over-ranks).  One process at a time."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)


def test_rccl_world1_broadcast_and_all_reduce(tmp_path):
    out = os.path.join(tmp_path, "rccl.json")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py"), out], env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    with open(out) as f:
        r = json.load(f)
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["broadcast_identity"] and r["all_reduce_identity"]
    assert r["broadcast_bytes"] > 90e6                       # UNet 89.0 MB + resamplers + schedule buffers
    assert r["grad_bucket_bytes"] == 4 * r["n_params"] and r["n_params"] == 22671699      # dDDPM x3


def test_bench_distributed_branch_over_rccl(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-train",
                        "--no-cpu-baseline"], env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", DDK_BENCH_FORCE_DIST="1"),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["weights_broadcast_bytes"] > 90e6
    assert line["value"] > 1 and line["roofline"]["frac"] <= 1.0
