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


def _bench_unaided(tmp_path, *argv, timeout=1100):
    """bench.py exactly as the driver may call it: no torchrun, no RANK / WORLD_SIZE in the environment"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(DDK_BENCH_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", DDK_BENCH_TIMEOUT="1000")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` unaided: two rank processes (gloo, both on cuda:0 -- the only two-rank world a one-GPU box hosts),
    C1 broadcast, per-rank figures gathered over the process group, value = both ranks' images over the slower rank's time."""
    line = _bench_unaided(tmp_path, "--gpus", "2", "--steps", "8", "--warmup", "2", "--no-full-chain")
    assert line["n_gpus"] == 2 and line["dist"]["world"] == 2 and line["dist"]["backend"] == "gloo"
    assert line["rccl_world"] is None                        # gloo rehearsal: no RCCL world is claimed
    ranks = line["dist"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["ms_per_step"] > 0 for r in ranks)
    assert line["config"]["global_batch"] == 64 and line["config"]["weights_broadcast_bytes"] > 90e6
    assert line["config"]["in_launch_groupnorm_per_rank"] == [0, 0]       # shared GPU: the in-launch exchange is switched off up front
    assert abs(line["ms_per_step"] - max(r["ms_per_step"] for r in ranks)) < 1e-6
    assert "cpu_baseline" not in line and line["value"] > 1


def test_bench_gpus_4_launches_four_ranks_itself(tmp_path):
    """the same with FOUR ranks sharing cuda:0 -- the widest world the one-GPU box admits next to the test process (its guard allows six
    GPU processes per user; the N = 8 launch itself is rehearsed with probe children in tests/test_host_logic.py): four rank records,
    one JSON line, `world == 4`"""
    line = _bench_unaided(tmp_path, "--gpus", "4", "--steps", "4", "--warmup", "1", "--no-full-chain")
    assert line["n_gpus"] == 4 and line["dist"]["world"] == 4 and line["dist"]["backend"] == "gloo"
    ranks = line["dist"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and all(r["ms_per_step"] > 0 for r in ranks)
    assert line["config"]["global_batch"] == 128 and line["config"]["in_launch_groupnorm_per_rank"] == [0, 0, 0, 0]
    assert line["value"] > 1


def test_bench_train_dp_two_ranks(tmp_path):
    """`python bench.py --gpus 2 --train-dp` unaided: the cfg5 optimiser step (2 micro-batches of 8 images of 256x256 per rank) through
    trainers.setup_trainer / TrainerDDPM.optimizer_step with the C2 all-reduce of the 89 MB gradient bucket between two ranks."""
    line = _bench_unaided(tmp_path, "--gpus", "2", "--train-dp", "--steps", "2", "--warmup", "2")
    assert line["n_gpus"] == 2 and line["dist"]["world"] == 2 and len(line["dist"]["ranks"]) == 2
    c = line["config"]
    assert c["batch_per_gpu"] == 8 and c["global_batch"] == 16 and c["accumulate"] == 2 and c["graph_train"]
    assert c["grad_bucket_bytes"] == 4 * c["n_params"] and c["n_params"] == 22254723        # the UNet at C_in = 3 (a plain DDPM has no other parameters)
    assert c["allreduce_alone_ms"] > 0 and line["ms_per_step"] >= c["ms_per_step_no_allreduce"] * 0.9
    assert line["value"] == pytest.approx(2 * 8 * 2 / (line["ms_per_step"] / 1e3), rel=1e-6)
