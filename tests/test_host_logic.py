"""CPU-side checks of the product package: state_dict layout, schedule buffers, C-ABI symbols, loud failure
without a device, CLI surface.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import ddpm_cfg, dddpm_cfg, golden, golden_keys
from conftest import ROOT

import models
from models import DDPM, DownsampleDDPM, DownsampleDDPMAutoencoder, Unet
from ddk import lib as L


def _shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def test_state_dict_layout_matches_reference():
    ks = golden_keys()
    cfg = ddpm_cfg(128, 3, 32)
    got = _shapes(DDPM(cfg, Unet(cfg), "cpu", 3))
    assert got == ks["ddpm_c3"] and list(got) == list(ks["ddpm_c3"])          # same keys, shapes AND order
    cfgd = dddpm_cfg(128, 256, 3)
    got = _shapes(DownsampleDDPM(cfgd, Unet(cfgd), "cpu", 3))
    assert got == ks["dddpm_x3"] and list(got) == list(ks["dddpm_x3"])
    cfgt = dddpm_cfg(32, 32, 2)
    got = _shapes(DownsampleDDPMAutoencoder(cfgt, Unet(cfgt), "cpu", 3))
    assert got == ks["dddpm_tiny_x2"] and list(got) == list(ks["dddpm_tiny_x2"])


def test_param_counts():
    cfg = dddpm_cfg(128, 256, 3)
    u = Unet(cfg)
    assert sum(p.numel() for p in u.parameters()) == 22261768
    assert sum(p.numel() for p in DownsampleDDPM(cfg, u, "cpu", 3).parameters()) == 22671699
    cfg3 = ddpm_cfg(128, 3, 32)
    assert sum(p.numel() for p in Unet(cfg3).parameters()) == 22254723


@pytest.mark.parametrize("kind,T", [("linear", 1000), ("linear", 200), ("cosine", 1000)])
def test_schedule_buffers_bit_exact(kind, T):
    g = golden("g1_schedule")
    m = DDPM(ddpm_cfg(32, 3, 16, T, kind), torch.nn.Identity(), "cpu", 3)
    sd = m.state_dict()
    from utils.synthetic import SCHEDULE_KEYS
    for k in SCHEDULE_KEYS:
        assert np.array_equal(sd[k].numpy(), g[f"{kind}_{T}_{k}"]), k
    assert np.array_equal(m.vlb_weights.numpy(), g[f"{kind}_{T}_vlb_weights"])
    assert "vlb_weights" not in sd and "posterior_sigma" not in sd           # non-persistent (ddpm.py:105)
    want = (0.5 * torch.tensor(g[f"{kind}_{T}_posterior_log_variance_clipped"])).exp()    # ddpm.py:227
    assert torch.equal(m.posterior_sigma, want)


def test_sample_shapes_and_attrs():
    cfg = dddpm_cfg(128, 256, 3)
    m = DownsampleDDPM(cfg, Unet(cfg), "cpu", 3)
    assert m.sample_shape == [8, 32, 32] and m.x_shape == [3, 256, 256] and m.timesteps == 1000
    assert m.t_rec_max == 100 and m.force_latent is True
    with pytest.raises(ValueError):
        c = ddpm_cfg(32, 3, 16); c["loss_flat"] = "max"; DDPM(c, torch.nn.Identity(), "cpu", 3)
    with pytest.raises(ValueError):
        c = ddpm_cfg(32, 3, 16); c["beta_schedule"] = "sqrt"; DDPM(c, torch.nn.Identity(), "cpu", 3)
    with pytest.raises(AssertionError):
        c = ddpm_cfg(32, 3, 16); c["loss_type"] = "l1"; DDPM(c, torch.nn.Identity(), "cpu", 3)


def test_unet_rejects_unsupported_width():
    """GroupNorm(8, C) (reference blocks.py:75): widths that are not multiples of 8 fail in the reference as well"""
    for chan in (20, 4, 520):
        with pytest.raises(L.DDKError):
            Unet(dict(unet_chan=chan, unet_in=3, unet_dims=(1, 2), unet_dropout=0.0))
    Unet(dict(unet_chan=16, unet_in=3, unet_dims=(1, 2), unet_dropout=0.0))       # a multiple of 8: accepted (generic kernels)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ddk.h")).read()
    declared = set(re.findall(r"\b(ddk_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None
    assert L.load().ddk_version() == L.ABI_VERSION == 400


def test_pack_jobs_layout_prefix_sums():
    """ddk_pack_jobs_layout is host arithmetic: totals and first-block prefix sums of the job table (1024 elements per block)"""
    jobs = (L.PackJob * 3)()
    for j, (kind, p) in zip(jobs, [(0, [128, 8, 9, 32, 8, 32]), (3, [256, 128, 128]), (2, [128, 256, 9, 256, 128])]):
        j.src, j.dst, j.kind = 4096, 8192, kind
        for q, v in enumerate(p):
            j.p[q] = v
    blocks = L.load().ddk_pack_jobs_layout(jobs, 3)
    totals = [128 * 9 * 32, 256 * 128, 256 * 9 * 128]
    assert [j.total for j in jobs] == totals
    assert [j.block0 for j in jobs] == [0, 36, 36 + 32] and blocks == 36 + 32 + 288
    jobs[1].p[2] = 100                                   # i_pad not a multiple of 32
    assert L.load().ddk_pack_jobs_layout(jobs, 3) < 0


def test_plan_slots_cover_state_dict():
    cfg = ddpm_cfg(128, 8, 32)
    u = Unet(cfg)
    u.flops(1, 32, 32)                                   # creates the plan (host only)
    names = set(u._plan.slot_names)
    assert names - {"@sinusoidal_freqs"} == set(u.state_dict())
    assert abs(u.flops(1, 32, 32) / 1e9 - 4.5947) < 1e-3
    assert abs(Unet(ddpm_cfg(128, 3, 32)).flops(1, 32, 32) / 1e9 - 4.5803) < 1e-3   # cfg2
    assert abs(Unet(ddpm_cfg(128, 1, 32)).flops(1, 32, 32) / 1e9 - 4.5745) < 1e-3   # cfg1
    assert abs(u.flops(1, 16, 16) / 1e9 - 1.1496) < 1e-3                            # cfg3


def test_no_cpu_fallback():
    cfg = ddpm_cfg(32, 3, 16)
    u = Unet(cfg).eval()
    with pytest.raises(L.DDKError):
        with torch.no_grad():
            u(torch.zeros(1, 3, 16, 16), torch.zeros(1, dtype=torch.long))
    m = DDPM(cfg, u, "cpu", 3)
    with pytest.raises(L.DDKError):
        m.sample(2)
    with pytest.raises(L.DDKError):
        m.q_sample(torch.zeros(1, 3, 16, 16), torch.zeros(1, dtype=torch.long), torch.zeros(1, 3, 16, 16))
    from ddk import ops
    with pytest.raises(L.DDKError):
        ops.mish(torch.zeros(8))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing in the package, include/ or tools/ may import it (bench.py's cpu_baseline
    leg and __graft_entry__.smoke() are the two sanctioned users outside tests/)."""
    for top in ("downsampled-diffusion_amd", "include", "tools"):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dp, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src, os.path.join(dp, f)


def test_cli_args_surface():
    from utils import DATASETS, get_args
    cfg, mute = get_args({"lr": 1e-3}, DATASETS, models.MODEL_NAMES,
                         ["-m", "ddpm", "-d", "celeba_hq", "-e", "7", "-bs", "32", "-is", "256", "-downsample", "3", "-mute"])
    assert mute and cfg["dataset"] == "celeba_hq" and cfg["n_steps"] == 7 and cfg["batch_size"] == 32
    assert cfg["image_size"] == 256 and cfg["n_downsamples"] == 3 and cfg["model"] == "ddpm"
    cfg, mute = get_args({}, DATASETS, models.MODEL_NAMES, [])
    assert not mute and cfg["image_size"] == 32 and cfg["batch_size"] == 32 and cfg["n_steps"] == 500
    for d in ("celeba_hq_65", "celeba_hq_64", "mnist"):
        assert d in DATASETS


def test_fix_samples_format():
    from utils import fix_samples
    from oracle.diffusion_ref import fix_samples as ref_fix
    x = torch.randn(3, 3, 8, 8)
    out = fix_samples(x)
    assert out.shape == (3, 8, 8, 3) and out.dtype == np.float32
    assert np.allclose(out, ref_fix(x), atol=1e-4) and out.min() == 0 and abs(out.max() - 255) < 1e-3


def _bench(env, *argv, timeout=120):
    import subprocess
    import sys
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` as the driver may call it (no torchrun in front): the parent -- which never touches the GPU, this
    container has none -- starts N rank processes with the torchrun environment and relays rank 0's ONE JSON line."""
    import json
    p = _bench({"DDK_BENCH_SAME_DEVICE": "1", "DDK_BENCH_LAUNCH_PROBE": "1"}, "--gpus", "3", "--steps", "4")
    assert p.returncode == 0, p.stderr[-2000:]
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1                                      # one line on stdout, the other ranks' output went to stderr
    line = json.loads(out[0])
    assert line["n_gpus"] == 3 and line["rank"] == 0 and line["argv"] == ["--gpus", "3", "--steps", "4"]
    assert line["master"][0] == "127.0.0.1" and int(line["master"][1]) > 0
    assert "probe rank 1 of 3" in p.stderr and "probe rank 2 of 3" in p.stderr


def test_bench_launcher_fails_loudly():
    """a rank that dies takes the job down (the others are killed, exit code non-zero), a job that outlives its limit is stopped,
    and more ranks than GPUs is refused unless it is the same-device rehearsal"""
    import time
    t0 = time.monotonic()
    p = _bench({"DDK_BENCH_SAME_DEVICE": "1", "DDK_BENCH_LAUNCH_PROBE": "fail1"}, "--gpus", "2")
    assert p.returncode != 0 and "rank 1 exited with code 7" in p.stderr and p.stdout.strip() == ""
    assert time.monotonic() - t0 < 40                         # rank 0 (sleeping for a minute) was killed, not waited for
    p = _bench({"DDK_BENCH_SAME_DEVICE": "1", "DDK_BENCH_LAUNCH_PROBE": "hang", "DDK_BENCH_TIMEOUT": "2"}, "--gpus", "2")
    assert p.returncode != 0 and "DDK_BENCH_TIMEOUT" in p.stderr
    if not torch.cuda.is_available():
        p = _bench({"DDK_BENCH_LAUNCH_PROBE": "1"}, "--gpus", "2")
        assert p.returncode == 2 and "only 0 GPU(s) visible" in p.stderr


def test_bench_launcher_default_timeout_is_under_the_drivers():
    """the driver stops bench.py at 600 s: the launcher's own limit must come first, so that IT stops the ranks"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(r'DDK_BENCH_TIMEOUT", "(\d+)"', src)
    assert m and int(m.group(1)) <= 500


def test_bench_launcher_parent_never_imports_torch(tmp_path):
    """the parent of `--gpus N` counts devices from sysfs and starts children: it must not load torch (and with it HIP) at all"""
    import subprocess
    import sys
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '8', '--steps', '1']\n"
            "import os; os.environ.update(DDK_BENCH_SAME_DEVICE='1', DDK_BENCH_LAUNCH_PROBE='1'); os.environ.pop('WORLD_SIZE', None)\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    rc = e.code\n"
            "bad = [m for m in sys.modules if m == 'torch' or m.startswith('torch.')]\n"
            "print('RC', rc, 'TORCH', len(bad), file=sys.stderr)") % os.path.join(ROOT, "bench.py")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert "RC 0 TORCH 0" in p.stderr, p.stderr[-2000:]
    import json
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8                                # the N = 8 launch itself, rehearsed with probe children
    for r in range(1, 8):
        assert f"probe rank {r} of 8" in p.stderr


def _alive(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"      # a zombie nobody reaps holds no resource
    except OSError:
        return False


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGKILL", "SIGINT"])
def test_bench_launcher_children_never_outlive_the_parent(tmp_path, sig):
    """What the driver's timeout does to `bench.py --gpus 8`: SIGTERM (the parent's handler kills the children's process groups) or
    SIGKILL (no handler runs: every child asked the kernel for SIGKILL on its parent's death before its program started).  Either
    way no rank process is left behind to hold a GPU."""
    import signal
    import subprocess
    import sys
    import time
    e = dict(os.environ, DDK_BENCH_SAME_DEVICE="1", DDK_BENCH_LAUNCH_PROBE="pids", DDK_BENCH_PROBE_DIR=str(tmp_path))
    e.pop("WORLD_SIZE", None)
    n = 8
    parent = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=e, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True)
    try:
        t0 = time.monotonic()
        files = [tmp_path / f"rank{r}.pid" for r in range(n)]
        while not all(f.exists() and f.read_text().strip() for f in files):
            assert time.monotonic() - t0 < 60 and parent.poll() is None, "the ranks did not start"
            time.sleep(0.1)
        pids = [int(f.read_text()) for f in files]
        assert all(_alive(p) for p in pids)
        parent.send_signal(getattr(signal, sig))
        rc = parent.wait(timeout=40)
        assert rc != 0
        t1 = time.monotonic()
        while any(_alive(p) for p in pids):
            assert time.monotonic() - t1 < 10, f"rank processes survived the parent's {sig}: {[p for p in pids if _alive(p)]}"
            time.sleep(0.05)
    finally:
        if parent.poll() is None:
            parent.kill()
        for f in tmp_path.glob("rank*.pid"):
            try:
                os.kill(int(f.read_text()), signal.SIGKILL)
            except (OSError, ValueError):
                pass


def test_bench_launcher_retries_on_a_taken_rendezvous_port(tmp_path):
    """the port is chosen by bind-then-close; when a rank reports it taken (exit 98) the job is started once more on another port"""
    import json
    p = _bench({"DDK_BENCH_SAME_DEVICE": "1", "DDK_BENCH_LAUNCH_PROBE": "eaddr", "DDK_BENCH_PROBE_DIR": str(tmp_path)}, "--gpus", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    assert "starting the ranks again on a new port" in p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 2


def test_visible_gpus_counts_from_sysfs_without_the_runtime(monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    n = b.visible_gpus()
    assert n is None or n >= 0
    if n:
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
        assert b.visible_gpus() == 1


def test_graph_owner_registers_and_unregisters_by_identity():
    """ddk.ops.graph_owner: while a capture runs, what captured launches address through device tables is appended to the capturing
    object's own list; lists are told apart by identity (two empty lists compare equal), nesting unwinds in order, and the
    process-wide list at the bottom is never removed"""
    from ddk import ops
    base = ops._graph_keep[0]
    depth = len(ops._graph_keep)
    a, b = [], []
    with ops.graph_owner(a):
        assert ops._graph_keep[-1] is a
        with ops.graph_owner(b):
            assert ops._graph_keep[-1] is b and ops._graph_keep[-2] is a
            ops._graph_keep[-1].append("x")
        assert ops._graph_keep[-1] is a and b == ["x"] and a == []
    assert len(ops._graph_keep) == depth and ops._graph_keep[0] is base
