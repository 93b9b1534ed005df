#!/usr/bin/env python3
"""Which torch-level ops (copies, adds, cats, fills) a training micro-batch issues besides the library's kernels, with the Python
frames that issue them (torch.profiler with stacks):  python tools/train_torch_ops.py [cfg3|cfg2]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from train_bench import DEV, cfg  # noqa: E402
from models import DDPM, DownsampleDDPMAutoencoder, Unet  # noqa: E402
from trainers.optim import FusedAdam  # noqa: E402
from utils import synthetic as syn  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg3":
    c = cfg(128, 8, 64, down=2); model = DownsampleDDPMAutoencoder(c, Unet(c), DEV, 3); xshape = (64, 3, 64, 64)
else:
    c = cfg(128, 3, 32); model = DDPM(c, Unet(c), DEV, 3); xshape = (64, 3, 32, 32)
model = model.to(DEV).train()
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
opt = FusedAdam(model, lr=2e-4)
x = torch.rand(xshape, device=DEV) * 2 - 1


def micro():
    out = model(x)
    obj = out[0] if isinstance(out, tuple) else out
    (obj / 2).backward()


micro(); micro(); opt.step(); opt.zero_grad()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    micro()
torch.cuda.synchronize()
by = collections.Counter()
where = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::detach", "aten::reshape",
                   "aten::select", "aten::slice", "aten::alias", "aten::_unsafe_view", "aten::permute", "aten::transpose", "aten::t",
                   "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::narrow", "aten::result_type", "aten::item",
                   "aten::_local_scalar_dense", "aten::is_nonzero", "aten::resize_", "aten::set_", "aten::lift_fresh", "aten::to",
                   "aten::contiguous", "aten::clone", "aten::zeros_like", "aten::zeros", "aten::ones_like", "aten::_to_copy"):
        continue
    shapes = str(ev.input_shapes)[:60]
    by[ev.name] += 1
    frames = [f for f in (ev.stack or []) if "downsampled-diffusion_amd" in f or "autograd" in f][:3]
    where[ev.name][(shapes, " <- ".join(f.split("downsampled-diffusion_amd/")[-1] for f in frames))] += 1
for name, n in by.most_common(12):
    print(f"{name}: {n}")
    for (shapes, fr), k in where[name].most_common(8):
        print(f"    {k:4d} x {shapes}  {fr}")
