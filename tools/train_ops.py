import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from train_bench import cfg, DEV
from models import DDPM, Unet
from trainers.optim import FusedAdam
from utils import synthetic as syn
c = cfg(128, 3, 32); model = DDPM(c, Unet(c), DEV, 3).to(DEV).train()
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
opt = FusedAdam(model, lr=2e-4)
x = torch.rand((64, 3, 32, 32), device=DEV) * 2 - 1
def step():
    for _ in range(2):
        obj = model(x); (obj / 2).backward()
    opt.step(); opt.zero_grad()
for _ in range(2): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
from collections import Counter
cnt = Counter()
for e in prof.events():
    if e.name.startswith("aten::"):
        cnt[e.name] += 1
for k, v in cnt.most_common(25):
    print(f"{v:6d}  {k}")
