import collections, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from torch.profiler import ProfilerActivity, profile
from train_bench import DEV, cfg
from models import DownsampleDDPMAutoencoder, Unet
from trainers.optim import FusedAdam
from utils import synthetic as syn
c = cfg(128, 8, 64, down=2); model = DownsampleDDPMAutoencoder(c, Unet(c), DEV, 3).to(DEV).train()
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
opt = FusedAdam(model, lr=2e-4)
x = torch.rand((64, 3, 64, 64), device=DEV) * 2 - 1
def micro():
    out = model(x); (out[0] / 2).backward()
micro(); micro(); opt.step(); opt.zero_grad(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    micro()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::_to_copy", "aten::zeros", "aten::zeros_like",
                   "aten::item", "aten::_local_scalar_dense", "aten::to", "aten::stack", "aten::select", "aten::index"):
        st = [f for f in (ev.stack or []) if "site-packages" not in f and "dist-packages" not in f][:4]
        cnt[(ev.name, str(ev.input_shapes)[:50], " <- ".join(s.split("/")[-1] for s in st))] += 1
for k, v in cnt.most_common(40):
    print(v, k)
