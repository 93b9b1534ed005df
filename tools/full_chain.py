import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from sample_bench import cfg
from models import DownsampleDDPM, Unet
from utils import synthetic as syn, fix_samples
c = cfg(8, 256, down=3)
m = DownsampleDDPM(c, Unet(c), "cuda", 3).to("cuda").eval()
m.load_state_dict(syn.fill_state_dict(m.state_dict(), skip=syn.SCHEDULE_KEYS))
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, z = m.sample(32)
    img = fix_samples(x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"chain {i}: model.sample(32) + fix_samples: {dt:.3f} s -> {32 / dt:.2f} images/s; finite={bool(torch.isfinite(x).all())} "
          f"x range [{float(x.min()):.3f}, {float(x.max()):.3f}], z std {float(z.std()):.3f}, img shape {img.shape}", flush=True)
