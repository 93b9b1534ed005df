import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ["DDK_TRACE"] = "1"
import torch
from models import Unet
cfg = dict(unet_chan=128, unet_in=8, unet_dims=(1, 2, 2, 2), unet_dropout=0.0)
u = Unet(cfg).cuda().eval()
with torch.no_grad():
    u(torch.randn(32, 8, 32, 32, device="cuda"), torch.zeros(32, dtype=torch.long, device="cuda"))
torch.cuda.synchronize()
