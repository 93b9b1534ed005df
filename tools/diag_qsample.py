import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops
from oracle import diffusion_ref as D
buf = D.schedule_buffers("linear", 1000)
g = torch.Generator().manual_seed(100)
x = torch.randn(5, 3, 16, 16, generator=g); eps = torch.randn(5, 3, 16, 16, generator=g)
t = torch.tensor([0, 1, 499, 998, 999])
out = ops.q_sample(x.cuda(), eps.cuda(), t.cuda(), buf["sqrt_alphas_cumprod"].cuda(), buf["sqrt_one_minus_alphas_cumprod"].cuda()).cpu()
ref = D.q_sample(buf, x, t, eps)
d = (out - ref).abs()
print("max diff", d.max().item(), "n mismatch", (out != ref).sum().item(), "of", out.numel())
a = buf["sqrt_alphas_cumprod"][t].reshape(5,1,1,1); b = buf["sqrt_one_minus_alphas_cumprod"][t].reshape(5,1,1,1)
ref_fma = torch.addcmul(b * eps, a, x)  # not exact fma but indicative
ref64 = (a.double()*x.double() + b.double()*eps.double())
print("out vs f64", (out.double()-ref64).abs().max().item(), "ref vs f64", (ref.double()-ref64).abs().max().item())
# which samples mismatch
print("per-sample mismatches", [(out[i] != ref[i]).sum().item() for i in range(5)])
gt = torch.ops.aten.add(torch.ops.aten.mul(a.cuda(), x.cuda()), torch.ops.aten.mul(b.cuda(), eps.cuda())).cpu()
print("torch-gpu vs cpu mismatches", (gt != ref).sum().item(), " torch-gpu vs ours", (gt != out).sum().item())
