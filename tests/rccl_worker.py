"""Child process of tests/test_rccl_gpu.py: the product's two collectives over RCCL (torch.distributed backend "nccl") on a
world of ONE rank -- the only RCCL world a 1-GPU box can host.  Started fresh, before any GPU call, with the process group bound
to cuda:0 (`device_id`), exactly like one rank of the 8-GPU launch.

  C1  parallel.broadcast_module_  on the cfg4 model: UNet + x3 encoder / decoder + schedule buffers, one flat fp32 bucket (~90.7 MB)
  C2  parallel.all_reduce_flat_(average=True) on the optimiser's flat gradient bucket (all parameters of the same model)
Both must be identities on one rank; what the run proves is that librccl loads and the device-tensor path of parallel/dist.py
works on an MI355X.  Semantics being parallelised: trainers/trainer_ddpm.py:118-144 (the reference itself has no collectives)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, HERE]


def main():
    out_path = sys.argv[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from models import DownsampleDDPMAutoencoder, Unet
    from parallel import all_reduce_flat_, broadcast_module_
    from trainers.optim import FusedAdam
    from utils import synthetic as syn
    cfg = dict(unet_chan=128, unet_in=8, unet_dims=(1, 2, 2, 2), unet_dropout=0.1, image_size=256, T=1000, loss_type="simple",
               beta_schedule="linear", loss_flat="sum", d_mode="convolutional_res", u_mode="convolutional_res", d_dropout=0, d_chans=64,
               d_n_blocks=3, u_n_blocks=3, ae_loss=True, t_rec_max=100, force_latent=True, n_downsamples=3)
    model = DownsampleDDPMAutoencoder(cfg, Unet(cfg), "cuda", 3)
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    model = model.to(dev)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    nbytes = broadcast_module_(model, src=0, force=True)                      # C1 over RCCL
    torch.cuda.synchronize()
    same = all(torch.equal(v, before[k]) for k, v in model.state_dict().items())

    opt = FusedAdam(model, lr=2e-4)
    flat = opt.fp.grad
    g = torch.Generator(device="cpu").manual_seed(5)
    flat.copy_(torch.randn(flat.numel(), generator=g))
    ref = flat.detach().clone()
    all_reduce_flat_(flat, average=True, force=True)                           # C2 over RCCL
    torch.cuda.synchronize()
    res = dict(backend=dist.get_backend(), world=dist.get_world_size(), broadcast_bytes=int(nbytes), broadcast_identity=bool(same),
               grad_bucket_bytes=int(flat.numel() * 4), all_reduce_identity=bool(torch.equal(flat, ref)),
               n_params=int(sum(p.numel() for p in model.parameters())))
    dist.barrier()
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
