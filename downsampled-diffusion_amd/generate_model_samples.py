#!/usr/bin/env python3
"""Sampling driver with the reference's behaviour (reference generate_model_samples.py:13-69): load
``{saved_model}.pt`` from CHECKPOINT_DIR (EMA weights preferred), rebuild the model from the stored config, draw
``fid_samples`` images in batches with ``model.sample``, convert with ``fix_samples`` and ``np.save`` the list of
batches (+ latent list for dDDPM), printing the same three timing lines.

The reference hard-codes its constants; here they are the defaults of optional flags.  Extensions:
  * runs one process per GPU under torchrun: rank 0 broadcasts the weights once (RCCL), every rank samples its own
    batches with its own Philox stream and writes ``{saved_model}.rank{r}.npy`` (no collective in the loop);
  * ``--synthetic CONFIG`` builds deterministic synthetic weights when no checkpoint exists (offline boxes).
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from models import DDPM, DownsampleDDPM, Unet
from parallel import broadcast_module_, init_from_env, shard_sizes
from utils import CHECKPOINT_DIR, SAMPLE_DIR, SAMPLE_LATENT_DIR, fix_samples, get_color_channels, get_model_state_dict
from utils import synthetic as syn


def main():
    ap = argparse.ArgumentParser(description="Generate samples from a trained DDPM / dDDPM checkpoint.")
    ap.add_argument("--saved_model", default="celeba_x2")
    ap.add_argument("--fid_samples", type=int, default=50000)
    ap.add_argument("--batch_size", type=int, default=192)
    ap.add_argument("--sample_every", type=int, default=1)
    ap.add_argument("--early_stop", type=int, default=None)
    ap.add_argument("--synthetic", default=None, help="JSON config file: use closed-form synthetic weights, no checkpoint")
    ap.add_argument("--out_dir", default=None)
    args = ap.parse_args()

    rank, world = init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = f"cuda:{local}"

    step = 0
    if args.synthetic:
        with open(args.synthetic) as f:
            config = json.load(f)
        model_state_dict = None
    else:
        save_data = torch.load(os.path.join(CHECKPOINT_DIR, f"{args.saved_model}.pt"), map_location="cpu")
        model_state_dict = get_model_state_dict(save_data)
        config = save_data["config"]
        step = save_data.get("step", 0)
    config["batch_size"] = args.batch_size

    latent_model = Unet(config)
    color_channels = get_color_channels(config["dataset"])
    if config["model"] == "ddpm":
        model = DDPM(config, latent_model, device, color_channels)
    elif config["model"] == "dddpm":
        model = DownsampleDDPM(config, latent_model, device, color_channels)
    else:
        raise NotImplementedError(config["model"])
    if rank == 0:
        if model_state_dict is None:
            model_state_dict = syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS)
        model.load_state_dict(model_state_dict)
    model = model.to(device).eval()
    broadcast_module_(model, src=0)
    model.rng_stream_id = rank
    torch.manual_seed(1234 + rank)

    n_mine = shard_sizes(args.fid_samples, world)[rank]
    if rank == 0:
        print(f"\nGenerating {args.fid_samples} samples from checkpoint {args.saved_model}.")
        print(f"Trained for {step} steps with configuration dict:")
        print(json.dumps(config, sort_keys=False, indent=4, default=str) + "\n")
    sample_list, latent_list = [], []
    time_start = time.time()
    n_batches = int(np.ceil(n_mine / config["batch_size"]))
    for _ in range(n_batches):
        samples = model.sample(config["batch_size"], args.sample_every, args.early_stop)
        if config["model"] == "dddpm":
            samples, latent_samples = samples
            latent_list.append(fix_samples(latent_samples))
        sample_list.append(fix_samples(samples))
    torch.cuda.synchronize()
    sampling_time = time.time() - time_start

    print(f"Using batch size {config['batch_size']}")
    print(f"Total time: {sampling_time}")
    print(f"Sample time: {sampling_time / max(n_mine, 1)}")
    print(f"Batch time: {sampling_time / max(n_batches, 1)}")

    suffix = "" if world == 1 else f".rank{rank}"
    out_dir = args.out_dir or SAMPLE_DIR
    os.makedirs(out_dir, exist_ok=True)
    save_path = os.path.join(out_dir, args.saved_model + suffix)
    np.save(save_path, sample_list, allow_pickle=False)
    print(f"Samples saved to {save_path}")
    if config["model"] == "dddpm":
        lat_dir = args.out_dir or SAMPLE_LATENT_DIR
        os.makedirs(lat_dir, exist_ok=True)
        save_path = os.path.join(lat_dir, args.saved_model + "_latent" + suffix if args.out_dir else args.saved_model + suffix)
        np.save(save_path, latent_list, allow_pickle=False)
        print(f"Latent samples saved to {save_path}")


if __name__ == "__main__":
    main()
