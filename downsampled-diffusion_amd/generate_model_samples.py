#!/usr/bin/env python3
"""Sampling driver with the reference's behaviour (reference generate_model_samples.py:13-69): load
``{saved_model}.pt`` from CHECKPOINT_DIR (EMA weights preferred), rebuild the model from the stored config, draw
``fid_samples`` images in batches with ``model.sample``, convert with ``fix_samples`` and ``np.save`` the list of
batches (+ latent list for dDDPM), printing the same three timing lines.

The reference hard-codes its constants; here they are the defaults of optional flags.  Extensions:
  * runs one process per GPU under torchrun: rank 0 broadcasts the weights once (RCCL), every rank samples a contiguous
    run of the job's batches (no collective in the loop) and writes ``{saved_model}.rank{r}.npy``; rank 0 then merges the
    shards into the single ``{saved_model}.npy`` the evaluator loads (reference evaluate_ddpm.py:52).  A batch's draws
    (x_T, Philox key) depend only on its GLOBAL batch index, so the merged file is bit-identical to a 1-process run;
  * the device -> host stage is asynchronous (utils.OutputStage: pinned double buffer + events), so a batch's copy
    overlaps the next batch's sampling;
  * ``--synthetic CONFIG`` builds deterministic synthetic weights when no checkpoint exists (offline boxes).
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from models import DDPM, DownsampleDDPM, Unet
from parallel import barrier, broadcast_module_, init_from_env, main_rank_does, shard_sizes
from utils import (CHECKPOINT_DIR, SAMPLE_DIR, SAMPLE_LATENT_DIR, OutputStage, get_color_channels, get_model_state_dict,
                   load_checkpoint_file, merge_rank_shards)
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
    ap.add_argument("--seed", type=int, default=1234, help="base seed: batch g of the job draws from seed + g")
    ap.add_argument("--keep_shards", action="store_true", help="keep the per-rank .rank{r}.npy files after the merge")
    args = ap.parse_args()

    rank, world = init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if local >= torch.cuda.device_count():
        # several ranks on one GPU only work as a gloo rehearsal: RCCL wants one device per rank and would deadlock or error
        if world > 1 and torch.distributed.get_backend() != "gloo":
            raise RuntimeError(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible: one process per GPU is "
                               "required with the nccl (RCCL) backend; set DDK_DIST_BACKEND=gloo to rehearse on one GPU")
        local = 0
    # ranks that share one GPU (the gloo rehearsal above, or a launcher that maps several ranks onto one device) cannot host a whole
    # cluster of the in-launch GroupNorm: switch that path off up front instead of waiting for its first (loud) give-up
    shared_gpu = world > 1 and (int(os.environ.get("LOCAL_RANK", "0")) >= torch.cuda.device_count()
                                or int(os.environ.get("LOCAL_WORLD_SIZE", "0")) > torch.cuda.device_count())
    if shared_gpu:
        print(f"[rank {rank}] several ranks share one GPU: the in-launch GroupNorm is switched off", flush=True)
    torch.cuda.set_device(local)
    device = f"cuda:{local}"

    step = 0
    if args.synthetic:
        with open(args.synthetic) as f:
            config = json.load(f)
        model_state_dict = None
    else:
        save_data = load_checkpoint_file(os.path.join(CHECKPOINT_DIR, f"{args.saved_model}.pt"))
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
    model.rng_stream_id = 0
    if shared_gpu:
        latent_model.plan().set_option(latent_model.plan().OPT_CLUSTER_GROUPNORM, 0)

    # the job = ceil(fid_samples / batch) batches; rank r takes a contiguous run of them
    bs = config["batch_size"]
    n_batches_total = int(np.ceil(args.fid_samples / bs))
    mine = shard_sizes(n_batches_total, world)
    g0, n_batches = sum(mine[:rank]), mine[rank]
    if rank == 0:
        print(f"\nGenerating {args.fid_samples} samples from checkpoint {args.saved_model}.")
        print(f"Trained for {step} steps with configuration dict:")
        print(json.dumps(config, sort_keys=False, indent=4, default=str) + "\n")
    stage, latent_stage = OutputStage(), OutputStage()
    time_start = time.time()
    for g in range(g0, g0 + n_batches):
        torch.manual_seed(args.seed + g)          # x_T and the Philox key of batch g: the same on whichever rank runs it
        samples = model.sample(bs, args.sample_every, args.early_stop)
        if config["model"] == "dddpm":
            samples, latent_samples = samples
            latent_stage.submit(latent_samples)
        stage.submit(samples)
    sample_list, latent_list = stage.finish(), latent_stage.finish()
    torch.cuda.synchronize()
    sampling_time = time.time() - time_start

    print(f"Using batch size {bs}")
    print(f"Total time: {sampling_time}")
    print(f"Sample time: {sampling_time / max(n_batches * bs, 1)}")
    print(f"Batch time: {sampling_time / max(n_batches, 1)}")

    def save(directory, name, batches):
        os.makedirs(directory, exist_ok=True)
        base = os.path.join(directory, name)
        if world == 1:
            np.save(base, batches, allow_pickle=False)
            return base
        # per-rank shards on a filesystem rank 0 can read (one node, or a shared mount: merge_rank_shards checks every shard exists)
        np.save(f"{base}.rank{rank}", batches, allow_pickle=False)
        barrier()                                   # every shard is on disk
        main_rank_does(lambda: merge_rank_shards(base, world, remove=not args.keep_shards) is None, "merge of the sampling shards")
        return base

    save_path = save(args.out_dir or SAMPLE_DIR, args.saved_model, sample_list)
    if rank == 0:
        print(f"Samples saved to {save_path}")
    if config["model"] == "dddpm":
        save_path = save(args.out_dir or SAMPLE_LATENT_DIR, args.saved_model + "_latent" if args.out_dir else args.saved_model, latent_list)
        if rank == 0:
            print(f"Latent samples saved to {save_path}")


if __name__ == "__main__":
    main()
