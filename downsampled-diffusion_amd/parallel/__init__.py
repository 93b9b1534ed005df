"""Multi-GPU helpers: one process per GPU over torch.distributed (backend 'nccl' = RCCL on ROCm, 'gloo' on CPU tests)."""
from .dist import (all_ranks_agree, all_reduce_flat_, barrier, broadcast_module_, flatten_tensors, init_from_env, is_main_rank, main_rank_does, shard_batch,
                   shard_sizes, unflatten_into_)
