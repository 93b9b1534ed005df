"""Host logic of the sampler output stage (no GPU): shard merge = the single file the evaluator loads
(reference generate_model_samples.py:61-69, evaluate_ddpm.py:52), checkpoint reader, EMA preference."""
import numpy as np
import pytest
import torch

from utils import get_model_state_dict, load_checkpoint_file, merge_rank_shards


def test_merge_rank_shards_concatenates_in_rank_order(tmp_path):
    base = str(tmp_path / "run")
    shards = [np.arange(2 * 2 * 4 * 4 * 3, dtype=np.float32).reshape(2, 2, 4, 4, 3) + 1000 * r for r in range(3)]
    shards[2] = shards[2][:1]                                   # the last rank may hold fewer batches
    for r, a in enumerate(shards):
        np.save(f"{base}.rank{r}", a)
    merged = merge_rank_shards(base, 3, remove=True)
    assert merged.shape == (5, 2, 4, 4, 3)
    assert np.array_equal(merged, np.concatenate(shards, axis=0))
    assert np.array_equal(np.load(base + ".npy"), merged)       # what evaluate_ddpm.py:52 loads
    assert not list(tmp_path.glob("run.rank*.npy"))
    with pytest.raises(FileNotFoundError):
        merge_rank_shards(base, 2)


def test_reference_shaped_checkpoint_loads_and_prefers_ema(tmp_path):
    """A checkpoint as the reference trainer writes it (trainers/trainer_ddpm.py:49-62): torch.optim.Adam state, numpy scalars
    in train_losses, a config with tuples, an ema_model.  torch >= 2.6 refuses these with weights_only=True; the one loader
    helper reads them, and get_model_state_dict picks the EMA weights (utils/utils.py:51-54)."""
    w = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(w.parameters(), lr=2e-4)
    w(torch.ones(1, 3)).sum().backward()
    opt.step()
    ck = {"optimizer": opt.state_dict(), "model": w.state_dict(), "config": {"unet_dims": (1, 2, 2, 2), "lr": 2e-4},
          "train_losses": [np.mean([1.0, 2.0]), np.float64(0.5)], "step": 1,
          "ema_model": {k: v + 1 for k, v in w.state_dict().items()}}
    path = tmp_path / "ref.pt"
    torch.save(ck, path)
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)
    got = load_checkpoint_file(str(path))
    assert got["config"]["unet_dims"] == (1, 2, 2, 2) and got["step"] == 1
    sd = get_model_state_dict(got)
    assert torch.equal(sd["weight"], w.state_dict()["weight"] + 1)
    del got["ema_model"]
    assert torch.equal(get_model_state_dict(got)["weight"], w.state_dict()["weight"])
