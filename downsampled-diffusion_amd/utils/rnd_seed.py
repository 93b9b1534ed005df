"""Seeding (reference utils/rnd_seed.py:7-17)."""
import os
import random

import numpy as np
import torch


def seed_everything(seed):
    if seed is None:
        return None
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
