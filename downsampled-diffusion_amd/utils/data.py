"""Input pipeline boundary (reference utils/data.py:12-246).

The reference builds torchvision datasets (ToTensor -> Resize -> CenterCrop -> x*2-1).  Neither torchvision nor
any dataset exists offline, and the input pipeline is outside the accelerated path, so this
module provides the same entry points over a SYNTHETIC dataset of the right shape (uniform in [-1, 1]);
if torchvision and the data are present the real datasets are used.
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset

DATASETS = ['cifar10', 'cifar100', 'mnist', 'omniglot', 'celeba', 'celeba_hq_65', 'celeba_hq_64', 'celeba_hq']
_GRAY = ('mnist', 'omniglot')


def get_color_channels(dataset: str) -> int:
    """data.py: 1 channel for mnist / omniglot, else 3."""
    return 1 if dataset in _GRAY else 3


class SyntheticImages(Dataset):
    """Deterministic uniform [-1,1] images, generated per index (no storage)."""

    def __init__(self, channels, size, length=50000, seed=4321):
        self.shape, self.length, self.seed = (channels, size, size), length, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return torch.rand(self.shape, generator=g) * 2 - 1, 0


def get_dataloader(config: dict, device: str = 'cpu', train: bool = True, data_root: str = './data/', val_split: float = 0):
    """Returns (train_loader, val_loader) like data.py:103-140; synthetic unless DDPM_REAL_DATA=1 and torchvision works."""
    channels = get_color_channels(config['dataset'])
    size, bs = config['image_size'], config['batch_size']
    ds = None
    if os.environ.get('DDPM_REAL_DATA') == '1':
        ds = _try_real_dataset(config, data_root, train)
    if ds is None:
        ds = SyntheticImages(channels, size)
    kw = dict(num_workers=4, pin_memory=True) if device == 'cuda' else {}
    if val_split and val_split > 0:
        n_val = int(len(ds) * val_split)
        tr, va = torch.utils.data.random_split(ds, [len(ds) - n_val, n_val])
        return (DataLoader(tr, batch_size=bs, shuffle=True, drop_last=True, **kw),
                DataLoader(va, batch_size=bs, shuffle=False, drop_last=True, **kw))
    return DataLoader(ds, batch_size=bs, shuffle=train, drop_last=True, **kw), None


def _try_real_dataset(config, data_root, train):
    try:
        from torchvision import datasets, transforms as T
    except Exception:
        return None
    tf = T.Compose([T.ToTensor(), T.Resize(config['image_size']), T.CenterCrop(config['image_size']),
                    T.Lambda(lambda x: x * 2 - 1)])
    name = config['dataset']
    try:
        if name == 'cifar10':
            return datasets.CIFAR10(data_root, train=train, transform=tf, download=False)
        if name == 'cifar100':
            return datasets.CIFAR100(data_root, train=train, transform=tf, download=False)
        if name == 'mnist':
            return datasets.MNIST(data_root, train=train, transform=tf, download=False)
        folder = {'celeba': 'celeba', 'celeba_hq': 'celeba_hq', 'celeba_hq_64': 'celeba_hq_64',
                  'celeba_hq_65': 'celeba_hq_64'}.get(name)
        if folder:
            return datasets.ImageFolder(os.path.join(data_root, folder), transform=tf)
    except Exception:
        return None
    return None
