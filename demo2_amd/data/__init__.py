"""Data-side contract of the hot path: the identity-balanced samplers and the batch layout the
reference's loaders hand to engine/processor.py, with an MI355X-first host->device hand-over.

    RandomIdentitySampler        data/datasets/sampler.py:8-67
    RandomIdentitySampler_DDP    data/datasets/sampler_ddp.py:117-201
    train_collate_fn / val_collate_fn   data/datasets/make_dataloader.py:142-184
    DeviceFeeder                 replaces the three per-modality .to(device) calls of engine/processor.py:60-64

Datasets, image decoding and augmentation (torchvision / PIL) stay the reference's own: they never touch the GPU path."""
from .collate import DeviceFeeder, train_collate_fn, val_collate_fn  # noqa: F401
from .sampler import RandomIdentitySampler, RandomIdentitySampler_DDP, epoch_length  # noqa: F401
