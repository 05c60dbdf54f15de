"""Batch layout of the reference's loaders (data/datasets/make_dataloader.py:142-184) and the host->device hand-over.

A dataset item is ((rgb, ni, ti) image tensors (3,H,W), pid, camid, viewid, img_path) (data/datasets/bases.py).  The collate
functions return exactly the reference's tuples -- `imgs` a dict {'RGB','NI','TI'} of (B,3,H,W) float32 -- so the loops of
engine/processor.py:57-66,232-236 run unchanged.  MI355X-first detail: the three modality tensors are VIEWS of one
(3, B, 3, H, W) buffer, modality-major, which is the [RGB; NI; TI] order in which the shared trunk consumes them as one batch of
3B images (hip_forward.trunk_forward); `DeviceFeeder` pins that buffer and moves it with ONE asynchronous copy per batch on a
side stream, double-buffered, so the copy of batch i+1 overlaps the step on batch i (measured on the bench: -0.9 %)."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator

import torch

MODALITIES = ("RGB", "NI", "TI")


def _stack_modalities(imgs) -> Dict[str, torch.Tensor]:
    B = len(imgs)
    first = imgs[0][0]
    buf = torch.empty((3, B) + tuple(first.shape), dtype=first.dtype)
    for b, triple in enumerate(imgs):
        if len(triple) != 3:
            raise ValueError(f"sample {b}: expected (RGB, NI, TI) images, got {len(triple)}")
        for m in range(3):
            buf[m, b].copy_(triple[m])
    return {name: buf[m] for m, name in enumerate(MODALITIES)}


def train_collate_fn(batch):
    """-> (imgs, pids int64 (B,), camids int64 (B,), viewids int64 (B,), img_paths)   (make_dataloader.py:142-163)"""
    imgs, pids, camids, viewids, paths = zip(*batch)
    return (_stack_modalities(imgs), torch.tensor(pids, dtype=torch.int64), torch.tensor(camids, dtype=torch.int64),
            torch.tensor(viewids, dtype=torch.int64), paths)


def val_collate_fn(batch):
    """-> (imgs, pids (tuple of ints), camids int64 (B,), camids_batch int64 (B,), viewids int64 (B,), img_paths)
    (make_dataloader.py:166-184: the pids stay a tuple, the camera ids appear twice)"""
    imgs, pids, camids, viewids, paths = zip(*batch)
    cam = torch.tensor(camids, dtype=torch.int64)
    return _stack_modalities(imgs), pids, cam, cam, torch.tensor(viewids, dtype=torch.int64), paths


def _parent(imgs: Dict[str, torch.Tensor]):
    """The (3,B,3,H,W) buffer the three modality views share, or None if they do not come from the collate functions above."""
    r, n, t = (imgs[k] for k in MODALITIES)
    if r.is_contiguous() and n.is_contiguous() and t.is_contiguous() and r.shape == n.shape == t.shape and r.dtype == n.dtype == t.dtype:
        step = r.numel() * r.element_size()
        if (n.data_ptr() == r.data_ptr() + step and t.data_ptr() == r.data_ptr() + 2 * step and r.storage_offset() == 0
                and r.untyped_storage().nbytes() >= 3 * step):
            return torch.as_strided(r, (3,) + tuple(r.shape), (r.numel(),) + tuple(r.stride()))
    return None


class DeviceFeeder:
    """Wraps a loader whose batches start with the `imgs` dict: yields the same tuples with every tensor already on `device`.
    Two device buffers, one side stream: while the model works on batch i, batch i+1 is copied; the consumer's stream waits on
    the copy's event only.  Tensors after the first element (labels, camera ids) are small and go the same way."""

    def __init__(self, loader: Iterable, device, pin: bool = True):
        self.loader, self.device = loader, torch.device(device)
        self.pin = pin and self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def _upload(self, batch):
        imgs = batch[0]
        whole = _parent(imgs)
        if whole is None:
            whole = torch.stack([imgs[k] for k in MODALITIES])
        if self.pin and not whole.is_pinned():
            whole = whole.pin_memory()
        if self.stream is None:
            dev = whole.to(self.device)
            rest = tuple(v.to(self.device) if torch.is_tensor(v) else v for v in batch[1:])
            return ({k: dev[m] for m, k in enumerate(MODALITIES)},) + rest, None
        with torch.cuda.stream(self.stream):
            dev = whole.to(self.device, non_blocking=True)
            rest = tuple(v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v for v in batch[1:])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return ({k: dev[m] for m, k in enumerate(MODALITIES)},) + rest, (ev, dev, whole)

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, hold = nxt
            try:
                nxt = self._upload(next(it))          # queued on the side stream before the consumer starts on `cur`
            except StopIteration:
                nxt = None
            if hold is not None:
                consumer = torch.cuda.current_stream(self.device)
                consumer.wait_event(hold[0])
                hold[1].record_stream(consumer)
                # the small tensors (labels, camera / view ids) were allocated on the side stream too: without this the caching
                # allocator may hand their blocks to a later side-stream upload while a kernel of the consumer still reads them
                for v in cur[1:]:
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(consumer)
            yield cur

    def __len__(self):
        return len(self.loader)
