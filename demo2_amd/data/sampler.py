"""P x K identity-balanced sampling (reference: data/datasets/sampler.py:8-67, sampler_ddp.py:117-201).

Contract kept from the reference (the triplet loss mines inside the local batch, layers/triplet_loss.py:121-135, so it matters):
  * a batch is `batch_size // num_instances` identities x `num_instances` samples each, identities never split across batches;
  * an identity with fewer than K samples is filled by drawing WITH replacement (sampler.py:44-45);
  * every identity contributes floor(max(n, K) / K) groups per epoch, so len(sampler) = sum_id (max(n,K) - max(n,K) % K)
    (sampler.py:30-37); the epoch ends when fewer than P identities still have a group (sampler.py:57-63);
  * data-parallel: every rank derives the SAME global index list from one shared seed and keeps its own mini-batch blocks
    (block b of rank r = global positions [(b*world + r)*mb, (b*world + r + 1)*mb)), so a rank's batch holds whole identities
    (sampler_ddp.py:164-174) and len is divided by the world size (sampler_ddp.py:152).

Different by design: draws come from a private numpy Generator (seedable, no global `random` / `np.random` state), the index
lists are built with array operations instead of per-sample Python loops, and the shared seed travels as ONE int64 broadcast
instead of a pickled all_gather over a second gloo group.  The index SEQUENCE therefore differs from the reference's for the
same seed; its distribution and every property above are the same (tests/test_data_cpu.py)."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


def _index_by_pid(data_source: Sequence) -> Dict[int, np.ndarray]:
    """data_source items are (img_path, pid, camid, viewid) (data/datasets/bases.py); -> pid -> sample indices."""
    d = defaultdict(list)
    for i, item in enumerate(data_source):
        d[item[1]].append(i)
    return {pid: np.asarray(v, dtype=np.int64) for pid, v in d.items()}


def epoch_length(index_dic: Dict[int, np.ndarray], K: int) -> int:
    return int(sum(max(len(v), K) - max(len(v), K) % K for v in index_dic.values()))


def _groups(index_dic: Dict[int, np.ndarray], K: int, rng: np.random.Generator) -> Dict[int, List[np.ndarray]]:
    """pid -> list of K-sample groups: a shuffled copy of the identity's samples (padded by replacement draws up to K) cut
    into floor(n/K) groups."""
    out = {}
    for pid, idxs in index_dic.items():
        pool = rng.choice(idxs, size=K, replace=True) if len(idxs) < K else rng.permutation(idxs)
        n = len(pool) // K
        out[pid] = list(pool[:n * K].reshape(n, K))
    return out


def _pk_order(index_dic: Dict[int, np.ndarray], P: int, K: int, rng: np.random.Generator) -> np.ndarray:
    """One epoch as a flat index array: repeatedly pick P distinct identities that still have a group, emit one group of each."""
    groups = _groups(index_dic, K, rng)
    avail = [pid for pid, g in groups.items() if g]
    batches = []
    while len(avail) >= P:
        pick = rng.choice(len(avail), size=P, replace=False)
        chosen = [avail[i] for i in pick]
        batches.append(np.concatenate([groups[pid].pop() for pid in chosen]))
        gone = {pid for pid in chosen if not groups[pid]}
        if gone:
            avail = [pid for pid in avail if pid not in gone]
    return np.concatenate(batches) if batches else np.zeros(0, dtype=np.int64)


class RandomIdentitySampler(Sampler):
    """Single-process P x K sampler; constructor arguments as data/datasets/sampler.py:18 (+ an optional seed)."""

    def __init__(self, data_source, batch_size, num_instances, seed: Optional[int] = None):
        if batch_size % num_instances:
            raise ValueError(f"batch_size {batch_size} is not a multiple of num_instances {num_instances}")
        self.data_source, self.batch_size, self.num_instances = data_source, batch_size, num_instances
        self.num_pids_per_batch = batch_size // num_instances
        self.index_dic = _index_by_pid(data_source)
        self.pids = list(self.index_dic)
        self.length = epoch_length(self.index_dic, num_instances)
        self._rng = np.random.default_rng(seed)

    def __iter__(self):
        order = _pk_order(self.index_dic, self.num_pids_per_batch, self.num_instances, self._rng)
        return iter(order.tolist())

    def __len__(self):
        return self.length


def _shared_seed(seed: Optional[int] = None, force: bool = False) -> int:
    """One seed for all ranks (sampler_ddp.py:103-114): rank 0's draw, broadcast as a single int64.
    force: broadcast in a one-rank group too (exercises the RCCL path on one GPU)."""
    s = int(np.random.SeedSequence().generate_state(1)[0]) if seed is None else int(seed)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([s], dtype=torch.int64, device=dev)
        dist.broadcast(t, 0)
        s = int(t.item())
    return s


class RandomIdentitySampler_DDP(Sampler):
    """Data-parallel P x K sampler; `batch_size` is the GLOBAL batch (cfg.SOLVER.IMS_PER_BATCH), each rank yields mini-batches
    of batch_size // world (sampler_ddp.py:127-133).  Use it under torch.utils.data.BatchSampler(sampler, mini_batch, True) as
    make_dataloader.py:215-218 does.  rank / world default to the initialised process group."""

    def __init__(self, data_source, batch_size, num_instances, rank: Optional[int] = None, world_size: Optional[int] = None,
                 seed: Optional[int] = None):
        on = dist.is_available() and dist.is_initialized()
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if on else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if on else 0)
        self.data_source, self.batch_size, self.num_instances = data_source, batch_size, num_instances
        self.mini_batch_size = batch_size // self.world_size
        if self.mini_batch_size % num_instances or self.mini_batch_size == 0:
            raise ValueError(f"per-rank batch {self.mini_batch_size} is not a positive multiple of num_instances {num_instances}")
        self.num_pids_per_batch = self.mini_batch_size // num_instances
        self.index_dic = _index_by_pid(data_source)
        self.pids = list(self.index_dic)
        self.length = epoch_length(self.index_dic, num_instances) // self.world_size
        self._seed, self._epoch = seed, 0

    def global_order(self, seed: int) -> np.ndarray:
        """The epoch's global list: consecutive mini-batch blocks, each P x K with whole identities (identical on every rank)."""
        return _pk_order(self.index_dic, self.num_pids_per_batch, self.num_instances, np.random.default_rng(seed))

    def local_part(self, order: np.ndarray) -> np.ndarray:
        mb, w = self.mini_batch_size, self.world_size
        nblocks = len(order) // (mb * w)                         # whole rounds only: every rank gets the same count
        blocks = order[:nblocks * mb * w].reshape(nblocks, w, mb)
        return blocks[:, self.rank].reshape(-1)

    def __iter__(self):
        seed = _shared_seed(None if self._seed is None else self._seed + self._epoch)
        self._epoch += 1
        mine = self.local_part(self.global_order(seed))
        self.length = len(mine)
        return iter(mine.tolist())

    def __len__(self):
        return self.length
