"""P x K identity-balanced sampling (reference: data/datasets/sampler.py:8-67, sampler_ddp.py:117-201).

Contract kept from the reference (the triplet loss mines inside the local batch, layers/triplet_loss.py:121-135, so it matters):
  * a batch is `batch_size // num_instances` identities x `num_instances` samples each, identities never split across batches;
  * an identity with fewer than K samples is filled by drawing WITH replacement (sampler.py:44-45);
  * every identity contributes floor(max(n, K) / K) groups per epoch, so len(sampler) = sum_id (max(n,K) - max(n,K) % K)
    (sampler.py:30-37); the epoch ends when fewer than P identities still have a group (sampler.py:57-63);
  * data-parallel: every rank derives the SAME global index list from one shared seed and keeps its own mini-batch blocks
    (block b of rank r = global positions [(b*world + r)*mb, (b*world + r + 1)*mb)), so a rank's batch holds whole identities
    (sampler_ddp.py:164-174) and len is divided by the world size (sampler_ddp.py:152).

Two modes:
  * default -- draws come from a private numpy Generator (seedable, no global `random` / `np.random` state), the index lists are
    built with array operations instead of per-sample Python loops, and the shared seed travels as ONE int64 broadcast instead of
    a pickled all_gather over a second gloo group.  The index SEQUENCE differs from the reference's for the same seed; its
    distribution and every property above are the same (tests/test_data_cpu.py);
  * `legacy_rng=True` -- the reference's index sequence BIT FOR BIT: the draws are taken from the GLOBAL `random` / `np.random`
    states that train_net.py:18-30 seeds with SOLVER.SEED, by the same library calls in the same order (sampler.py:39-64:
    np.random.choice for the padding, random.shuffle per identity, random.sample per batch; sampler_ddp.py:153-197:
    np.random.randint for the seed draw, np.random.seed, np.random.choice / np.random.shuffle inside the batch loop), so a run on
    real RGBNT201 data sees exactly the reference's batches.  Pinned by tests/golden/sampler_legacy.npz, written by the
    reference's own sampler classes (oracle/gen_golden.py: gen_sampler)."""
from __future__ import annotations

import math
import random
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


def _legacy_order(index_dic: Dict[int, np.ndarray], P: int, K: int) -> List[int]:
    """Single-process epoch with the reference's draw sequence on the global `random` / `np.random` states (sampler.py:39-64):
    per identity in first-seen order -- np.random.choice(., K, replace=True) when it has fewer than K samples, then random.shuffle,
    then cut into K-groups; per batch -- random.sample(available identities, P), first remaining group of each, an identity leaves
    the pool (list.remove: order of the rest kept) when its groups are used up."""
    groups: Dict[int, List[List[int]]] = {}
    for pid, idxs in index_dic.items():
        pool = list(idxs.tolist())
        if len(pool) < K:
            pool = np.random.choice(pool, size=K, replace=True)
        random.shuffle(pool)                                     # (on the list, or on the ndarray np.random.choice returned)
        pool = [int(v) for v in pool]
        groups[pid] = [pool[i:i + K] for i in range(0, len(pool) - len(pool) % K, K)]
    avail = list(index_dic)
    out: List[int] = []
    while len(avail) >= P:
        for pid in random.sample(avail, P):
            out.extend(groups[pid].pop(0))
            if not groups[pid]:
                avail.remove(pid)
    return out


def _legacy_global_order(index_dic: Dict[int, np.ndarray], P: int, K: int) -> List[int]:
    """Data-parallel epoch with the reference's draw sequence on the (already seeded) global np.random state
    (sampler_ddp.py:176-197): per batch np.random.choice(available identities, P, replace=False); an identity's samples are drawn
    (np.random.choice with replacement below K) and np.random.shuffle'd the FIRST time it is selected; it gives its next K samples and
    leaves the pool when fewer than K remain."""
    avail = list(index_dic)
    pools: Dict[int, List[int]] = {}
    out: List[int] = []
    while len(avail) >= P:
        for pid in np.random.choice(avail, P, replace=False).tolist():
            if pid not in pools:
                pool = index_dic[pid].tolist()
                if len(pool) < K:
                    pool = np.random.choice(pool, size=K, replace=True).tolist()
                np.random.shuffle(pool)
                pools[pid] = pool
            out.extend(pools[pid][:K])
            del pools[pid][:K]
            if len(pools[pid]) < K:
                avail.remove(pid)
    return out


class RandomIdentitySampler(Sampler):
    """Single-process P x K sampler; constructor arguments as data/datasets/sampler.py:18 (+ an optional seed, + legacy_rng: the
    reference's exact index sequence from the global `random` / `np.random` states, see the module docstring)."""

    def __init__(self, data_source, batch_size, num_instances, seed: Optional[int] = None, legacy_rng: bool = False):
        if batch_size % num_instances:
            raise ValueError(f"batch_size {batch_size} is not a multiple of num_instances {num_instances}")
        self.data_source, self.batch_size, self.num_instances = data_source, batch_size, num_instances
        self.num_pids_per_batch = batch_size // num_instances
        self.index_dic = _index_by_pid(data_source)
        self.pids = list(self.index_dic)
        self.length = epoch_length(self.index_dic, num_instances)
        self.legacy_rng = bool(legacy_rng)
        if legacy_rng and seed is not None:
            raise ValueError("legacy_rng draws from the global random / np.random states (seed them as train_net.py:18-30 does); "
                             "a private seed has no meaning there")
        self._rng = np.random.default_rng(seed)

    def __iter__(self):
        if self.legacy_rng:
            return iter(_legacy_order(self.index_dic, self.num_pids_per_batch, self.num_instances))
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
                 seed: Optional[int] = None, legacy_rng: bool = False):
        on = dist.is_available() and dist.is_initialized()
        self.legacy_rng = bool(legacy_rng)
        if legacy_rng and seed is not None:
            raise ValueError("legacy_rng takes its epoch seed from np.random.randint on the global state (sampler_ddp.py:103-114,154)")
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

    def legacy_local_part(self, order: Sequence[int]) -> List[int]:
        """The reference's block selection (sampler_ddp.py:153-174): ceil(total / world) // mini_batch blocks per rank, block b of rank r
        at global positions [(b * world + r) * mb, ...) truncated at the end of the list."""
        mb, w, total = self.mini_batch_size, self.world_size, len(order)
        block_num = int(math.ceil(total * 1.0 / w)) // mb
        mine: List[int] = []
        for i in range(0, block_num * w, w):
            mine.extend(order[mb * self.rank + mb * i:min(mb * self.rank + mb * (i + 1), total)])
        return mine

    def __iter__(self):
        if self.legacy_rng:
            # every rank draws (and so advances its global state, as in the reference); rank 0's value is the epoch seed
            seed = _shared_seed(int(np.random.randint(2 ** 31)))
            np.random.seed(seed)
            self._seed_used = int(seed)
            mine = self.legacy_local_part(_legacy_global_order(self.index_dic, self.num_pids_per_batch, self.num_instances))
            self.length = len(mine)
            return iter(mine)
        seed = _shared_seed(None if self._seed is None else self._seed + self._epoch)
        self._epoch += 1
        mine = self.local_part(self.global_order(seed))
        self.length = len(mine)
        return iter(mine.tolist())

    def __len__(self):
        return self.length
