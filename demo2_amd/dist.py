"""Data-parallel gradient reduction for the flat gradient buffer (reference: DDP in engine/processor.py:34-37).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm).  The reference wraps the model
in DistributedDataParallel (25 MB buckets discovered from autograd hooks, find_unused_parameters=True).  Here the
backward pass is an explicit kernel schedule, so the reduction is scheduled explicitly too:

  * every parameter gradient lives in ONE flat fp32 buffer laid out in forward order (make_model.py: FlatTable), so a
    bucket is a contiguous slice -- no gather/scatter copies, no per-parameter hooks;
  * the schedule calls `tail_ready()` once the fusion/head gradients are final and `block_ready(i)` after trunk block
    i's gradients are final (blocks finish 11 -> 0); each call launches an asynchronous all-reduce of the slices that
    became complete, which overlaps with the remaining backward kernels (RCCL runs on its own stream);
  * buckets are few and large (a group of trunk blocks ~ 85 MB): on the point-to-point xGMI mesh a ring step is bound by
    one ~153 GB/s link, so large messages amortise the per-collective latency;
  * `backward_done()` launches the last slice (embedding side + camera embedding) and makes the compute stream wait.
Gradients are averaged over ranks (DDP semantics).  BatchNorm statistics stay per rank, as in the reference
(no SyncBatchNorm).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model=None, *, flat_grad=None, block_ranges: Optional[Sequence[Tuple[int, int]]] = None,
                 tail_range: Optional[Tuple[int, int]] = None, last_ranges: Optional[Sequence[Tuple[int, int]]] = None,
                 blocks_per_bucket: int = 3, group=None):
        """Either pass `model` (a demo2_amd DeMo) or explicit element ranges over `flat_grad` (used by the CPU tests)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.blocks_per_bucket = max(1, blocks_per_bucket)
        self.handles: List = []
        self._pending: List[int] = []
        if model is not None:
            self._flat = model.grad_buffer
            block_ranges, tail_range, last_ranges = self.ranges_of(model)
        else:
            self._flat = (lambda: flat_grad)
        self.block_ranges = list(block_ranges)
        self.tail_range = tail_range
        self.last_ranges = list(last_ranges)
        ranges = sorted(self.block_ranges + [self.tail_range] + self.last_ranges)
        for (lo, hi), (lo2, _) in zip(ranges, ranges[1:]):
            if not (lo <= hi <= lo2):
                raise ValueError(f"GradSync: gradient ranges overlap or are out of order: ({lo}, {hi}) then ({lo2}, ...)")
        self.covered = sum(hi - lo for lo, hi in ranges)
        if model is not None:
            # the ranges are derived from the ORDER of named_parameters (embedding side, blocks 0..L-1, ln_post/proj, fusion + heads):
            # a model whose layout breaks that order would silently leave gradients unreduced
            n = model.ptab.flat.numel()
            if self.covered != n or ranges[0][0] != 0 or ranges[-1][1] != n:
                raise ValueError(f"GradSync: ranges cover {self.covered} of {n} gradient elements")
            if any(b[0] >= b[1] for b in self.block_ranges) or any(a[1] != b[0] for a, b in zip(self.block_ranges, self.block_ranges[1:])):
                raise ValueError("GradSync: trunk block ranges are not contiguous in layer order")
        self.reduced_elements = 0          # elements handed to a collective since construction (tests / diagnostics)
        self.force = False                 # issue the collectives even in a one-rank group (exercises the RCCL path on one GPU)

    @staticmethod
    def ranges_of(model):
        """Contiguous element ranges of the flat buffer: per trunk block, the fusion/head tail, and what completes last."""
        P, g = model.ptab, model.geom
        n = P.flat.numel()
        first = [P.off(g.block_keys(i)["ln1"] + ".weight") for i in range(g.layers)]
        starts = []
        for i in range(g.layers):
            ks = g.block_keys(i)
            starts.append(min(P.off(v if not k.startswith("ln") else v + ".weight") for k, v in ks.items()))
        post = P.off(g.k_lnpost + ".weight")
        ends = starts[1:] + [post]
        blocks = list(zip(starts, ends))
        tail_start = P.off("rgb_reduce.0.weight")
        del first
        return blocks, (tail_start, n), [(0, starts[0]), (post, tail_start)]

    # ------------------------------------------------------------------ collective launch
    def _reduce(self, lo: int, hi: int):
        if hi <= lo or (self.world == 1 and not self.force):
            return
        view = self._flat()[lo:hi]
        self.reduced_elements += hi - lo
        if view.is_cuda and dist.get_backend(self.group) == "nccl":
            h = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:                                     # gloo has no AVG (CPU tests; GPU rehearsals of the N > 1 flow on one card)
            view.div_(self.world)
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append(h)

    def tail_ready(self):
        self._reduce(*self.tail_range)

    def block_ready(self, i: int):
        """Blocks complete in descending order; a bucket is launched when `blocks_per_bucket` adjacent blocks are done."""
        self._pending.append(i)
        if len(self._pending) == self.blocks_per_bucket or i == 0:
            lo = min(self.block_ranges[j][0] for j in self._pending)
            hi = max(self.block_ranges[j][1] for j in self._pending)
            self._pending = []
            self._reduce(lo, hi)

    def backward_done(self):
        if self._pending:
            self.block_ready(0)
        for lo, hi in self.last_ranges:
            self._reduce(lo, hi)
        for h in self.handles:
            h.wait()                              # stream-level wait on CUDA/HIP tensors
        self.handles = []


def broadcast_parameters(model, src: int = 0, group=None, force: bool = False):
    """One broadcast of the flat parameter and statistics buffers at start-up (DDP's initial parameter broadcast).
    force: issue the collectives in a one-rank group too (exercises the RCCL path on one GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    dist.broadcast(model.ptab.flat, src, group=group)
    dist.broadcast(model.stab.flat, src, group=group)
    model.ptab.shadow_version = -1


def ensure_grad_sync(model):
    """Called by the training-mode forward.  In a multi-rank job every parameter gradient is written by kernels straight into the flat
    buffer behind ONE autograd node, so torch's DistributedDataParallel (engine/processor.py:34-37 wraps the model with
    find_unused_parameters=True) sees no parameter in the graph, marks all of them unused and reduces NOTHING -- silently.  The
    reduction therefore belongs to the model: when a process group with more than one rank exists and the caller installed no GradSync,
    one is installed here (it works with or without the DDP wrapper around the model: DDP leaves globally unused gradients alone).
    `model.grad_sync = False` opts out explicitly (per-rank gradients wanted)."""
    if model.grad_sync is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        model.grad_sync = GradSync(model)
        import warnings
        warnings.warn("demo2_amd: torch.distributed is initialised with %d ranks and no GradSync was installed; installing "
                      "demo2_amd.dist.GradSync(model) (gradient mean over ranks inside the backward schedule).  A "
                      "DistributedDataParallel wrapper around this model reduces nothing by itself." % dist.get_world_size())
    return model.grad_sync if model.grad_sync else None
