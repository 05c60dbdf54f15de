"""CPU, world_size 2 over gloo: the bucketed gradient reduction that bench.py --gpus N / TrainStep use over RCCL.
The flat-buffer slices, bucket grouping and completion order are identical on GPU; only the backend differs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from demo2_amd.dist import GradSync, broadcast_parameters
from demo2_amd.modeling import make_model
from util import small_cfg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = make_model(small_cfg(), 11, 3, 0)              # CPU: parameters only, no kernels
        if rank == 1:
            with torch.no_grad():
                model.ptab.flat.add_(1.0)
        broadcast_parameters(model)
        psum = model.ptab.flat.double().sum().item()
        g = model.grad_buffer()
        gen = torch.Generator().manual_seed(100 + rank)
        g.copy_(torch.randn(g.numel(), generator=gen))
        mine = g.clone()
        other = torch.randn(g.numel(), generator=torch.Generator().manual_seed(100 + (1 - rank)))
        sync = GradSync(model, blocks_per_bucket=1)
        blocks, tail, last = GradSync.ranges_of(model)
        covered = torch.zeros(g.numel(), dtype=torch.bool)
        for lo, hi in blocks + [tail] + last:
            assert not covered[lo:hi].any()                    # disjoint
            covered[lo:hi] = True
        assert covered.all()                                   # the buckets tile the whole flat buffer
        sync.tail_ready()                                      # schedule order of trunk_backward
        for i in reversed(range(model.geom.layers)):
            sync.block_ready(i)
        sync.backward_done()
        want = (mine + other) / 2
        ok = torch.allclose(g, want, rtol=1e-6, atol=1e-6)
        # grouped buckets give the same result
        g.copy_(mine)
        sync2 = GradSync(model, blocks_per_bucket=3)
        sync2.tail_ready()
        for i in reversed(range(model.geom.layers)):
            sync2.block_ready(i)
        sync2.backward_done()
        ok2 = torch.allclose(g, want, rtol=1e-6, atol=1e-6)
        q.put((rank, ok, ok2, psum))
    finally:
        dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and ok2 for _, ok, ok2, _ in res), res
    assert abs(res[0][3] - res[1][3]) < 1e-9                  # parameters broadcast from rank 0


def test_gradsync_single_process_is_noop():
    g = torch.randn(1000)
    keep = g.clone()
    s = GradSync(flat_grad=g, block_ranges=[(100, 400), (400, 700)], tail_range=(700, 1000), last_ranges=[(0, 100)])
    s.tail_ready()
    s.block_ready(1)
    s.block_ready(0)
    s.backward_done()
    assert torch.equal(g, keep) and s.covered == 1000
