"""GPU, two processes on one card over gloo (RCCL needs one GPU per rank; the control flow and the flat-buffer slices are the same):
the in-backward gradient reduction of the REAL kernel schedule -- GradSync.tail_ready / block_ready / backward_done called from
hip_backward.trunk_backward -- leaves every rank with the mean of the two ranks' gradients, i.e. what a single process gets by
running both batches and averaging."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model_and_batch(rank_seed):
    from demo2_amd.modeling import make_model
    from oracle import demo_oracle as O
    from util import small_cfg, spec_of, synth_batch
    cfg = small_cfg("clip", head=4, direct=1)
    torch.manual_seed(0)
    m = make_model(cfg, 11, 3, 0)
    m.load_state_dict(O.init_state_dict(spec_of(m), seed=1))
    m = m.to("cuda").train().set_compute_dtype(torch.float32)
    m.geom.hdm_dropout = 0.0
    x, cam = synth_batch(8, 64, 32, 3, seed=500 + rank_seed, device="cuda")
    target = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], device="cuda")
    return cfg, m, x, cam, target


def _worker(rank, world, port, q, explicit=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from demo2_amd.dist import broadcast_parameters
        from demo2_amd.modeling.hip_backward import TrainStep
        cfg, m, x, cam, target = _model_and_batch(rank)
        broadcast_parameters(m)
        if explicit:
            ts = TrainStep(m, cfg, world, optimizer=False)           # installs GradSync (world > 1)
            assert m.grad_sync is not None
        else:
            # what a caller that only knows the reference's loop does: nothing (or a DistributedDataParallel wrapper, which sees no
            # parameter in the autograd graph and reduces nothing).  The training-mode forward must install the reduction itself.
            ts = TrainStep(m, cfg, 1, optimizer=False)
            assert m.grad_sync is None
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            loss = ts.step(x, target, cam)
        assert m.grad_sync is not None and (explicit or any("GradSync" in str(i.message) for i in w))
        torch.cuda.synchronize()
        q.put((rank, m.grad_buffer().detach().cpu().numpy(), float(loss)))     # by value: the worker exits right after
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("explicit", [True, False])
def test_two_rank_gradient_mean_through_the_backward_schedule(explicit):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, explicit)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g0, g1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(g0, g1)                                       # every rank holds the same reduced gradient
    # single process: both batches, no synchronisation, averaged by hand
    from demo2_amd.modeling.hip_backward import TrainStep
    want = None
    for r in range(2):
        cfg, m, x, cam, target = _model_and_batch(r)
        ts = TrainStep(m, cfg, 1, optimizer=False)
        loss = ts.step(x, target, cam)
        assert abs(float(loss) - res[r][2]) < 1e-4 * abs(res[r][2]) + 1e-5
        g = m.grad_buffer().detach().cpu()
        want = g if want is None else want + g
    want = want / 2
    err = ((g0 - want).norm() / want.norm()).item()
    assert err < 1e-4, err
    assert (g0 - want).abs().max().item() < 1e-4 * want.abs().max().item() + 1e-6
