"""GPU: the RCCL ("nccl" backend on ROCm) code path of the data-parallel pieces, EXECUTED on one GPU.

No multi-GPU node is available to the build container or to `gpurun`, so no scaling curve can be measured here; what can be done is to run
every collective call the N > 1 job issues -- ReduceOp.AVG all-reduce with async_op on slices of the flat gradient buffer launched from the
real backward schedule, the handles' stream-level wait, the start-up broadcast of the flat buffers, the sampler's seed broadcast from a
device tensor -- through a real one-rank `nccl` process group (GradSync.force / broadcast_parameters(force=True) lift the world == 1
early return).  Runs in a child process so that the process group never leaks into the test session."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    out = {}
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        out["backend"] = dist.get_backend()
        from demo2_amd.data.sampler import _shared_seed
        from demo2_amd.dist import GradSync, broadcast_parameters
        from demo2_amd.modeling import make_model
        from demo2_amd.modeling.hip_backward import TrainStep
        from oracle import demo_oracle as O
        from util import small_cfg, spec_of, synth_batch
        cfg = small_cfg("clip", head=4, direct=1, TRUNK_LAYERS=4)
        torch.manual_seed(0)
        m = make_model(cfg, 11, 3, 0)
        m.load_state_dict(O.init_state_dict(spec_of(m), seed=1))
        m = m.to("cuda").train().set_compute_dtype(torch.float32)
        m.geom.hdm_dropout = 0.0
        x, cam = synth_batch(8, 64, 32, 3, device="cuda")
        target = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], device="cuda")
        # (1) start-up broadcast of the flat parameter / statistics buffers
        before = m.ptab.flat.clone()
        broadcast_parameters(m, force=True)
        torch.cuda.synchronize()
        out["broadcast_unchanged"] = bool(torch.equal(before, m.ptab.flat))
        # (2) gradients without any synchronisation
        ts = TrainStep(m, cfg, 1, optimizer=False)
        loss0 = float(ts.step(x, target, cam))
        g0 = m.grad_buffer().clone()
        m.attach_grads(zero=True)
        # (3) the same step with the in-backward reduction forced through RCCL: AVG over one rank must change nothing
        sync = GradSync(m, blocks_per_bucket=2)
        sync.force = True
        m.grad_sync = sync
        seen = []
        orig = sync._reduce

        def spy(lo, hi):
            n = len(sync.handles)
            orig(lo, hi)
            if len(sync.handles) > n:
                seen.append((lo, hi, type(sync.handles[-1]).__name__))
        sync._reduce = spy
        loss1 = float(ts.step(x, target, cam))
        torch.cuda.synchronize()
        g1 = m.grad_buffer().clone()
        # (two separate backward passes: fp32 atomics make them equal up to summation order, not bit for bit)
        out.update(loss0=loss0, loss1=loss1, grads_equal=bool(((g0 - g1).norm() / g0.norm()) < 1e-5), max_diff=float((g0 - g1).abs().max()),
                   reduced=sync.reduced_elements, numel=g0.numel(), n_collectives=len(seen), handles_left=len(sync.handles),
                   finite=bool(torch.isfinite(g1).all()), handle_type=seen[0][2] if seen else "")
        # buckets: tail first, then trunk blocks in descending order, then the embedding side
        out["order_ok"] = bool(seen and seen[0][0] == sync.tail_range[0] and all(a[0] > b[0] for a, b in zip(seen[1:3], seen[2:4])))
        # (4) a second accumulation pass: 2x the gradient (the reduction runs on the accumulated buffer, so AVG over 1 rank keeps it)
        ts.step(x, target, cam)
        torch.cuda.synchronize()
        g2 = m.grad_buffer()
        out["accumulates"] = float(((g2 - 2 * g1).norm() / (2 * g1).norm()))
        # (5) the data-parallel sampler's seed agreement, broadcast from a device tensor
        out["seed"] = _shared_seed(123456789, force=True)
    except Exception as e:                                            # noqa: BLE001 -- reported to the parent
        import traceback
        out["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    q.put(out)


def test_rccl_calls_of_the_data_parallel_path_execute_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    r = q.get(timeout=600)
    p.join(timeout=120)
    assert "error" not in r, r["error"]
    assert r["backend"] == "nccl"
    assert r["broadcast_unchanged"]
    assert r["finite"] and r["grads_equal"], r                     # AVG over a one-rank group is the identity
    assert abs(r["loss0"] - r["loss1"]) < 1e-6 * abs(r["loss0"]) + 1e-7
    assert r["reduced"] == r["numel"] and r["handles_left"] == 0    # every gradient element went through a collective; all waited
    assert r["n_collectives"] == 1 + 2 + 2 and r["order_ok"], r     # tail, 4 blocks in buckets of 2, the two embedding-side slices
    assert r["accumulates"] < 1e-5
    assert r["seed"] == 123456789
    assert p.exitcode == 0
