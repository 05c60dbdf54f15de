"""GPU: the retrieval evaluator (csrc/evalrank.hip + demo2_amd/utils/metrics.py) against the reference-generated golden
CMC / mAP, against the oracle's numpy restatement of eval_func on synthetic galleries, and on its size-independent
properties (the order is sorted and a permutation; AP of a perfect / worst ranking)."""
import numpy as np
import pytest
import torch

from demo2_amd import ops
from demo2_amd.utils.metrics import R1_mAP, R1_mAP_eval, eval_func, eval_func_msrv, euclidean_distance
from oracle import demo_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_golden_evalfunc(golden):
    g = golden("evalfunc")
    nq = int(g["nq"])
    ev = R1_mAP_eval(nq, max_rank=10, feat_norm=True)
    ev.update((g["feats"].to(DEV), g["pids"].numpy(), g["camids"].numpy(), None))
    cmc, mAP, distmat, *_ = ev.compute()
    np.testing.assert_allclose(cmc, g["cmc"].numpy()[:len(cmc)], rtol=1e-6)
    assert abs(mAP - float(g["mAP"])) < 1e-6


def synth(nq, ng, D, ids, cams, seed):
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(ids, D)).astype(np.float32)
    pids = rng.integers(0, ids, nq + ng)
    feats = centers[pids] + 0.8 * rng.normal(size=(nq + ng, D)).astype(np.float32)
    camids = rng.integers(0, cams, nq + ng)
    return torch.from_numpy(feats), pids, camids


@pytest.mark.parametrize("nq,ng,D,ids,cams,max_rank", [(17, 53, 64, 6, 3, 10), (200, 836, 512, 40, 4, 50), (64, 5000, 256, 50, 6, 50),
                                                      (8, 16384, 32, 30, 4, 50)])
def test_eval_func_matches_oracle(nq, ng, D, ids, cams, max_rank):
    """(max_rank is kept below the smallest kept-gallery size: the reference itself fails on ragged CMC rows.)"""
    feats, pids, camids = synth(nq, ng, D, ids, cams, seed=ng)
    cmc_ref, map_ref = O.r1_map(feats, pids, camids, nq, feat_norm=True, max_rank=max_rank)
    ev = R1_mAP_eval(nq, max_rank=max_rank, feat_norm=True)
    for lo in range(0, nq + ng, 1000):                       # batches, as do_inference feeds them
        hi = min(nq + ng, lo + 1000)
        ev.update((feats[lo:hi].to(DEV), pids[lo:hi], camids[lo:hi], None))
    cmc, mAP, distmat, *_ = ev.compute()
    assert cmc.dtype == np.float32 and cmc.shape == cmc_ref.shape
    np.testing.assert_allclose(cmc, cmc_ref, rtol=0, atol=1e-6)
    assert abs(mAP - map_ref) < 1e-6                        # the +-0.1 mAP claim of BASELINE.json, with a lot of room
    # distance matrix itself against the reference formula
    f = torch.nn.functional.normalize(feats, dim=1)
    ref = (f[:nq] ** 2).sum(1, keepdim=True) + (f[nq:] ** 2).sum(1, keepdim=True).t() - 2 * f[:nq] @ f[nq:].t()
    torch.testing.assert_close(distmat.cpu(), ref, rtol=1e-4, atol=2e-5)


def test_msvr310_scene_protocol():
    feats, pids, camids = synth(40, 300, 64, 12, 8, seed=5)
    scenes = np.random.default_rng(9).integers(0, 3, 340)
    f = torch.nn.functional.normalize(feats, dim=1)
    dist = (f[:40] ** 2).sum(1, keepdim=True) + (f[40:] ** 2).sum(1, keepdim=True).t() - 2 * f[:40] @ f[40:].t()
    # eval_func_msrv = eval_func with the scene id in the role of the camera id (utils/metrics.py:67 vs :140)
    cmc_ref, map_ref = O.eval_market1501(dist.numpy(), pids[:40], pids[40:], scenes[:40], scenes[40:], 50)
    cmc, mAP = eval_func_msrv(dist.to(DEV), pids[:40], pids[40:], camids[:40], camids[40:], scenes[:40], scenes[40:])
    np.testing.assert_allclose(cmc, cmc_ref, atol=1e-6)
    assert abs(mAP - map_ref) < 1e-6
    ev = R1_mAP(40, feat_norm="yes")
    ev.update((feats.to(DEV), pids, camids, scenes, None))
    cmc2, mAP2, *_ = ev.compute()
    np.testing.assert_allclose(cmc2, cmc_ref, atol=1e-6)
    assert abs(mAP2 - map_ref) < 1e-6


def test_msvr310_protocol_matches_reference_golden(golden):
    """The GPU evaluator's R1_mAP / eval_func_msrv against the fixture the reference's own R1_mAP.compute produced
    (tests/golden/evalfunc_msrv.npz, oracle/gen_golden.py: gen_eval_msrv)."""
    g = golden("evalfunc_msrv")
    nq = int(g["nq"])
    ev = R1_mAP(nq, max_rank=50, feat_norm="yes")
    ev.update((g["feats"].to(DEV), g["pids"].numpy(), g["camids"].numpy(), g["sceneids"].numpy(), None))
    cmc, mAP, *_ = ev.compute()
    np.testing.assert_allclose(cmc, g["cmc"].numpy(), atol=1e-6)
    assert abs(mAP - float(g["mAP"])) < 1e-6


def test_rank_eval_properties():
    """Bit-exact index work: the returned order is a permutation that sorts each row (ties by index), AP = 1 when every
    match precedes every non-match, and queries whose identity is absent are flagged."""
    torch.manual_seed(0)
    nq, ng = 9, 1000
    dist = torch.rand(nq, ng, device=DEV)
    dist[:, 10:20] = dist[:, 0:10]                          # exact ties
    g_pids = torch.arange(ng, device=DEV) % 50
    g_cam = torch.zeros(ng, dtype=torch.int64, device=DEV)
    q_pids = torch.arange(nq, device=DEV)
    q_pids[-1] = 777                                        # absent identity
    q_cam = torch.ones(nq, dtype=torch.int64, device=DEV)
    ap, first, order = ops.rank_eval(dist, q_pids, g_pids, q_cam, g_cam, want_order=True)
    o = order.cpu().long()
    assert torch.equal(o.sort(1).values, torch.arange(ng).expand(nq, ng))
    ref_order = torch.from_numpy(np.argsort(dist.cpu().numpy(), axis=1, kind="stable"))
    assert torch.equal(o, ref_order)
    assert ap[-1].item() == -1.0 and first[-1].item() == -1
    # perfect ranking: matches get the smallest distances
    d2 = torch.rand(nq, ng, device=DEV) + 1.0
    d2[g_pids[None, :] == q_pids[:, None]] -= 1.0
    ap2, first2 = ops.rank_eval(d2, q_pids, g_pids, q_cam, g_cam)
    assert torch.all(ap2[:-1] == 1.0) and torch.all(first2[:-1] == 0)


def test_eval_func_accepts_numpy_distmat():
    feats, pids, camids = synth(10, 40, 16, 5, 2, seed=3)
    dist = euclidean_distance(feats[:10].to(DEV), feats[10:].to(DEV))
    a = eval_func(dist, pids[:10], pids[10:], camids[:10], camids[10:], 20)
    b = eval_func(dist.cpu().numpy(), pids[:10], pids[10:], camids[:10], camids[10:], 20)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1] == b[1]
