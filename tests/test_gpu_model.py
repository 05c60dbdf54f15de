"""GPU: the whole hot path (make_model(...).forward through the C ABI) against the CPU oracle on the same
seeded weights and inputs.

fp32 compute mode: tolerance 2e-4 -- "within fp32 tolerance" of the north star (identical algorithm, different
summation order).  bf16 compute mode (the benchmarked mode; the reference trains under fp16 autocast): features
are O(1) after BatchNorm/LayerNorm, tolerance 6e-2 absolute on bf16-rounded activations through 12 blocks;
mAP on a synthetic gallery must agree within 0.1 points (north star).
"""
import numpy as np
import pytest
import torch

from demo2_amd.modeling import make_model
from oracle import demo_oracle as O
from util import cpu_sd, make_cfg, small_cfg, spec_of, synth_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(cfg, classes, cams, seed=0, dtype=torch.float32):
    torch.manual_seed(seed)
    m = make_model(cfg, classes, cams, 0)
    sd = O.init_state_dict(spec_of(m), seed=seed + 1)
    m.load_state_dict(sd)
    m = m.to(DEV).eval().set_compute_dtype(dtype)
    return m, sd


@pytest.mark.parametrize("arch", ["clip", "vit"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_small_eval_all_patterns(arch, dtype, tol):
    m, sd = build(small_cfg(arch), 11, 3, dtype=dtype)
    spec = spec_of(m)
    x, cam = synth_batch(4, 64, 32, 3)
    xd = {k: v.to(DEV) for k, v in x.items()}
    for rp in (1, 2, 3):
        want = O.model_forward(sd, spec, x, cam, return_pattern=rp)
        with torch.no_grad():
            got = m(xd, cam_label=cam.to(DEV), return_pattern=rp)
        assert got.shape == want.shape and got.dtype == torch.float32
        torch.testing.assert_close(got.cpu(), want, rtol=tol, atol=tol)
    # cam_label inside the dict overrides the keyword (make_model.py:241-242)
    xd2 = dict(xd, cam_label=cam.to(DEV))
    with torch.no_grad():
        got = m(xd2, cam_label=torch.zeros_like(cam).to(DEV), return_pattern=3)
    torch.testing.assert_close(got.cpu(), O.model_forward(sd, spec, x, cam, return_pattern=3), rtol=tol, atol=tol)


@pytest.mark.parametrize("miss", ["r", "n", "t", "rn", "rt", "nt", "nothing"])
def test_missing_modality(miss):
    m, sd = build(small_cfg("clip", MISS=miss), 11, 3)
    spec = spec_of(m)
    assert spec.miss == miss
    x, cam = synth_batch(3, 64, 32, 3, seed=5)
    want = O.model_forward(sd, spec, x, cam, return_pattern=3)
    with torch.no_grad():
        got = m({k: v.to(DEV) for k, v in x.items()}, cam_label=cam.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("kw", [dict(HDM=True, ATM=False), dict(HDM=False, ATM=False), dict(GLOBAL_LOCAL=False), dict(SIE_CAMERA=False),
                                dict(head=2), dict(direct=0)])
def test_config_switches(kw):
    kw = dict(kw)
    head, direct = kw.pop("head", 4), kw.pop("direct", 1)
    m, sd = build(small_cfg("clip", head=head, direct=direct, **kw), 11, 3)
    spec = spec_of(m)
    x, cam = synth_batch(2, 64, 32, 3, seed=9)
    want = O.model_forward(sd, spec, x, cam, return_pattern=3)
    with torch.no_grad():
        got = m({k: v.to(DEV) for k, v in x.items()}, cam_label=cam.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=2e-4, atol=2e-4)


def test_single_sample_eval():
    """B=1: the reference squeezes to (C,) and re-expands (AttnMOE.py:95-102,181-189); output stays (1, D)."""
    m, sd = build(small_cfg("clip"), 11, 3)
    x, cam = synth_batch(1, 64, 32, 3, seed=2)
    want = O.model_forward(sd, spec_of(m), x, cam)
    with torch.no_grad():
        got = m({k: v.to(DEV) for k, v in x.items()}, cam_label=cam.to(DEV))
    assert got.shape == (1, 10 * 128)
    torch.testing.assert_close(got.cpu(), want, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_graph_replay_equals_eager_forward(dtype):
    """EvalGraph (hipGraph replay of the inference forward, small-batch serving): the eager forward's result for every batch fed through
    the captured graph (to the run-to-run spread of the split-K fp32 atomics: 1e-5), re-captured after a parameter update, same
    IndexError for an out-of-range camera id."""
    from demo2_amd.modeling.eval_graph import EvalGraph
    m, sd = build(small_cfg("clip"), 11, 3, dtype=dtype)
    g = EvalGraph(m, 2, 64, 32, return_pattern=3)
    for seed in (3, 4, 5):
        x, cam = synth_batch(2, 64, 32, 3, seed=seed)
        xd = {k: v.to(DEV) for k, v in x.items()}
        with torch.no_grad():
            want = m(xd, cam_label=cam.to(DEV), return_pattern=3)
        got = g(xd, cam_label=cam.to(DEV))
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    if dtype == torch.float32:
        torch.testing.assert_close(got.cpu(), O.model_forward(sd, spec_of(m), x, cam, return_pattern=3), rtol=2e-4, atol=2e-4)
    with torch.no_grad():                                   # a parameter update invalidates the capture
        next(p for n, p in m.named_parameters() if n.endswith("q_.weight")).mul_(1.5)
    with torch.no_grad():
        want = m(xd, cam_label=cam.to(DEV), return_pattern=3)
    assert (want - got).abs().max() > 1e-3
    torch.testing.assert_close(g(xd, cam_label=cam.to(DEV)), want, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        g({k: v[:1] for k, v in xd.items()}, cam_label=cam[:1].to(DEV))
    bad = cam.clone()
    bad[0] = 7                                              # 3 cameras
    g(xd, cam_label=bad.to(DEV))
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        g(xd, cam_label=cam.to(DEV))
    g(xd, cam_label=cam.to(DEV))                            # the report was consumed: the path works again


@pytest.mark.parametrize("name,classes,cams,size", [("RGBNT201", 171, 4, (256, 128)), ("RGBNT100", 50, 8, (128, 256))])
def test_full_size_eval_fp32_and_bf16(name, classes, cams, size):
    """BASELINE configs A and B at full model size (98.8 M parameters), B=2."""
    cfg = make_cfg(name)
    m, sd = build(cfg, classes, cams)
    spec = spec_of(m)
    x, cam = synth_batch(2, size[0], size[1], cams)
    want = O.model_forward(sd, spec, x, cam, return_pattern=3)
    xd = {k: v.to(DEV) for k, v in x.items()}
    with torch.no_grad():
        got = m(xd, cam_label=cam.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=5e-4, atol=5e-4)
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        got = m(xd, cam_label=cam.to(DEV))
    err = (got.cpu() - want).abs()
    assert err.max() < 0.15 and err.mean() < 0.02, (err.max(), err.mean())


def test_vit_b_full_size_eval():
    """Config V: ImageNet ViT-B trunk (C=768, HEAD=12, SIE on every token)."""
    cfg = make_cfg("RGBNT201", TRANSFORMER_TYPE="vit_base_patch16_224", HEAD=12, SIE_COE=3.0)
    m, sd = build(cfg, 171, 6)
    x, cam = synth_batch(2, 256, 128, 6)
    want = O.model_forward(sd, spec_of(m), x, cam)
    with torch.no_grad():
        got = m({k: v.to(DEV) for k, v in x.items()}, cam_label=cam.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=5e-4, atol=5e-4)


def _synthetic_gallery(ids, per, noise, size, seed=7):
    """ids x per tri-modal images = a fixed random prototype per identity + noise; cameras alternate inside an identity; shuffled."""
    g = torch.Generator().manual_seed(seed)
    protos = {k: torch.randn(ids, 3, size[0], size[1], generator=g) for k in ("RGB", "NI", "TI")}
    pid = np.repeat(np.arange(ids), per)
    camid = np.tile(np.arange(per) % 2, ids)
    x = {k: protos[k][pid] + noise * torch.randn(ids * per, 3, size[0], size[1], generator=g) for k in protos}
    order = np.random.RandomState(0).permutation(ids * per)
    return {k: v[order] for k, v in x.items()}, pid[order], camid[order]


def test_synthetic_map_parity():
    """north star: mAP / Rank-1 on a synthetic eval set within +-0.1 points of the reference CPU path.  SURVEY 8(d) sizes: 30 identities x 4
    images (2 cameras), 60 queries against 60 gallery items, market1501 protocol (same-pid-same-camera removal), full-size config A.
    The set is built NOT to be saturated or degenerate: with the seeded random weights the camera (SIE) embedding would dominate the cls
    token after ln_pre and every ranking would be 'same camera first' (mAP 3.8 %, Rank-1 0 at ANY noise level, measured on the oracle), so
    these synthetic weights carry a camera embedding scaled by 0.01; the noise level (1.8 x the prototype's std) puts the fp32 oracle at
    mAP ~ 70 %, Rank-1 ~ 60 % -- asserted to lie inside (20 %, 95 %).  The reference evaluates WITHOUT autocast (engine/processor.py:
    247-282), i.e. in fp32: the fp32 compute mode is the protocol-faithful one, the bf16 mode (what bench.py --mode fwd times) is held to
    the same +-0.1."""
    cfg = make_cfg("RGBNT201")
    torch.manual_seed(0)
    m = make_model(cfg, 171, 4, 0)
    spec = spec_of(m)
    sd = O.init_state_dict(spec, seed=1)
    sd["BACKBONE.cv_embed"] = sd["BACKBONE.cv_embed"] * 0.01
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    ids, per, nq = 30, 4, 60
    x, pid, camid = _synthetic_gallery(ids, per, 1.8, (256, 128))
    cam = torch.tensor(camid)
    feats = {"oracle": [], torch.float32: [], torch.bfloat16: [], "bf16+res": []}
    for i in range(0, ids * per, 8):
        xb = {k: v[i:i + 8] for k, v in x.items()}
        feats["oracle"].append(O.model_forward(sd, spec, xb, cam[i:i + 8]))
        xd = {k: v.to(DEV) for k, v in xb.items()}
        m.bf16_residual = False                                  # fp32 residual stream: fp32 mode, and bf16 products on an fp32 stream
        for dt in (torch.float32, torch.bfloat16):
            with torch.no_grad():
                feats[dt].append(m.set_compute_dtype(dt)(xd, cam_label=cam[i:i + 8].to(DEV)).cpu())
        m.bf16_residual = True                                   # the default bf16 mode: bf16 residual stream (adds in the GEMM epilogues)
        with torch.no_grad():
            feats["bf16+res"].append(m(xd, cam_label=cam[i:i + 8].to(DEV)).cpu())
    res = {k: O.r1_map(torch.cat(v), pid, camid, nq, max_rank=10) for k, v in feats.items()}
    (cmc_r, map_r), (cmc_f, map_f), (cmc_h, map_h), (cmc_s, map_s) = res["oracle"], res[torch.float32], res[torch.bfloat16], res["bf16+res"]
    fo, fh, fs = torch.cat(feats["oracle"]), torch.cat(feats[torch.bfloat16]), torch.cat(feats["bf16+res"])
    print(f"\nfeature error vs oracle: bf16 mode rel {float((fh - fo).norm() / fo.norm()):.4f} max {float((fh - fo).abs().max()):.4f};  "
          f"+ bf16 residual stream rel {float((fs - fo).norm() / fo.norm()):.4f} max {float((fs - fo).abs().max()):.4f};  "
          f"mAP / Rank-1 with the bf16 residual stream {100 * map_s:.2f} / {100 * cmc_s[0]:.2f}")
    print(f"\nsynthetic eval (30 ids x 4, nq 60): mAP / Rank-1 / Rank-5 [%]  oracle fp32 {100 * map_r:.2f} / {100 * cmc_r[0]:.2f} / {100 * cmc_r[4]:.2f}"
          f"   HIP fp32 {100 * map_f:.2f} / {100 * cmc_f[0]:.2f} / {100 * cmc_f[4]:.2f}   HIP bf16 {100 * map_h:.2f} / {100 * cmc_h[0]:.2f} / {100 * cmc_h[4]:.2f}")
    assert 0.20 < map_r < 0.95 and 0.20 < cmc_r[0] < 0.95, (map_r, cmc_r[0])          # neither saturated nor degenerate
    assert abs(map_r - map_f) * 100 <= 0.1 and abs(cmc_r[0] - cmc_f[0]) * 100 <= 0.1, (map_r, map_f, cmc_r[0], cmc_f[0])
    assert abs(map_r - map_h) * 100 <= 0.1 and abs(cmc_r[0] - cmc_h[0]) * 100 <= 0.1, (map_r, map_h, cmc_r[0], cmc_h[0])
    assert abs(map_r - map_s) * 100 <= 0.1 and abs(cmc_r[0] - cmc_s[0]) * 100 <= 0.1, (map_r, map_s, cmc_r[0], cmc_s[0])
    # the GPU evaluator on the GPU features gives the same figures as the numpy protocol on them
    from demo2_amd.utils.metrics import R1_mAP_eval
    ev = R1_mAP_eval(nq, max_rank=10, feat_norm="yes")
    ev.update((torch.cat(feats[torch.bfloat16]).to(DEV), pid.tolist(), camid.tolist(), [str(i) for i in range(ids * per)]))
    out = ev.compute()
    assert abs(out[1] - map_h) < 1e-5 and abs(float(out[0][0]) - float(cmc_h[0])) < 1e-6


def test_bench_size_properties_bf16():
    """BASELINE.json's sizes (config A, bf16, B = 64 eval / 128 train) are too big for the CPU oracle to finish in seconds, so
    they are checked through size-independent properties of the path:
      * samples are independent in eval (LayerNorm + running-stat BatchNorm): the features of a batch of 64 equal those of its
        two halves run separately (different tile/row-split decisions of the GEMM dispatcher: bf16 noise only);
      * a missing modality at eval (TEST.MISS) is exactly the same as feeding zero images for it (make_model.py:233-239);
      * the batch order does not matter: permuting the samples permutes the features."""
    cfg = make_cfg("RGBNT201")
    m, _ = build(cfg, 171, 4, dtype=torch.bfloat16)
    x, cam = synth_batch(64, 256, 128, 4, seed=11)
    xd, camd = {k: v.to(DEV) for k, v in x.items()}, cam.to(DEV)
    with torch.no_grad():
        full = m(xd, cam_label=camd).cpu()
        halves = torch.cat([m({k: v[:32] for k, v in xd.items()}, cam_label=camd[:32]).cpu(),
                            m({k: v[32:] for k, v in xd.items()}, cam_label=camd[32:]).cpu()])
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(3))
        shuffled = m({k: v[perm.to(DEV)] for k, v in xd.items()}, cam_label=camd[perm.to(DEV)]).cpu()
    assert full.shape == (64, 5120) and torch.isfinite(full).all()
    assert (full - halves).abs().max() < 2e-2 and (full - halves).norm() / full.norm() < 2e-3
    assert (full[perm] - shuffled).abs().max() < 2e-2
    m.miss_type = "n"
    with torch.no_grad():
        miss = m(xd, cam_label=camd).cpu()
    m.miss_type = "nothing"
    with torch.no_grad():
        zeroed = m(dict(xd, NI=torch.zeros_like(xd["NI"])), cam_label=camd).cpu()
    assert (miss - zeroed).abs().max() < 2e-2 and (miss - full).abs().max() > 0.1
