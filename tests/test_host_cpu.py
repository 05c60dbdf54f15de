"""CPU: host-side logic -- config shim, state-dict surface, flat parameter storage, C-ABI exports.
No kernel is launched here."""
import ctypes
import os

import pytest
import torch

from demo2_amd import _lib
from demo2_amd.config import cfg as default_cfg
from demo2_amd.modeling import make_model
from oracle import demo_oracle as O
from util import make_cfg, small_cfg, spec_of


def test_cfg_surface():
    c = default_cfg.clone()
    c.merge_from_file(os.path.join(os.path.dirname(__file__), "..", "demo2_amd", "configs", "RGBNT100", "DeMo.yml"))
    c.merge_from_list(["MODEL.HEAD", "4", "TEST.MISS", "rn", "SOLVER.BASE_LR", "0.001"])
    c.freeze()
    assert (c.MODEL.HEAD, c.TEST.MISS, c.MODEL.DIRECT, c.INPUT.SIZE_TRAIN, c.SOLVER.BASE_LR) == (4, "rn", 0, [128, 256], 0.001)
    assert c.MODEL.get("ARCH", "x") == "DeMo"
    with pytest.raises(AttributeError):
        c.MODEL.HEAD = 2
    with pytest.raises(KeyError):
        default_cfg.clone().merge_from_list(["MODEL.NOPE", "1"])


@pytest.mark.parametrize("name,classes,cams", [("RGBNT201", 171, 4), ("RGBNT100", 50, 8)])
def test_state_dict_matches_reference_names(name, classes, cams):
    c = make_cfg(name)
    m = make_model(c, classes, cams, 0)
    spec = spec_of(m)
    want = O.init_state_dict(spec)
    got = m.state_dict()
    assert set(got) == set(want)
    assert all(tuple(got[k].shape) == tuple(want[k].shape) for k in want)
    n = sum(p.numel() for p in m.parameters())
    if name == "RGBNT201":
        assert abs(n / 1e6 - 98.79) < 0.05                                    # SURVEY 8(a) / results/Ablation.png row E
    # SURVEY 8(a) note: HDM 7,357,952 and MoM 2,830,336 parameters at C=512, H=4
    if name == "RGBNT201":
        hdm = sum(v.numel() for k, v in m.named_parameters() if k.startswith("generalFusion.") and ".moe." not in k)
        mom = sum(v.numel() for k, v in m.named_parameters() if k.startswith("generalFusion.moe."))
        assert (hdm, mom) == (7357952, 2830336)


def test_vit_variant_names():
    c = make_cfg("RGBNT201", TRANSFORMER_TYPE="vit_base_patch16_224", HEAD=12, SIE_COE=3.0)
    m = make_model(c, 171, 6, 0)
    want = O.init_state_dict(spec_of(m))
    assert set(m.state_dict()) == set(want) and m.feat_dim == 768


def test_flat_storage_and_load():
    m = make_model(small_cfg(), 11, 3, 0)
    flat = m.ptab.flat
    for name, p in m.named_parameters():
        o, shape, n = m.ptab.index[name]
        assert p.data_ptr() == flat.data_ptr() + 4 * o and o % 64 == 0
    sd = O.init_state_dict(spec_of(m), seed=3)
    assert not m.load_state_dict(sd).missing_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert torch.equal(m.ptab.f("classifier.weight"), sd["classifier.weight"])
    m.attach_grads()
    g = m.grad_buffer()
    assert all(p.grad.data_ptr() == g.data_ptr() + 4 * m.ptab.index[n][0] for n, p in m.named_parameters() if p.requires_grad)
    assert m.bottleneck.bias.grad is None                       # frozen (make_model.py:177)
    m2 = m.float()                                              # nn.Module._apply keeps the flat layout
    assert m2.classifier.weight.data_ptr() == m2.ptab.flat.data_ptr() + 4 * m2.ptab.off("classifier.weight")
    # equally spaced parameter blocks that the grouped GEMMs rely on
    m.ptab.stride(*[f"generalFusion.{b}.in_proj_weight" for b in ("r", "n", "t", "rn", "rt", "nt", "rnt")])
    m.ptab.stride("rgb_reduce.1.weight", "nir_reduce.1.weight", "tir_reduce.1.weight")


def test_forward_refuses_cpu():
    m = make_model(small_cfg(), 11, 3, 0).eval()
    x = {k: torch.randn(2, 3, 64, 32) for k in ("RGB", "NI", "TI")}
    with pytest.raises(_lib.DemoHipError):
        m(x, cam_label=torch.zeros(2, dtype=torch.long))


def test_out_of_scope_is_loud():
    with pytest.raises(NotImplementedError):
        make_model(small_cfg(USE_SDTPS=True), 11, 3, 0)
    with pytest.raises(NotImplementedError):
        make_model(small_cfg(ARCH="DeMo_Parallel"), 11, 3, 0)


def test_cabi_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build the HIP extension first (make -C demo2_amd/csrc)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _lib.header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/demo_hip.h but not exported"
    assert set(_lib.PROTOTYPES) | {"demo_last_error"} == set(declared)
    assert _lib.load().demo_abi_version() == 1
    assert ctypes.sizeof(_lib.GemmDesc) == _lib.load().demo_gemm_desc_size()


def test_evaluator_host_aggregation_matches_oracle(monkeypatch):
    """demo2_amd.utils.metrics: everything around the rank_eval kernel (id upload, valid-query mask, CMC from first-hit ranks,
    mean AP) against the oracle's eval_func, with the kernel replaced by a numpy stand-in producing its (ap, first_hit)."""
    import numpy as np
    from demo2_amd.utils import metrics as M
    from oracle import demo_oracle as O

    rng = np.random.default_rng(0)
    nq, ng = 23, 97
    dist = rng.random((nq, ng)).astype(np.float32)
    q_pids, g_pids = rng.integers(0, 9, nq), rng.integers(0, 8, ng)       # identity 8 never appears in the gallery
    q_cam, g_cam = rng.integers(0, 3, nq), rng.integers(0, 3, ng)

    def fake_rank_eval(distmat, qp, gp, qk, gk, want_order=False):
        d, qp, gp, qk, gk = (t.cpu().numpy() for t in (distmat, qp, gp, qk, gk))
        ap, first = np.full(nq, -1.0), np.full(nq, -1, dtype=np.int32)
        for i in range(nq):
            order = np.argsort(d[i], kind="stable")
            keep = ~((gp[order] == qp[i]) & (gk[order] == qk[i]))
            m = (gp[order] == qp[i])[keep].astype(np.int64)
            if m.any():
                first[i] = int(np.argmax(m))
                ap[i] = ((m.cumsum() / (np.arange(m.size) + 1.0)) * m).sum() / m.sum()
        return torch.from_numpy(ap), torch.from_numpy(first)

    monkeypatch.setattr(M.ops, "rank_eval", fake_rank_eval)
    cmc, mAP = M._rank_metrics(torch.from_numpy(dist), q_pids, g_pids, q_cam, g_cam, 20)
    cmc_ref, map_ref = O.eval_market1501(dist, q_pids, g_pids, q_cam, g_cam, 20)
    np.testing.assert_allclose(cmc, cmc_ref, atol=1e-7)
    assert abs(mAP - map_ref) < 1e-12


def test_dgaf_state_dict_matches_reference_fixture(golden):
    """MODEL.USE_DGAF builds the reference's DualGatedAdaptiveFusionV3 keys and shapes (fixture = the reference module's own
    state_dict, C=64 / 4 heads) plus classifier_dgaf / bottleneck_dgaf (make_model.py:630-634); combinations the reference
    does not define stay loud."""
    g = golden("dgaf_v3_c64_h4")
    c = small_cfg(HDM=False, ATM=False, USE_DGAF=True, DGAF_NUM_HEADS=4, TRUNK_OUT_DIM=64)
    m = make_model(c, 11, 3, 0)
    got = m.dgaf.state_dict()
    assert set(got) == set(g["sd"])
    assert all(tuple(got[k].shape) == tuple(g["sd"][k].shape) for k in got)
    assert tuple(m.classifier_dgaf.weight.shape) == (11, 192) and m.bottleneck_dgaf.bias.requires_grad is False
    want = O.init_state_dict(spec_of(m))
    assert set(m.state_dict()) == set(want)
    with pytest.raises(NotImplementedError):
        make_model(small_cfg(USE_DGAF=True), 11, 3, 0)                         # together with HDM/ATM: undefined in the reference
    with pytest.raises(NotImplementedError):
        make_model(small_cfg(HDM=False, ATM=False, USE_DGAF=True, DGAF_VERSION="v1"), 11, 3, 0)
    with pytest.raises(NotImplementedError):
        make_model(small_cfg(USE_FRCA=True), 11, 3, 0)


def test_checkpoint_round_trip_through_load_param(tmp_path):
    """engine/processor.py:157-168 saves `model.state_dict()` with torch.save; test_net.py:43 restores it with load_param.  The
    flat-buffer views must serialise as ordinary per-key tensors and come back into a fresh model's flat buffers."""
    import torch
    torch.manual_seed(0)
    a = make_model(small_cfg(), 11, 3, 0)
    with torch.no_grad():
        for p_ in a.parameters():
            p_.add_(torch.randn_like(p_) * 0.01)                 # (through the views: the alignment gaps of the flat buffer stay zero)
        for n_, b_ in a.named_buffers():
            if b_.is_floating_point():
                b_.add_(0.5)
        a._nbt_flat += 7
    path = tmp_path / "DeMo_10.pth"
    torch.save(a.state_dict(), path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert all(v.shape == a.state_dict()[k].shape for k, v in sd.items())
    assert sum(v.numel() for v in sd.values()) < 1.01 * (a.ptab.flat.numel() + a.stab.flat.numel() + a._nbt_flat.numel())   # no whole-buffer copies per key
    torch.manual_seed(1)
    b = make_model(small_cfg(), 11, 3, 0)
    assert not torch.equal(a.ptab.flat, b.ptab.flat)
    b.load_param(str(path))
    assert torch.equal(a.ptab.flat, b.ptab.flat) and torch.equal(a.stab.flat, b.stab.flat) and torch.equal(a._nbt_flat, b._nbt_flat)
    assert int(b.bottleneck.num_batches_tracked) == 7
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


# ---- round-2 advisor items (host side) -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(direct=0), dict(GLOBAL_LOCAL=False), dict(HDM=False, ATM=False, USE_DGAF=True),
                                dict(HDM=False, ATM=False, USE_DGAF=True, direct=0), dict(HDM=True, ATM=False)])
def test_unused_parameters_are_the_ones_autograd_leaves_without_gradient(kw):
    """torch optimizers skip parameters whose .grad is None; the flat optimizer step must leave exactly those alone.  The set the model
    declares (DeMo.unused_parameter_names) is checked against autograd of the CPU oracle on the same configuration."""
    kw = dict(kw)
    direct = kw.pop("direct", 1)
    m = make_model(small_cfg("clip", head=4, direct=direct, **kw), 11, 3, 0)
    spec = spec_of(m)
    spec.hdm_dropout = 0.0
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in O.init_state_dict(spec, seed=1).items()}
    from util import synth_batch
    x, cam = synth_batch(4, 64, 32, 3)
    out = O.model_forward(sd, spec, x, cam, training=True, updates={})
    O.train_loss(out, torch.tensor([0, 0, 1, 1]), 0.25, 1.0).backward()
    names = {n for n, _ in m.named_parameters()}
    no_grad = {k for k, v in sd.items() if k in names and v.requires_grad and v.grad is None}
    assert m.unused_parameter_names() == no_grad, (sorted(m.unused_parameter_names() ^ no_grad))


def test_param_groups_follow_make_optimizer():
    """solver/make_optimizer.py:5-17: 'bias' keys take WEIGHT_DECAY_BIAS, classifiers 2 x lr under LARGE_FC_LR, requires_grad False is
    skipped; never-used parameters (grad None in torch) are left alone too."""
    m = make_model(small_cfg("clip", head=4, direct=1, HDM=False, ATM=False, USE_DGAF=True), 11, 3, 0)
    blk, tab = m.param_groups(1e-4, 5e-5, large_fc_lr=True)
    tab = tab.view(-1, 2)
    assert tab[0].tolist() == [0.0, 0.0] and blk.numel() * 64 == m.ptab.flat.numel()

    def cls_of(name):
        o, _, n = m.ptab.index[name]
        c = blk[o // 64:(o + n + 63) // 64]
        assert (c == c[0]).all()
        return tuple(round(v, 7) for v in tab[int(c[0])].tolist())
    assert cls_of("BACKBONE.base.transformer.resblocks.0.attn.in_proj_weight") == (1.0, 1e-4)
    assert cls_of("BACKBONE.base.transformer.resblocks.0.attn.in_proj_bias") == (1.0, 5e-5)
    assert cls_of("classifier_dgaf.weight") == (2.0, 1e-4)
    assert cls_of("bottleneck_dgaf.bias") == (0.0, 0.0)            # requires_grad False (make_model.py:137,177)
    assert cls_of("rgb_reduce.1.weight") == (0.0, 0.0)             # not on the DGAF branch's path: torch never moves it
    assert cls_of("classifier.weight") == (0.0, 0.0)


def test_train_step_rejects_configurations_it_does_not_implement():
    from demo2_amd.modeling.hip_backward import TrainStep, loss_settings
    base = small_cfg("clip")
    m = make_model(base, 11, 3, 0)
    for path, val in ((("SOLVER", "OPTIMIZER_NAME"), "RMSprop"), (("DATALOADER", "SAMPLER"), "triplet"),
                      (("MODEL", "METRIC_LOSS_TYPE"), "triplet_center"), (("MODEL", "IF_WITH_CENTER"), "yes")):
        c = base.clone()
        c[path[0]][path[1]] = val
        with pytest.raises(NotImplementedError):
            TrainStep(m, c, 1)
    for name in ("Adam", "AdamW", "SGD"):
        c = base.clone()
        c.SOLVER.OPTIMIZER_NAME = name
        TrainStep(m, c, 1)
    c = base.clone()
    c.MODEL.NO_MARGIN, c.MODEL.IF_LABELSMOOTH = False, "off"
    assert loss_settings(c) == dict(idw=0.25, trw=1.0, eps=0.0, margin=0.3)
    c.DATALOADER.SAMPLER = "softmax"
    assert loss_settings(c) == dict(idw=1.0, trw=0.0, eps=0.0, margin=None)


def test_gradsync_rejects_ranges_that_do_not_tile_the_buffer():
    from demo2_amd.dist import GradSync
    g = torch.zeros(1000)
    with pytest.raises(ValueError):
        GradSync(flat_grad=g, block_ranges=[(100, 500), (400, 700)], tail_range=(700, 1000), last_ranges=[(0, 100)])      # overlap
    m = make_model(small_cfg(), 11, 3, 0)
    s = GradSync(m)
    assert s.covered == m.ptab.flat.numel()
    with pytest.raises(ValueError):
        GradSync(flat_grad=g, block_ranges=[(500, 100)], tail_range=(700, 1000), last_ranges=[])                          # hi < lo
