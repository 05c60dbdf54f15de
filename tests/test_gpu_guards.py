"""GPU: the guards the round-1 advisor asked for -- out-of-range index arguments (camera ids, class labels) at the C ABI, the
reference's optimizer / loss configurations in the fused training step, parameters torch would never move."""
import pytest
import torch

from demo2_amd import ops
from demo2_amd.modeling import make_model
from demo2_amd.modeling.hip_backward import TrainStep, fused_loss_and_grads, make_loss_fn, total_loss
from oracle import demo_oracle as O
from util import small_cfg, spec_of, synth_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(cfg, dtype=torch.float32, train=False):
    torch.manual_seed(0)
    m = make_model(cfg, 11, 3, 0)
    m.load_state_dict(O.init_state_dict(spec_of(m), seed=1))
    m = m.to(DEV).set_compute_dtype(dtype)
    m.geom.hdm_dropout = 0.0
    return m.train() if train else m.eval()


def test_out_of_range_camera_id_is_reported_and_reads_nothing_outside_the_table():
    """The reference's cv_embed[cam_label] raises IndexError (meta_arch.py:95-99).  Here: a Python int is checked on the host at
    once; a tensor value is clamped by the kernel (cv_embed sits inside the flat parameter buffer: an unclamped index would read the
    neighbouring parameters) and reported through the device error word -- at check_errors(), or by the NEXT forward's poll."""
    m = build(small_cfg("clip"))
    x, cam = synth_batch(4, 64, 32, 3)
    xd = {k: v.to(DEV) for k, v in x.items()}
    with pytest.raises(IndexError):
        m(xd, cam_label=3)
    with pytest.raises(IndexError):
        m(xd, cam_label=-1)
    with torch.no_grad():
        good = m(xd, cam_label=torch.tensor([2, 2, 2, 2], device=DEV))
        m.check_errors()
        got = m(xd, cam_label=torch.tensor([2, 7, 2, 2], device=DEV))        # 7 >= camera_num = 3: clamped to the last row
    torch.testing.assert_close(got, good, rtol=1e-5, atol=1e-5)                  # (split-K fp32 atomics: not bit-identical run to run)
    with pytest.raises(IndexError, match="cam_label"):
        m.check_errors()
    m.check_errors()                                                          # the word was cleared by the raise
    with torch.no_grad():
        m(xd, cam_label=torch.tensor([0, -5, 1, 1], device=DEV))
        torch.cuda.synchronize()
        with pytest.raises(IndexError, match="cam_label"):
            m(xd, cam_label=cam.to(DEV))                                      # polled at the next call, no device sync needed


def test_out_of_range_camera_id_in_training_leaves_the_gradient_table_alone():
    cfg = small_cfg("clip")
    m = build(cfg, train=True)
    x, _ = synth_batch(8, 64, 32, 3)
    xd = {k: v.to(DEV) for k, v in x.items()}
    target = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], device=DEV)
    ts = TrainStep(m, cfg, 1, optimizer=False)
    ts.step(xd, target, torch.tensor([0, 1, 2, 9, 0, 1, 2, -3], device=DEV))
    with pytest.raises(IndexError, match="cam_label"):
        m.check_errors()
    g = m.grad_buffer()
    assert torch.isfinite(g).all()
    # the neighbours of cv_embed in the flat buffer got only their own gradients: compare with a run whose bad rows carry valid ids
    # that receive the clamped rows' forward (ids 2 and 0) -- the bad rows add NOTHING to dcam, everything else is identical
    o, shape, n = m.ptab.index["BACKBONE.cv_embed"]
    bad = g.clone()
    m.attach_grads(zero=True)
    ts.step(xd, target, torch.tensor([0, 1, 2, 2, 0, 1, 2, 0], device=DEV))
    m.check_errors()
    ok = m.grad_buffer()
    mask = torch.ones_like(ok, dtype=torch.bool)
    mask[o:o + n] = False
    torch.testing.assert_close(bad[mask], ok[mask], rtol=1e-4, atol=1e-6)
    assert (bad[o:o + n] - ok[o:o + n]).abs().max() > 0                       # rows 3 and 7 are missing from dcam in the bad run


def test_out_of_range_label_poisons_the_loss_and_is_reported():
    """softmax_loss.py:30-31: scatter_ raises for a label >= num_classes.  The kernel reads nothing out of bounds, makes the loss NaN
    (a wrong label must not train silently) and sets the error bit."""
    err = ops.DeviceErrors(DEV)
    logits = torch.randn(6, 11, device=DEV)
    loss = torch.zeros(1, device=DEV)
    d = ops.xent_label_smooth(logits, torch.tensor([0, 1, 11, 3, 4, 5], device=DEV), loss, 0.1, 1.0, err=err)
    assert torch.isnan(loss).all() and torch.isnan(d[2]).all() and torch.isfinite(d[[0, 1, 3, 4, 5]]).all()
    with pytest.raises(IndexError, match="label"):
        err.check()
    loss.zero_()
    ops.xent_label_smooth(logits, torch.tensor([0, 1, -1, 3, 4, 5], device=DEV), loss, 0.1, 1.0, err=err)
    with pytest.raises(IndexError, match="label"):
        err.check()
    loss.zero_()
    ops.xent_label_smooth(logits, torch.tensor([0, 1, 10, 3, 4, 5], device=DEV), loss, 0.1, 1.0, err=err)
    err.check()
    assert torch.isfinite(loss).all()


@pytest.mark.parametrize("over", [dict(NO_MARGIN=False), dict(IF_LABELSMOOTH="off"), dict(NO_MARGIN=False, IF_LABELSMOOTH="off"), dict(SAMPLER="softmax")])
def test_fused_loss_follows_make_loss_configuration(over):
    """layers/make_loss.py: margin triplet (MODEL.NO_MARGIN False, SOLVER.MARGIN), plain cross entropy (IF_LABELSMOOTH 'off'), the
    'softmax' sampler's cross-entropy-only loss -- kernels against the torch restatement of the same branches."""
    cfg = small_cfg("clip")
    for k, v in over.items():
        if k == "SAMPLER":
            cfg.DATALOADER.SAMPLER = v
        else:
            cfg.MODEL[k] = v
    torch.manual_seed(5)
    target = (torch.arange(32) // 4).to(DEV)
    # features clustered by identity (unit prototypes + noise of two sizes): the margin hinge is active for some anchors only
    # (drawn on the CPU: the same numbers whatever the device generator does; 16 / 16 anchors on either side of the hinge)
    protos = torch.randn(8, 384)[torch.arange(32) // 4] * 0.06
    noise = torch.randn(32, 384) * torch.where(torch.arange(32) % 2 == 0, 0.01, 0.06)[:, None]
    outs = [torch.randn(32, 11).to(DEV).requires_grad_(True), torch.randn(32, 896).to(DEV).requires_grad_(True),
            torch.randn(32, 11).to(DEV).requires_grad_(True), (protos + noise).to(DEV).requires_grad_(True), 0]
    ref = total_loss(outs, target, make_loss_fn(cfg))
    ref.backward()
    loss, roots, grads = fused_loss_and_grads(outs, target, cfg)
    torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-5)
    assert len(roots) == (2 if "SAMPLER" in over else 4)
    for r, g in zip(roots, grads):
        torch.testing.assert_close(g, r.grad, rtol=2e-4, atol=2e-6)
    if "NO_MARGIN" in over:                                      # the hinge is active for some anchors and inactive for others
        f = outs[3].detach()
        d = torch.cdist(f, f)
        same = target[:, None] == target[None, :]
        z = torch.where(same, d, d.new_full((), -1e9)).max(1).values - torch.where(~same, d, d.new_full((), 1e9)).min(1).values + 0.3
        assert (z > 0).any() and (z < 0).any()


@pytest.mark.parametrize("name", ["Adam", "AdamW", "SGD"])
def test_flat_optimizer_step_equals_make_optimizer_groups(name):
    """solver/make_optimizer.py as the reference builds it -- per-parameter groups, WEIGHT_DECAY_BIAS for 'bias' keys, 2 x lr for
    classifiers under LARGE_FC_LR, requires_grad-False parameters skipped -- two steps of torch's optimizer on the same gradients
    against two fused flat steps."""
    cfg = small_cfg("clip")
    cfg.SOLVER.OPTIMIZER_NAME, cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_BIAS = name, 1e-3, 1e-2, 5e-3
    cfg.SOLVER.LARGE_FC_LR = True
    m = build(cfg, train=True)
    x, cam = synth_batch(8, 64, 32, 3)
    xd, camd = {k: v.to(DEV) for k, v in x.items()}, cam.to(DEV)
    target = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], device=DEV)
    ref = {n: p.detach().cpu().clone().requires_grad_(p.requires_grad) for n, p in m.named_parameters()}
    groups = []
    for key, value in ref.items():                               # make_optimizer.py:5-17
        if not value.requires_grad:
            continue
        lr, wd = 1e-3, 1e-2
        if "bias" in key:
            wd = 5e-3
        if "classifier" in key or "arcface" in key:
            lr = 2e-3
        groups.append({"params": [value], "lr": lr, "weight_decay": wd})
    if name == "SGD":
        opt = torch.optim.SGD(groups, momentum=cfg.SOLVER.MOMENTUM)
    elif name == "AdamW":
        opt = torch.optim.AdamW(groups, lr=1e-3, weight_decay=1e-2)
    else:
        opt = torch.optim.Adam(groups)
    ts = TrainStep(m, cfg, 1, optimizer=False)
    ts2 = TrainStep(m, cfg, 1)                                   # the stepping one (shares the model)
    frozen_before = m.bottleneck.bias.detach().clone()
    for it in range(2):
        ts.step(xd, target, camd)                                # gradients only
        g = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}
        for n, p in ref.items():
            if p.requires_grad:
                p.grad = g[n]
        opt.step()
        ts2.t += 1
        m.optim_step(ts2.opt_kind, ts2.m, ts2.v, 1e-3, cfg.SOLVER.MOMENTUM if name == "SGD" else 0.9, 0.999, 1e-8, 1e-2, ts2.t,
                     groups=m.param_groups(1e-2, 5e-3, True))
        for n, p in m.named_parameters():
            torch.testing.assert_close(p.detach().cpu(), ref[n].detach(), rtol=2e-5, atol=2e-6, msg=lambda s, n=n: f"{name} step {it} {n}: {s}")
    assert torch.equal(m.bottleneck.bias.detach(), frozen_before)


def test_never_used_parameters_do_not_drift_under_weight_decay():
    """DGAF-only branch: rgb/nir/tir_reduce and the fused 'ori' head exist (reference constructor) but get no gradient; torch skips
    them, so after optimizer steps with weight decay they -- and their optimizer state -- are bit-identical."""
    cfg = small_cfg("clip", HDM=False, ATM=False, USE_DGAF=True)
    cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_BIAS = 0.05, 0.05
    m = build(cfg, train=True)
    x, cam = synth_batch(8, 64, 32, 3)
    target = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], device=DEV)
    unused = m.unused_parameter_names()
    assert "rgb_reduce.1.weight" in unused and "classifier.weight" in unused
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    ts = TrainStep(m, cfg, 1, lr=1e-2)
    for _ in range(3):
        ts.step({k: v.to(DEV) for k, v in x.items()}, target, cam.to(DEV))
    moved = 0
    for n, p in m.named_parameters():
        if n in unused or not p.requires_grad:
            assert torch.equal(p.detach(), before[n]), n
            o, _, k = m.ptab.index[n]
            assert ts.m[o:o + k].abs().max() == 0 and ts.v[o:o + k].abs().max() == 0
        else:
            moved += int(not torch.equal(p.detach(), before[n]))
    assert moved > 50
