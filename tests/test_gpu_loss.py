"""GPU: the fused loss kernels (csrc/loss.hip) against the reference-generated golden values and against torch autograd
of the oracle's restatement (value and gradient)."""
import pytest
import torch

from demo2_amd import ops
from oracle import demo_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_losses_match_reference_golden(golden):
    g = golden("losses")
    score, feat, target = (g[k].to(DEV) for k in ("score", "feat", "target"))
    loss = torch.zeros(1, device=DEV)
    ops.xent_label_smooth(score, target, loss, 0.1, 1.0)
    torch.testing.assert_close(loss.cpu()[0], torch.tensor(float(g["xent"])), rtol=1e-5, atol=1e-6)
    loss.zero_()
    ops.triplet_soft_margin(feat, target, loss, 1.0)
    torch.testing.assert_close(loss.cpu()[0], torch.tensor(float(g["triplet"])), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,K", [(8, 11), (128, 171), (33, 300)])
def test_xent_value_and_grad(B, K):
    torch.manual_seed(B)
    x = (torch.randn(B, K) * 3).requires_grad_(True)
    t = torch.randint(0, K, (B,))
    ref = O.xent_label_smooth(x, t) * 0.25
    ref.backward()
    loss = torch.full((1,), 2.0, device=DEV)                        # accumulates into the given scalar
    d = ops.xent_label_smooth(x.detach().to(DEV), t.to(DEV), loss, 0.1, 0.25)
    torch.testing.assert_close(loss.cpu()[0] - 2.0, ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d.cpu(), x.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,D,per_id", [(8, 48, 2), (128, 1536, 4), (128, 3584, 4), (20, 100, 20), (12, 64, 1)])
def test_triplet_value_and_grad(B, D, per_id):
    """per_id = B: no negatives at all; per_id = 1: the hardest positive is the anchor itself (clamped distance, zero gradient)."""
    torch.manual_seed(D)
    f = torch.randn(B, D).requires_grad_(True)
    labels = torch.arange(B) // per_id
    if per_id == B:
        ref_val, ref_grad = torch.tensor(0.0), torch.zeros(B, D)    # the kernel skips anchors without a negative
    else:
        ref = O.triplet_soft_margin(f, labels)
        ref.backward()
        ref_val, ref_grad = ref.detach(), f.grad
    loss = torch.zeros(1, device=DEV)
    d = ops.triplet_soft_margin(f.detach().to(DEV), labels.to(DEV), loss, 1.0)
    torch.testing.assert_close(loss.cpu()[0], ref_val, rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(d.cpu(), ref_grad, rtol=2e-4, atol=2e-6)


def test_fused_total_matches_torch_glue():
    from demo2_amd.modeling.hip_backward import fused_loss_and_grads, make_loss_fn, total_loss
    from util import make_cfg
    cfg = make_cfg("RGBNT201")
    torch.manual_seed(3)
    target = (torch.arange(32) // 4).to(DEV)
    outs = [torch.randn(32, 171, device=DEV, requires_grad=True), torch.randn(32, 3584, device=DEV, requires_grad=True),
            torch.randn(32, 171, device=DEV, requires_grad=True), torch.randn(32, 1536, device=DEV, requires_grad=True), 0]
    ref = total_loss(outs, target, make_loss_fn(cfg))
    ref.backward()
    loss, roots, grads = fused_loss_and_grads(outs, target, cfg)
    torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-5)
    assert len(roots) == 4
    for r, g in zip(roots, grads):
        torch.testing.assert_close(g, r.grad, rtol=2e-4, atol=2e-6)
