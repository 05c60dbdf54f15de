"""CPU, build container only: oracle vs the LIVE reference at full size (BASELINE configs A / B / V).
Skipped wherever /root/reference is absent (e.g. the GPU box)."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import REFERENCE
from oracle import demo_oracle as O

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "modeling")), reason="reference not mounted")]


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    from oracle.gen_golden import load_reference
    return load_reference()


@pytest.mark.parametrize("C,H,B", [(512, 4, 2), (512, 8, 3), (768, 12, 2)])
def test_general_fusion_full(ref, C, H, B):
    torch.manual_seed(0)
    cfg = SimpleNamespace(MODEL=SimpleNamespace(HDM=True, ATM=True))
    gf = ref.attnmoe.GeneralFusion(feat_dim=C, num_experts=7, head=H, reg_weight=0, cfg=cfg).eval()
    sd = {"generalFusion." + k: v for k, v in gf.state_dict().items()}
    ins = [torch.randn(B, 128, C) for _ in range(3)] + [torch.randn(B, C) for _ in range(3)]
    with torch.no_grad():
        want = gf(*ins)
    got = O.general_fusion(sd, ins[:3], ins[3:], O.Spec(feat_dim=C, moe_head=H), training=False)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5)


def test_clip_vit_b16_full(ref):
    torch.manual_seed(0)
    cfg = SimpleNamespace(MODEL=SimpleNamespace(PROMPT=False, ADAPTER=False))
    vt = ref.clip.VisionTransformer(16, 8, 16, 16, 768, 12, 12, 512, cfg).eval()
    img, cv = torch.randn(2, 3, 256, 128), torch.randn(2, 1, 768) * 0.1
    with torch.no_grad():
        want = vt(img, cv)
    got = O.clip_visual(vt.state_dict(), "", img, cv, O.Spec())
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-4)


def test_vit_b16_full(ref):
    torch.manual_seed(0)
    m = ref.vit.vit_base_patch16_224(img_size=(256, 128), stride_size=16, camera=6, sie_xishu=3.0).eval()
    img, cam = torch.randn(2, 3, 256, 128), torch.tensor([5, 1])
    with torch.no_grad():
        want = m(img, cam_label=cam)
    spec = O.Spec(arch="vit", feat_dim=768, camera_num=6, sie_coe=3.0)
    got = O.vit_visual(m.state_dict(), "", img, cam, spec)
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-4)
