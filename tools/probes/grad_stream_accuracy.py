#!/usr/bin/env python3
"""Gradient accuracy of the bf16 compute mode with the fp32 / bf16 residual-gradient stream (DEMO_BF16_GRAD_STREAM), config A at
full size (12 blocks), B = 8, against the same model's own fp32 compute mode (dropout off).  Prints, over all parameters with a
non-negligible gradient, the norm-relative error: median / 90th / max, and the error of the whole flat gradient."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from demo2_amd.modeling import make_model  # noqa: E402
from demo2_amd.modeling.hip_backward import TrainStep  # noqa: E402
from util import make_cfg, synth_batch  # noqa: E402

cfg = make_cfg("RGBNT201")
torch.manual_seed(0)
m = make_model(cfg, 171, 4, 0)
with torch.no_grad():                                     # the module's own initialisation, made "alive": biases, BN statistics, token scale
    g_ = torch.Generator().manual_seed(1)
    for n_, p_ in m.named_parameters():
        p_.add_(torch.randn(p_.shape, generator=g_) * (0.02 if p_.dim() > 1 else 0.05))
    for n_, b_ in m.named_buffers():
        if "running_var" in n_:
            b_.mul_(1 + 0.1 * torch.rand(b_.shape, generator=g_))
        elif "running_mean" in n_:
            b_.add_(0.1 * torch.randn(b_.shape, generator=g_))
m = m.to("cuda").train()
m.geom.hdm_dropout = 0.0
x, cam = synth_batch(8, 256, 128, 4, device="cuda")
target = torch.tensor([3, 3, 3, 3, 9, 9, 9, 9], device="cuda")
grads = {}
feats = {}
VARIANTS = (("fp32", torch.float32, False, False), ("bf16 / fp32 stream", torch.bfloat16, False, False),
            ("bf16 / bf16 stream", torch.bfloat16, True, False), ("bf16 / bf16 fwd+bwd", torch.bfloat16, True, True))
for name, dt, lps, fwd in VARIANTS:                       # eval features first: the training steps below move the BatchNorm statistics
    m.set_compute_dtype(dt)
    m.bf16_grad_stream, m.bf16_fwd_stream = lps, fwd
    with torch.no_grad():
        feats[name] = m.eval()(x, cam_label=cam).float().cpu()
m.train()
stats0 = m.stab.flat.clone()
for name, dt, lps, fwd in VARIANTS:
    m.set_compute_dtype(dt)
    m.bf16_grad_stream = lps
    m.bf16_fwd_stream = fwd
    m.stab.flat.copy_(stats0)
    ts = TrainStep(m, cfg, 1, optimizer=False)
    m.grad_buffer().zero_()
    loss = ts.step(x, target, cam)
    torch.cuda.synchronize()
    grads[name] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    print(f"{name:20s} loss {loss.item():.5f}")
ref = grads["fp32"]
for name in list(grads)[1:]:
    errs = []
    num = den = 0.0
    for n, g in grads[name].items():
        r = ref[n]
        if r.norm() > 1e-6:
            errs.append(((g - r).norm() / r.norm()).item())
        num += (g - r).double().pow(2).sum().item()
        den += r.double().pow(2).sum().item()
    e = torch.tensor(errs)
    fe = (feats[name] - feats["fp32"])
    print(f"{name:20s} eval features vs fp32 mode: max abs {fe.abs().max():.4f}  relative {fe.norm() / feats['fp32'].norm():.5f}")
    print(f"{name:20s} per-parameter relative error: median {e.median():.4f}  p90 {e.quantile(0.9):.4f}  max {e.max():.4f}   whole gradient {(num / den) ** 0.5:.4f}")
