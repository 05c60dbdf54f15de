#!/usr/bin/env python3
"""LayerNorm kernels of the trunk at the training shape (49536 x 768): forward on fp32 / bf16 streams, backward (bf16 gradient stream)
with fp32 x, bf16 x on the 4-column kernel and bf16 x on the 8-column kernel; interleaved rounds, median, algorithmic GB/s."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demo2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
M, D = 49536, 768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x32 = torch.randn(M, D, device=dev, generator=g)
x16 = x32.bfloat16()
w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
dy = torch.randn(M, D, device=dev, generator=g).bfloat16()
res = torch.randn(M, D, device=dev, generator=g).bfloat16()
out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
_, mean, rstd = ops.layernorm_fwd(x32, w, b, 1e-5, torch.bfloat16, save=True)
dw, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)


def bwd(x, lp8):
    lib.demo_layernorm_set_lp8(lp8)
    ops.layernorm_bwd_lp(dy, x, w, mean, rstd, res, out, dw=dw, db=db)


arms = [("fwd fp32 x -> bf16 (229 MB)", lambda: ops.layernorm_fwd(x32, w, b, 1e-5, torch.bfloat16, save=True), 229e6),
        ("fwd bf16 x -> bf16, wave per row (152 MB)", lambda: (lib.demo_layernorm_set_lp8(0), ops.layernorm_add_fwd(x16, None, w, b, 1e-5, save=True)), 152e6),
        ("fwd bf16 x -> bf16, half wave per row (152 MB)", lambda: (lib.demo_layernorm_set_lp8(1), ops.layernorm_add_fwd(x16, None, w, b, 1e-5, save=True)), 152e6),
        ("bwd lp, fp32 x, 4 cols (380 MB)", lambda: bwd(x32, 1), 380e6),
        ("bwd lp, bf16 x, 4 cols (304 MB)", lambda: bwd(x16, 0), 304e6),
        ("bwd lp, bf16 x, 8 cols, 128 threads (304 MB)", lambda: bwd(x16, 2), 304e6),
        ("bwd lp, bf16 x, half wave per row (304 MB)", lambda: bwd(x16, 3), 304e6),
        ("bwd lp, bf16 x, wave per row, 2 rows ahead (304 MB)", lambda: bwd(x16, 1), 304e6)]
t = {n: [] for n, _, _ in arms}
for _ in range(3):
    for n, f, _ in arms:
        f()
torch.cuda.synchronize()
for _ in range(25):
    for n, f, _ in arms:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f()
        e1.record()
        e1.synchronize()
        t[n].append(e0.elapsed_time(e1) * 1e3 / 4)
lib.demo_layernorm_set_lp8(1)
for n, _, nb in arms:
    med = statistics.median(t[n])
    print(f"{n:52s} median {med:7.1f} us  min {min(t[n]):7.1f} us  {nb / med / 1e6:5.2f} TB/s")
