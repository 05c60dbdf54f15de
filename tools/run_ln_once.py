#!/usr/bin/env python3
"""A few launches of LayerNorm forward / backward at the training shape -- target for rocprofv3 --pmc runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import ops  # noqa: E402

M, D = 49536, 768
x = torch.randn(M, D, device="cuda")
w = torch.randn(D, device="cuda")
b = torch.randn(D, device="cuda")
y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, out_dtype=torch.bfloat16, save=True)
dy = torch.randn(M, D, device="cuda").bfloat16()
dx = torch.zeros(M, D, device="cuda")
dxlp = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
dw = torch.zeros(D, device="cuda")
db = torch.zeros(D, device="cuda")
for _ in range(3):
    ops.layernorm_fwd(x, w, b, 1e-5, out_dtype=torch.bfloat16, out=y)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx=dx, accumulate_dx=True, dw=dw, db=db, dx_lp=dxlp)
torch.cuda.synchronize()
