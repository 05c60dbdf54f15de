#!/usr/bin/env python3
"""A few launches of the TN (weight-gradient) GEMM on the fc1 / proj shapes -- target for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import ops
M = 49536
for (n, k) in [(3072, 768), (768, 768)]:
    dy = torch.randn(M, n, device="cuda").bfloat16(); x = torch.randn(M, k, device="cuda").bfloat16()
    g = torch.zeros(n, k, device="cuda")
    for _ in range(3):
        ops.gemm_tn(dy, x, g, accumulate=True)
torch.cuda.synchronize()
