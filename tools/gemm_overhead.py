#!/usr/bin/env python3
"""NT GEMM cost structure: (1) a K sweep at fixed M x N (fixed cost = launch + prologue + epilogue, slope = main loop),
(2) the trunk's real epilogue variants at the training shape.   python tools/gemm_overhead.py [--sweep]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
M = 49536
dev = "cuda"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def both(fn):
    res = []
    for cfg in (1282, 2566):
        lib.demo_gemm_force_tile(cfg)
        res.append(min(timeit(fn) for _ in range(2)))
    lib.demo_gemm_force_tile(0)
    return res


if "--sweep" in sys.argv:
    for n in (768, 2304):
        for k in (64, 128, 256, 768, 1536, 3072):
            x = torch.randn(M, k, device=dev).bfloat16()
            w = torch.randn(n, k, device=dev).bfloat16()
            bias = torch.randn(n, device=dev)
            out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
            r = both(lambda: ops.gemm(x, w, bias=bias, out=out))
            print(f"N={n} K={k}: 128^2 {r[0]:7.1f} us   256^2 {r[1]:7.1f} us", flush=True)

W = 768
x = torch.randn(M, W, device=dev).bfloat16()
h = torch.randn(M, 4 * W, device=dev).bfloat16()
w1 = (torch.randn(4 * W, W, device=dev) * W ** -0.5).bfloat16()
w2 = (torch.randn(W, 4 * W, device=dev) * (4 * W) ** -0.5).bfloat16()
b1 = torch.randn(4 * W, device=dev)
b2 = torch.randn(W, device=dev)
res32 = torch.randn(M, W, device=dev)
o16 = torch.empty(M, 4 * W, device=dev, dtype=torch.bfloat16)
p16 = torch.empty(M, 4 * W, device=dev, dtype=torch.bfloat16)
o32 = torch.empty(M, W, device=dev)
o16s = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
cases = [
    ("fc1 fwd   bias -> bf16", 2.0 * M * 4 * W * W, lambda: ops.gemm(x, w1, bias=b1, out=o16)),
    ("fc1 fwd   bias, pre_out, quickgelu -> bf16", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, bias=b1, out=o16, act=_lib.ACT_QUICKGELU, pre_out=p16)),
    ("fc2 bwd   dY W2^T * act'(pre) -> bf16 (N=3072,K=768)", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, out=o16, act_grad_pre=p16, act_grad_kind=_lib.ACT_QUICKGELU)),
    ("fc2 fwd   bias + residual -> f32 (N=768,K=3072)", 2.0 * M * 4 * W * W, lambda: ops.gemm(h, w2, bias=b2, residual=res32, out=o32)),
    ("fc1 bwd   -> bf16 (N=768,K=3072)", 2.0 * M * 4 * W * W, lambda: ops.gemm(h, w2, out=o16s)),
]
for name, fl, fn in cases:
    r = both(fn)
    print(f"{name:55s}: 128^2 {r[0]:7.1f} us {fl / r[0] / 1e6:5.0f} TF   256^2 {r[1]:7.1f} us {fl / r[1] / 1e6:5.0f} TF", flush=True)
