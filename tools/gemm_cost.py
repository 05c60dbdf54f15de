#!/usr/bin/env python3
"""Cost structure of the 256x256 NT kernel on the training step's products: run once per library build (DEMO_HIP_LIB=<variant>), e.g. the
default, an epilogue that computes but stores nothing (abl1), no epilogue at all (abl2):
    full - abl1 = the output stores;  abl1 - abl2 = the epilogue's arithmetic and operand loads;  abl2 = prologue + main loop.
Also a K sweep at N = 3072 (slope = main loop per 64 of K, intercept = per-tile fixed cost x rounds).
    DEMO_HIP_LIB=$PWD/demo2_amd/lib/libdemo_hip_abl1.so python tools/gemm_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
M, W = 49536, 768
tag = os.path.basename(os.environ.get("DEMO_HIP_LIB", "default"))


def timeit(fn, reps=10, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(best)[len(best) // 2]


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x = rn(M, W).bfloat16()
h = rn(M, 4 * W).bfloat16()
q3 = rn(M, 3 * W).bfloat16()
w1 = (rn(4 * W, W) * W ** -0.5).bfloat16()
w2 = (rn(W, 4 * W) * (4 * W) ** -0.5).bfloat16()
wqkv = (rn(3 * W, W) * W ** -0.5).bfloat16()
wqkvT = (rn(W, 3 * W) * (3 * W) ** -0.5).bfloat16()
wp = (rn(W, W) * W ** -0.5).bfloat16()
b1, b2, b3 = rn(4 * W), rn(W), rn(3 * W)
res32 = rn(M, W)
o16 = torch.empty(M, 4 * W, device=dev, dtype=torch.bfloat16)
p16 = (rn(M, 4 * W) * 0.5).bfloat16()
p8 = torch.randint(0, 253, (M, 4 * W), device=dev, dtype=torch.uint8)
o3 = torch.empty(M, 3 * W, device=dev, dtype=torch.bfloat16)
o32 = torch.empty(M, W, device=dev)
o16s = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
cases = [
    ("qkv fwd   bias                      N=2304 K=768 ", 2.0 * M * 3 * W * W, lambda: ops.gemm(x, wqkv, bias=b3, out=o3)),
    ("qkv fwd   plain                     N=2304 K=768 ", 2.0 * M * 3 * W * W, lambda: ops.gemm(x, wqkv, out=o3)),
    ("fc1 fwd   plain                     N=3072 K=768 ", 2.0 * M * 4 * W * W, lambda: ops.gemm(x, w1, out=o16)),
    ("fc1 fwd   bias+quickgelu+save act'  N=3072 K=768 ", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, bias=b1, out=o16, act=_lib.ACT_QUICKGELU, pre_out=p16, pre_out_deriv=True)),
    ("fc1 fwd   bias+quickgelu+save act' D8 N=3072 K=768", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, bias=b1, out=o16, act=_lib.ACT_QUICKGELU, pre_out=p8, pre_out_deriv=True)),
    ("fc1 fwd   bias+quickgelu (1 output) N=3072 K=768 ", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, bias=b1, out=o16, act=_lib.ACT_QUICKGELU)),
    ("fc1 fwd   bias only                 N=3072 K=768 ", 2.0 * M * 4 * W * W, lambda: ops.gemm(x, w1, bias=b1, out=o16)),
    ("fc2 bwd   * saved act'              N=3072 K=768 ", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, out=o16, act_grad_pre=p16, act_grad_kind=_lib.ACT_SAVED)),
    ("fc2 bwd   * saved act' D8           N=3072 K=768 ", 2.0 * M * 4 * W * W,
     lambda: ops.gemm(x, w1, out=o16, act_grad_pre=p8, act_grad_kind=_lib.ACT_SAVED)),
    ("fc2 fwd   bias+fp32 residual        N=768  K=3072", 2.0 * M * 4 * W * W, lambda: ops.gemm(h, w2, bias=b2, residual=res32, out=o32)),
    ("fc1 bwd   plain                     N=768  K=3072", 2.0 * M * 4 * W * W, lambda: ops.gemm(h, w2, out=o16s)),
    ("qkv bwd   plain                     N=768  K=2304", 2.0 * M * 3 * W * W, lambda: ops.gemm(q3, wqkvT, out=o16s)),
    ("proj fwd  bias+fp32 residual        N=768  K=768 ", 2.0 * M * W * W, lambda: ops.gemm(x, wp, bias=b2, residual=res32, out=o32)),
    ("proj bwd  plain                     N=768  K=768 ", 2.0 * M * W * W, lambda: ops.gemm(x, wp, out=o16s)),
]
TILE = int(os.environ.get("DEMO_FORCE_TILE", "2566"))
tag = f"{tag}:{TILE}"
lib.demo_gemm_force_tile(TILE)
lib.demo_gemm_set_split(0)
if "DEMO_GEMM_DEPHASE" in os.environ:
    tag += ":dephase" + os.environ["DEMO_GEMM_DEPHASE"]
try:
    for name, fl, fn in cases:
        t = timeit(fn)
        print(f"{tag:24s} {name}: {t:7.1f} us {fl / t / 1e6:6.0f} TF", flush=True)
    for k in (128, 256, 512, 768, 1536):
        xk = rn(M, k).bfloat16()
        wk = (rn(4 * W, k) * k ** -0.5).bfloat16()
        t = timeit(lambda: ops.gemm(xk, wk, bias=b1, out=o16))
        print(f"{tag:24s} sweep N=3072 K={k:5d} bias: {t:7.1f} us", flush=True)
finally:
    lib.demo_gemm_force_tile(0)
    lib.demo_gemm_set_split(128)
