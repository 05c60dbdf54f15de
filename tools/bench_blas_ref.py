#!/usr/bin/env python3
"""Reference point, not a product path: the vendor BLAS (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the training step's NT
products, next to this library's kernels on the same box.  What it answers: is there a structure on this chip that runs the K = 768
products faster than the 256^2 kernel here?  (Plain products only: the vendor call has no fused epilogue.)
    python tools/bench_blas_ref.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
M, W = 49536, 768


def timeit(fn, reps=10, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device=dev).manual_seed(0)
for name, n, k in (("qkv fwd", 3 * W, W), ("fc1 fwd", 4 * W, W), ("fc2 fwd", W, 4 * W), ("dqkv->dh", W, 3 * W), ("proj", W, W)):
    a = torch.randn(M, k, device=dev, generator=g).bfloat16()
    w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    t_blas = timeit(lambda: torch.mm(a, w.t(), out=out))
    t_here = timeit(lambda: ops.gemm(a, w, out=out))
    fl = 2.0 * M * n * k
    print(f"{name:9s} M={M} N={n:5d} K={k:5d} plain: vendor BLAS {t_blas:7.1f} us {fl / t_blas / 1e6:6.0f} TF | this library "
          f"{t_here:7.1f} us {fl / t_here / 1e6:6.0f} TF", flush=True)
# weight-gradient shape (TN): dW = dY^T X
for name, n, k in (("dW fc1", 4 * W, W), ("dW qkv", 3 * W, W), ("dW proj", W, W)):
    dy = torch.randn(M, n, device=dev, generator=g).bfloat16()
    x = torch.randn(M, k, device=dev, generator=g).bfloat16()
    out = torch.empty(n, k, device=dev, dtype=torch.bfloat16)
    t_blas = timeit(lambda: torch.mm(dy.t(), x, out=out))
    fl = 2.0 * M * n * k
    print(f"{name:9s} ({n} x {k}) over {M} rows: vendor BLAS {t_blas:7.1f} us {fl / t_blas / 1e6:6.0f} TF", flush=True)
