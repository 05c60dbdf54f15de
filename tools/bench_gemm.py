#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels on the trunk's real shapes (interleaved A/B in one process, random data).
    python tools/bench_gemm.py [--batch 64] [--tn] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--tn", action="store_true")
a = ap.parse_args()
M = 3 * a.batch * 129
W = 768
dev = "cuda"
lib = _lib.load()
shapes = [("qkv", M, 3 * W, W), ("proj", M, W, W), ("fc1", M, 4 * W, W), ("fc2", M, W, 4 * W), ("clipproj", M, 512, W)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"M={M}")
for name, m, n, k in shapes:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for tile in (1282, 2562, 2566):
            lib.demo_gemm_force_tile(tile)
            ms = timeit(lambda: ops.gemm(x, w, bias=bias, out=out), a.reps)
            res.setdefault(tile, []).append(ms)
    lib.demo_gemm_force_tile(0)
    fl = 2.0 * m * n * k
    line = f"NT {name:9s} N={n} K={k}: " + " ".join(f"{t}:{fl / min(v) / 1e9:6.0f}TF" for t, v in res.items())
    if a.tn:
        dy = torch.randn(m, n, device=dev).bfloat16()
        g = torch.zeros(n, k, device=dev)
        ms = min(timeit(lambda: ops.gemm_tn(dy, x, g, accumulate=True), a.reps) for _ in range(3))
        line += f"   TN: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF"
    print(line, flush=True)
