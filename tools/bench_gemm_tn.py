#!/usr/bin/env python3
"""Weight-gradient (TN) GEMM: correctness of both forms against fp32 torch on small/ragged shapes, then timing on the trunk's
training shapes (A/B interleaved in one process).   python tools/bench_gemm_tn.py [--batch 128] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)

bad = 0
for (rows, n, k) in [(2048, 256, 256), (2050, 512, 256), (3097, 256, 768), (4100, 768, 512), (2048 + 31, 256, 256)]:
    dy = torch.randn(rows, n, device=dev).bfloat16()
    x = torch.randn(rows, k, device=dev).bfloat16()
    ref = dy.float().t() @ x.float()
    refb = dy.float().sum(0)
    for force in (0, 1280, 1290):
        lib.demo_gemm_force_tile(force)
        g = torch.zeros(n, k, device=dev)
        bsum = torch.zeros(n, device=dev)
        ops.gemm_tn(dy, x, g, accumulate=True, colsum_out=bsum)
        g2 = torch.full((n, k), 7.0, device=dev)
        ops.gemm_tn(dy, x, g2, accumulate=False)
        e1 = ((g - ref).abs().max() / ref.abs().max()).item()
        e2 = ((g2 - ref).abs().max() / ref.abs().max()).item()
        e3 = ((bsum - refb).abs().max() / refb.abs().max()).item()
        ok = e1 < 2e-3 and e2 < 2e-3 and e3 < 2e-3
        bad += not ok
        print(f"check rows={rows} n={n} k={k} form={ {0: 'auto', 1280: '128', 1290: '256-atomic'}[force]}: {e1:.1e} {e2:.1e} colsum {e3:.1e} {'OK' if ok else 'FAIL'}")
lib.demo_gemm_force_tile(0)
if bad:
    sys.exit(1)


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


M, W = 3 * a.batch * 129, 768
for name, n, k in [("qkv", 3 * W, W), ("proj", W, W), ("fc1", 4 * W, W), ("fc2", W, 4 * W), ("clipproj", 512, W)]:
    dy = torch.randn(M, n, device=dev).bfloat16()
    x = torch.randn(M, k, device=dev).bfloat16()
    g = torch.zeros(n, k, device=dev)
    bsum = torch.zeros(n, device=dev)
    res = {}
    for rnd in range(3):
        for force in (1280, 1290, 0):
            lib.demo_gemm_force_tile(force)
            ms = timeit(lambda: ops.gemm_tn(dy, x, g, accumulate=True, colsum_out=bsum), a.reps)
            res.setdefault(force, []).append(ms)
    lib.demo_gemm_force_tile(0)
    fl = 2.0 * M * n * k
    print(f"TN {name:5s} out {n}x{k}: 128^2 {min(res[1280]) * 1e3:7.1f} us {fl / min(res[1280]) / 1e9:6.0f} TF   "
          f"256^2 atomics {min(res[1290]) * 1e3:7.1f} us {fl / min(res[1290]) / 1e9:6.0f} TF   "
          f"256^2 two-pass {min(res[0]) * 1e3:7.1f} us {fl / min(res[0]) / 1e9:6.0f} TF", flush=True)
