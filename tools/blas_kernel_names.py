#!/usr/bin/env python3
"""Which vendor-BLAS kernels torch.mm picks for the step's NT shapes (run under `rocprofv3 --kernel-trace --stats`; the kernel names carry the
macro-tile).  Reference point only; the product never calls the vendor BLAS.
    rocprofv3 --kernel-trace --stats -d gpurun_out/blasnames -o blas -- python3 tools/blas_kernel_names.py"""
import torch

dev = "cuda"
M, W = 49536, 768
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(M, 3 * W, W), (M, 4 * W, W), (M, W, 4 * W), (M, W, 3 * W), (M, W, W), (43520, W, 3 * W), (43520, W, W), (8192, 8192, 8192)]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.mm(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f"M={m} N={n} K={k}: {t:.1f} us {2.0 * m * n * k / t / 1e6:.0f} TF", flush=True)
