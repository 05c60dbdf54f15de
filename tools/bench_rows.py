#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound row kernels at the trunk's training shape (3*B*129 rows x 768).
    python tools/bench_rows.py [--batch 128] [--reps 20]
Prints achieved GB/s against the algorithmic bytes of each kernel."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
M, D = 3 * a.batch * 129, 768
dev = "cuda"


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


x = torch.randn(M, D, device=dev)
w = torch.randn(D, device=dev)
b = torch.randn(D, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, out_dtype=torch.bfloat16, save=True)
ms = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-5, out_dtype=torch.bfloat16, out=y), a.reps)
print(f"layernorm_fwd  f32->bf16: {ms * 1e3:7.1f} us  {M * D * 6 / ms / 1e6:7.0f} GB/s")

# practical HBM ceilings for mixed read/write streams (library elementwise kernels on the same buffers)
t0, t1 = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
ms = timeit(lambda: t1.copy_(t0), a.reps)
print(f"torch copy 1R+1W f32     : {ms * 1e3:7.1f} us  {M * D * 8 / ms / 1e6:7.0f} GB/s")
ms = timeit(lambda: torch.add(t0, x, out=t1), a.reps)
print(f"torch add  2R+1W f32     : {ms * 1e3:7.1f} us  {M * D * 12 / ms / 1e6:7.0f} GB/s")
ms = timeit(lambda: t1.add_(t0), a.reps)
print(f"torch add_ 2R+1W inplace : {ms * 1e3:7.1f} us  {M * D * 12 / ms / 1e6:7.0f} GB/s")
del t0, t1

dy = torch.randn(M, D, device=dev).bfloat16()
dx = torch.zeros(M, D, device=dev)
dxlp = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(D, device=dev)
db = torch.zeros(D, device=dev)
for acc, lp in ((True, True), (True, False), (False, False)):
    ms = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dx=dx, accumulate_dx=acc, dw=dw, db=db,
                                          dx_lp=dxlp if lp else None), a.reps)
    nbytes = M * D * (4 + 2 + 4 + (4 if acc else 0) + (2 if lp else 0))
    print(f"layernorm_bwd  acc={int(acc)} lp={int(lp)}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e6:7.0f} GB/s")

# self-attention forward / backward at the training shape (3B sequences x 12 heads x 129 tokens, head dim 64)
Bn, L, H = 3 * a.batch, 129, 12
qkv = (torch.randn(Bn * L, 3 * H * 64, device=dev) * 0.5).bfloat16()
out, lse = ops.attention_fwd(qkv, Bn, L, H, save_lse=True)
dout = torch.randn_like(out)
ms = timeit(lambda: ops.attention_fwd(qkv, Bn, L, H, save_lse=True), a.reps)
print(f"attention_fwd            : {ms * 1e3:7.1f} us  {Bn * L * H * 64 * 2 * 4 / ms / 1e6:7.0f} GB/s")
ms = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, Bn, L, H), a.reps)
print(f"attention_bwd            : {ms * 1e3:7.1f} us  {Bn * L * H * 64 * 2 * 8 / ms / 1e6:7.0f} GB/s")
