import os, sys, torch
sys.path.insert(0, '/root/repo')
from demo2_amd import _lib, ops
lib = _lib.load()
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, k) in [(3072, 768), (768, 768)]:
    for rows in (6192, 12384, 24768, 49536, 99072):
        dy = torch.randn(rows, n, device='cuda').bfloat16(); x = torch.randn(rows, k, device='cuda').bfloat16()
        g = torch.zeros(n, k, device='cuda'); b = torch.zeros(n, device='cuda')
        us = min(timeit(lambda: ops.gemm_tn(dy, x, g, accumulate=True, colsum_out=b)) for _ in range(2))
        us2 = min(timeit(lambda: ops.gemm_tn(dy, x, g, accumulate=True)) for _ in range(2))
        print(f"out {n}x{k} rows={rows}: {us:7.1f} us {2.0*rows*n*k/us/1e6:6.0f} TF   (no colsum {us2:7.1f} us)", flush=True)
