import os, sys, torch
sys.path.insert(0, '/root/repo')
from demo2_amd import _lib, ops
lib = _lib.load()
M = 49536
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (768, 2304):
  for k in (64, 128, 256, 768, 1536, 3072):
    x = torch.randn(M, k, device='cuda').bfloat16(); w = torch.randn(n, k, device='cuda').bfloat16(); bias = torch.randn(n, device='cuda')
    out = torch.empty(M, n, device='cuda', dtype=torch.bfloat16)
    res = []
    for cfg in (1282, 2566):
        lib.demo_gemm_force_tile(cfg)
        us = min(timeit(lambda: ops.gemm(x, w, bias=bias, out=out)) for _ in range(2))
        res.append(us)
    lib.demo_gemm_force_tile(0)
    print(f"N={n} K={k}: 1282 {res[0]:7.1f} us   2566 {res[1]:7.1f} us", flush=True)
