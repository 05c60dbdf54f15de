import sys, torch
sys.path.insert(0, '/root/repo')
from demo2_amd import _lib, ops
lib = _lib.load()
torch.manual_seed(0)
for cfg in (1282, 2566, 1283, 1284, 2562, 2563, 1292, 2592):
    lib.demo_gemm_force_tile(cfg)
    for (M, N, K) in [(1161, 768, 768), (300, 256, 192), (5000, 512, 64), (700, 256, 3072)]:
        a = torch.randn(M, K).bfloat16(); b = (torch.randn(N, K) * K ** -0.5).bfloat16(); bias = torch.randn(N)
        ref = a.float() @ b.float().t() + bias
        got = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), out_dtype=torch.float32).cpu()
        err = (got - ref).abs().max().item()
        print(cfg, (M, N, K), 'max err', f'{err:.2e}', 'OK' if err < 2e-3 else 'FAIL')
lib.demo_gemm_force_tile(0)
