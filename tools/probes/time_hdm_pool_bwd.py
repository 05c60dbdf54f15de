import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from demo2_amd import ops
B, L, C, Hh, ld = 128, 129, 512, 8, 128
tok = torch.randn(3*B*L, C, device='cuda').bfloat16()
dpooled = torch.randn(B,7,Hh,C, device='cuda'); pooled = torch.randn(B,7,Hh,C, device='cuda')
probs = torch.rand(B,7,Hh,3*L, device='cuda'); psum = torch.ones(B,7,Hh, device='cuda')
dpsum = torch.zeros(B,7,Hh, device='cuda')
def f(): return ops.hdm_softmax_pool_bwd(dpooled, dpsum, probs, tok, pooled, psum, B, L, C, Hh, ld, 0.1, 1234)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("hdm pool bwd us", e0.elapsed_time(e1)/20*1e3)
