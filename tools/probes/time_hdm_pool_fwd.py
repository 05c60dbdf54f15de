import sys, torch
sys.path.insert(0, '/root/repo')
from demo2_amd import ops
B, L, C, Hh, ld = 128, 129, 512, 8, 128
tok = torch.randn(3*B*L, C, device='cuda').bfloat16()
S = torch.randn(3*B*L, ld, device='cuda')
def f(): return ops.hdm_softmax_pool(S, tok, B, L, C, Hh, ld, 0.1, 77, save_probs=True)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("hdm pool fwd us", e0.elapsed_time(e1)/20*1e3)
