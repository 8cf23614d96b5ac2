import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N, K = 512, 1024
A = torch.randn(16384, K, device="cuda"); B = torch.randn(N, K, device="cuda"); out = torch.empty(16384, N, device="cuda")
for M in (4096, 8192, 8320, 10000, 12288, 16384):
    for prec in ("bf16x3", "f16s"):
        t = timeit(lambda: ops.gemm_nt(A[:M], B, out=out[:M], prec=prec))
        print(f"M={M:6d} WGs={(M+127)//128*4:4d} {prec:7s}: {t:7.1f} us  ({2*M*N*K/t/1e6:.1f} TF fp32-equiv)")
