import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
M, N, K = 10000, 512, 1024
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
for _ in range(5):
    ops.gemm_nt(A, B, prec=prec)
torch.cuda.synchronize()
