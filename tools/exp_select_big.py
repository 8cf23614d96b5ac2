"""Timing of mhimx_select_mask on large score vectors (the c3 / c5 selects): N = 50 000 / 200 000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
for n, k in ((50000, 3000), (200000, 12000)):
    s = torch.rand(n, device="cuda")
    perm = torch.randperm(k, device="cuda")
    for _ in range(5):
        ops.select_mask(s, k, k // 2, True, perm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.select_mask(s, k, k // 2, True, perm)
    e1.record(); torch.cuda.synchronize()
    print(f"N={n} k={k}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per select (incl. its memset and workspace allocation)")
