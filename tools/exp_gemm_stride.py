"""Experiment: does the row pitch of the fp32 operands (4 KiB = power of two) camp on memory channels?"""
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

M, N, K = 10000, 512, 1024
for pa in (0,):
    for pb in (0,):
        A = torch.randn(M, K + pa, device="cuda")[:, :K]
        B = torch.randn(N, K + pb, device="cuda")[:, :K]
        out = torch.empty(M, N, device="cuda")
        for prec in ("bf16x3", "f16s"):
            # bypass contiguity check
            import ctypes as C
            from mhim_mil_amd import _lib as L
            g = L.GemmNT(A=A.data_ptr(), lda=A.stride(0), rows=None, B=B.data_ptr(), ldb=B.stride(0), C=out.data_ptr(), ldc=N,
                         M=M, N=N, K=K, prec=L.PREC[prec])
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            t = timeit(lambda: L.lib().mhimx_gemm_nt(st, C.byref(g)))
            print(f"pad_a={pa:3d} pad_b={pb:3d} {prec:7s}: {t:7.1f} us  ({2*M*N*K/t/1e6:.1f} TF fp32-equiv)")

A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
for prec in ("bf16x3", "f16s"):
    pl = ops.split_planes(B, prec)
    t = timeit(lambda: ops.gemm_nt(A, B, out=out, prec=prec, b_planes=pl))
    print(f"planes 128x256 {prec:7s}: {t:7.1f} us  ({2*M*N*K/t/1e6:.1f} TF fp32-equiv)")
    for MM in (9984, 8192):
        t = timeit(lambda: ops.gemm_nt(A[:MM], B, out=out[:MM], prec=prec, b_planes=pl))
        print(f"planes 128x256 {prec:7s} M={MM}: {t:7.1f} us  ({2*MM*N*K/t/1e6:.1f} TF fp32-equiv)")
