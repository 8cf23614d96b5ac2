"""Fixed overheads of the GEMM kernels: tiny K (one or two k-steps) so that prologue + epilogue dominate; and the small
GEMMs of the MHIM step (few tiles, long K)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops

def run(tag, fn, reps=40):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3): fn(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{tag:44s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)

def case(M, N, K, **kw):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05
    out = torch.empty(M, N, device="cuda")
    run(f"nt M={M} N={N} K={K} bf16x3 plain", lambda i: ops.gemm_nt(A, B, out=out, prec="bf16x3", **kw))
    if K % 32 == 0 and M > 64 and N % 128 == 0:
        Ap, Bp = ops.pair_planes(A), ops.pair_planes(B)
        run(f"nt M={M} N={N} K={K} paired/feat", lambda i: ops.gemm_nt(Ap, Bp, out=out, prec="bf16x3", paired=True, **kw))

x = torch.randn(1 << 20, device="cuda")
run("axpby-size elementwise kernel (tick)", lambda i: ops.tick(torch.zeros(1, dtype=torch.int64, device="cuda")) if False else ops.colsum(x.view(1024, 1024)))
case(10000, 512, 32)
case(10000, 512, 64)
case(10000, 512, 1024)
case(10000, 128, 512)
case(970, 1024, 512)
case(970, 512, 1024)
case(970, 512, 512)
