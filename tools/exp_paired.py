"""Time the feature projection GEMM (M=10000, N=512, K=1024, bias+GELU+dropout epilogue) in its forms; 8 rotating bags."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
M, N, K = int(os.environ.get("M", 10000)), 512, 1024
bags = [torch.randn(M, K, device="cuda").abs_() for _ in range(8)]
W = torch.randn(N, K, device="cuda") * 0.03
bias = torch.zeros(N, device="cuda")
out = torch.empty(M, N, device="cuda")
def run(tag, fn, reps=40):
    """40 back-to-back launches replayed as ONE hipGraph (the host enqueue rate, ~15 us/call from Python, is out of the way)."""
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3): fn(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:28s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)
run("plain bf16x3", lambda i: ops.gemm_nt(bags[i % 8], W, out=out, bias=bias, act=2, drop_p=0.25, drop_seed=7, prec="bf16x3"))
xps = [ops.pair_planes(b) for b in bags]
wp = ops.pair_planes(W)
run("paired (gemm only)", lambda i: ops.gemm_nt(xps[i % 8], wp, out=out, bias=bias, act=2, drop_p=0.25, drop_seed=7, prec="bf16x3", paired=True))
run("paired no epilogue", lambda i: ops.gemm_nt(xps[i % 8], wp, out=out, prec="bf16x3", paired=True))
run("pair_planes(X)", lambda i: ops.pair_planes(bags[i % 8]))
run("pair_planes(W)", lambda i: ops.pair_planes(W))
rows_small = (torch.arange(M, device="cuda") % 320).contiguous()
rows_id = torch.arange(M, device="cuda")
run("paired noepi rows=identity", lambda i: ops.gemm_nt(xps[i % 8], wp, out=out, rows=rows_id, M=M, prec="bf16x3", paired=True))
run("paired noepi rows%320 (L2)", lambda i: ops.gemm_nt(xps[i % 8], wp, out=out, rows=rows_small, M=M, prec="bf16x3", paired=True))
