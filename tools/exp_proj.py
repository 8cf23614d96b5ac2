"""Standalone check + timing of mhimx_bag_project (teacher + student projection in one pass) against fp64 and feat_gemm."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops

dev = "cuda"
N, D, E = int(os.environ.get("N", 10000)), 1024, 512
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g).abs_()
wt = torch.randn(E, D, device=dev, generator=g) * 0.036
ws = torch.randn(E, D, device=dev, generator=g) * 0.036
bt = torch.randn(E, device=dev, generator=g) * 0.1
bs = torch.randn(E, device=dev, generator=g) * 0.1
wtp, wsp = ops.pair_planes(wt), ops.pair_planes(ws)

def run(p=0.0, seed=5):
    hs = [ops.ProjHead(wtp, bt, drop_p=p, drop_seed=seed), ops.ProjHead(wsp, bs, drop_p=p, drop_seed=seed + 1, want_dact=True)]
    return ops.bag_project(x, hs, act=2)

hs = run()
torch.cuda.synchronize()
for h, w, b in ((hs[0], wt, bt), (hs[1], ws, bs)):
    pre = x.double() @ w.double().t() + b.double()
    ref = torch.nn.functional.gelu(pre)
    err = (h.out.double() - ref).abs().max().item()
    print("max abs err H:", err, "scale", ref.abs().max().item())
pre = x.double() @ ws.double().t() + bs.double()
gref = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * 3.141592653589793) ** 0.5
print("max abs err dact (fp16):", (hs[1].dact.double() - gref).abs().max().item())
hd = run(0.25)
torch.cuda.synchronize()
kept = hd[1].out != 0
print("keep rate", kept.float().mean().item(), "row spread", (kept.float().mean(1) - 0.75).abs().max().item(), "col spread",
      (kept.float().mean(0) - 0.75).abs().max().item())
nz = hs[1].out != 0
print("kept values scaled:", ((hd[1].out - hs[1].out / 0.75).abs() * kept).max().item())
print("teacher/student masks differ:", (kept != (hd[0].out != 0)).float().mean().item())
# timing: 20 launches per captured graph (host launch cost would otherwise hide the kernel)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3
outs = [torch.empty(N, E, device=dev), torch.empty(N, E, device=dev)]
dact = torch.empty(N, E, device=dev, dtype=torch.float16)
def runp(p):
    hs = [ops.ProjHead(wtp, bt, drop_p=p, drop_seed=5, out=outs[0]), ops.ProjHead(wsp, bs, drop_p=p, drop_seed=6, out=outs[1], want_dact=True, dact=dact)]
    ops.bag_project(x, hs, act=2)
for p in (0.0, 0.25):
    print(f"bag_project p={p}: {timed(lambda: runp(p)):.1f} us")
xp = ops.pair_planes(x)
out = torch.empty(N, E, device=dev); da = torch.empty(N, E, device=dev)
def old():
    ops.gemm_nt(xp, wtp, out=out, bias=bt, act=2, drop_p=0.25, drop_seed=3, prec="bf16x3", paired=True)
    ops.gemm_nt(xp, wsp, out=out, bias=bs, act=2, drop_p=0.25, drop_seed=4, prec="bf16x3", paired=True, dact=da)
print(f"two feat_gemm launches: {timed(old):.1f} us (+ pair_planes of X)")
