"""Phase stamps of the wave-specialised projection (library built with -DPW_PROF, MHIMX_LIB_NAME): shader cycles per phase and k-step of
consumer waves 0 (group 0) and 4 (group 1) and producer wave 8 of the first workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
dev = "cuda"
N, D, E = 10000, int(os.environ.get("D", 1024)), 512
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g).abs_()
wt = torch.randn(E, D, device=dev, generator=g) * 0.036
wtp = ops.pair_planes(wt)
for rep in range(3):
    hs = [ops.ProjHead(wtp, None, drop_p=0.25, drop_seed=5), ops.ProjHead(wtp, None, drop_p=0.25, drop_seed=6, want_dact=True)]
    ops.bag_project(x, hs, act=2)
    torch.cuda.synchronize()
nk = D // 32
cn = ["load phase (18 reads + wait)", "barrier", "compute phase (60 MFMA issue)", "barrier"]
pn = ["slot A: split+store 3 units", "A loads + 4 DMA issue", "lgkm wait", "barrier", "slot B: split+store 2 units", "A loads + 4 DMA + lgkm", "vmcnt(13) wait", "barrier"]
for w, names in ((0, cn), (4, cn), (8, pn)):
    v = hs[0].out[w, :20].cpu().tolist()
    print(f"wave {w}: entry->loop {v[8]:.0f}, main loop {v[9]:.0f}, epilogue {v[10]:.0f} cycles")
    print(f"wave {w}: " + "; ".join(f"{n}: {c / nk:.0f}" for n, c in zip(names, v)), " | per k-step:", round(sum(v[:len(names)]) / nk))
