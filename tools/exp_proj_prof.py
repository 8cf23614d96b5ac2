"""Phase stamps of the ping-pong projection (library built with -DPJ_PP_ORDER=1 -DPJ_PP_PROF): cycles per slot phase, wave 0 / wave 4."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
dev = "cuda"
N, D, E = 10000, 1024, 512
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g).abs_()
wt = torch.randn(E, D, device=dev, generator=g) * 0.036
wtp = ops.pair_planes(wt)
for rep in range(3):
    hs = [ops.ProjHead(wtp, None, drop_p=0.25, drop_seed=5), ops.ProjHead(wtp, None, drop_p=0.25, drop_seed=6, want_dact=True)]
    ops.bag_project(x, hs, act=2)
    torch.cuda.synchronize()
names = ["load: reads+stores+A loads+lgkm wait", "load: DMA issue", "load: vmcnt wait", "barrier after load", "compute: 60 MFMA", "barrier after compute", "-", "-"]
for w in (0, 4):
    v = hs[0].out[160 + w, :8].cpu().tolist()
    print(f"wave {w}: " + "; ".join(f"{n}: {c / 32:.0f}" for n, c in zip(names, v) if n != "-"), " | sum per k-step:", sum(v) / 32)
