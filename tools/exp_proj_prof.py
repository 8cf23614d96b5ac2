"""Phase stamps of the wave-specialised projection (library built with -DPW_PROF, MHIMX_LIB_NAME): shader cycles per phase and k-step of
consumer waves 0 (group 0) and 4 (group 1) and producer wave 8 of the first workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
dev = "cuda"
N, D, E = int(os.environ.get("N", 10000)), int(os.environ.get("D", 1024)), 512
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g).abs_()
wt = torch.randn(E, D, device=dev, generator=g) * 0.036
wtp = ops.pair_planes(wt)
for rep in range(int(os.environ.get("REPS", 3))):
    P, ACT = float(os.environ.get("P", 0.25)), int(os.environ.get("ACT", 2))
    hs = [ops.ProjHead(wtp, None, drop_p=P, drop_seed=5), ops.ProjHead(wtp, None, drop_p=P, drop_seed=6, want_dact=ACT != 0)]
    ops.bag_project(x, hs, act=ACT)
    torch.cuda.synchronize()
nk = D // 32
M0 = int(os.environ.get("M0", 0))          # first row of the stamped workgroup (-DPW_PROF_BLOCK)
cn = ["load phase (18 reads + wait)", "barrier", "compute phase (60 MFMA issue)", "barrier"]
pn = ["slot A: split+store 3 units", "A loads + 4 DMA issue", "lgkm wait", "barrier", "slot B: split+store 2 units", "A loads + 4 DMA + lgkm", "vmcnt(13) wait", "barrier"]
for w, names in ((0, cn), (4, cn), (8, pn)):
    v = hs[0].out[M0 + w, :20].cpu().tolist()
    print(f"wave {w}: entry->loop {v[8]:.0f}, main loop {v[9]:.0f}, epilogue {v[10]:.0f} cycles")
    if w == 0 and v[13] > 0:                 # round 6: the constant 100 MHz clock over the same span -> the shader clock this launch held
        print(f"wave 0: {v[14]:.0f} shader cycles in {v[13] / 100:.2f} us -> {v[14] / v[13] / 10:.3f} GHz")
    print(f"wave {w}: " + "; ".join(f"{n}: {c / nk:.0f}" for n, c in zip(names, v)), " | per k-step:", round(sum(v[:len(names)]) / nk))
if os.environ.get("PE"):                                    # library built with -DPW_PROF=2: the epilogue's phases instead of the k loop's
    en = ["resid request + barrier (everyone out of the k loop)", "accumulators -> LDS tile", "barrier", "row loop (LDS -> act / dropout -> stores)"]
    for w in (0, 4, 8, 11):
        v = hs[0].out[M0 + w, :20].cpu().tolist()
        print(f"wave {w} epilogue (both halves): " + "; ".join(f"{n}: {c:.0f}" for n, c in zip(en, v)), f"| last stores left after {v[11]:.0f}")
