"""mhimx_bmm_chain against the launch-per-product path: values, time (eager + hipGraph replay) and, with CH_PROF=1 and a -DCH_PROF=<step>
build (MHIMX_LIB_NAME), the phase stamps of one step."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from mhim_mil_amd import nystrom as ny
torch.manual_seed(0)
dev = torch.device("cuda")
lm = torch.randn(256, 2 * 512, device=dev) * 0.5
dz = torch.randn(8, 256, 256, device=dev) * 0.1
res = {}
for flag in (False, True):
    ny._CHAIN = flag
    a2, z, z0, stats, chain = ny._landmark_pinv_forward(lm, 0.125)
    dlm = torch.zeros_like(lm)
    ny._landmark_pinv_backward(lm, 0.125, a2, z0, stats, chain, dz.clone(), dlm, accumulate=False)
    torch.cuda.synchronize()
    res[flag] = (z, dlm)
print("z   max diff", (res[0][0] - res[1][0]).abs().max().item(), "scale", res[0][0].abs().max().item())
print("dlm max diff", (res[0][1] - res[1][1]).abs().max().item(), "scale", res[0][1].abs().max().item())
print("gave up:", ny.chain_gave_up(dev), " counters", ny._CTRS[list(ny._CTRS)[0]][:513].sum().item())


def fb():
    a2, z, z0, stats, chain = ny._landmark_pinv_forward(lm, 0.125)
    return a2, z, z0, stats, chain


def bw(saved, dlm):
    a2, z, z0, stats, chain = saved
    ny._landmark_pinv_backward(lm, 0.125, a2, z0, stats, chain, dz, dlm, accumulate=False)


for flag in (False, True):
    ny._CHAIN = flag
    for name, fn in (("forward", lambda: fb()), ("forward+backward", lambda: bw(fb(), torch.empty_like(lm)))):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                keep = fn()
        g.replay(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): g.replay()
        torch.cuda.synchronize(); print("chain   " if flag else "launches", name, "graph replay %.1f us" % ((time.perf_counter() - t) / 100 * 1e6))
if os.environ.get("CH_PROF"):
    import numpy as np
    ny._CHAIN = True
    sv = fb()
    if os.environ.get("CH_PROF") == "bwd": bw(sv, torch.empty_like(lm))          # the stamps of the LAST launch (backward, second half) remain
    torch.cuda.synchronize()
    c = ny._CTRS[list(ny._CTRS)[0]]
    st = c[576:576 + 192].cpu().numpy().view(np.uint64).reshape(8, 12).astype(np.int64)
    order = [0, 1, 2, 3, 4, 8, 9, 10, 5, 6, 7]
    names = ["poll", "own DMA landed", "barrier (all DMA)", "mfma issue", "barrier (all mfma)", "partials + barrier", "reduce + tiles", "barrier", "stores + wait + barrier", "atomic"]
    st = st[:, order]
    d = np.diff(st, axis=1)
    print("shader-clock cycles per phase of step CH_PROF, workgroups 0..7:")
    for i, n in enumerate(names): print(f"  {n:24s}", d[:, i].tolist())
    print("  total", (st[:, -1] - st[:, 0]).tolist())
# stress: the hand-off with other work running beside it - every repetition must reproduce the first bit for bit
if os.environ.get("CH_STRESS"):
    ny._CHAIN = True
    n_rep = int(os.environ["CH_STRESS"])
    side = torch.cuda.Stream()
    big = torch.randn(8192, 8192, device=dev)
    ref = None
    bad = 0
    for rep in range(n_rep):
        if rep % 3 != 0:
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 4): big2 = big @ big if rep % 2 else big * 1.0001
        sv = fb(); d = torch.empty_like(lm); bw(sv, d)
        torch.cuda.synchronize()
        cur = (sv[1].clone(), d.clone())
        if ref is None: ref = cur
        elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])): bad += 1
    print("stress:", n_rep, "repetitions,", bad, "differ from the first; counters", ny._CTRS[list(ny._CTRS)[0]][:513].sum().item(), "gave up:", ny.chain_gave_up(dev))
    import hashlib
    print("digest", hashlib.sha1(ref[0].cpu().numpy().tobytes() + ref[1].cpu().numpy().tobytes()).hexdigest())
