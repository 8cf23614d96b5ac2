"""A/B: projection + one-pass scorer vs the scored projection + finalize (graph replays, c2 shapes)."""
import sys, torch
sys.path.insert(0, ".")
from mhim_mil_amd import ops, _lib as L
from mhim_mil_amd import synth
from tests.test_ops_gpu import rnd
DEV = "cuda:0"
n, d, E, A, Cc = 10000, 1024, 512, 128, 2
x = torch.from_numpy(synth.bag(5100, n, d)).to(DEV)
w1 = rnd(201, (E, d), std=0.03).to(DEV); b1 = rnd(202, (E,), std=0.1).to(DEV)
w1s = rnd(203, (E, d), std=0.03).to(DEV); b1s = rnd(204, (E,), std=0.1).to(DEV)
wa = rnd(205, (A, E), std=0.05).to(DEV); wc = (rnd(206, (1, A), std=0.3) * 5).to(DEV)
wp = rnd(207, (Cc, E), std=0.05).to(DEV); bp = rnd(208, (Cc,), std=0.1).to(DEV)
act, sact = L.ACT["relu"], L.ACT["tanh"]
tick = torch.zeros(1, dtype=torch.int64, device=DEV)
pw, pws = ops.pair_planes(w1), ops.pair_planes(w1s)
frag = torch.empty_like(wa); ops.prep_batch([(ops.PREP_FRAG, wa, frag)])
sc = ops.ScorerW(wa, wc, sact, wa_frag=frag)
img = torch.empty(144 * E, device=DEV)
ops.prep_batch([(ops.PREP_FRAG16, wa, img), (ops.PREP_FRAG16, wp, img[128 * E:])])
buf = ops.ProjScoreBuf(n, img, wc.view(-1).contiguous(), sact, Cc, x.device)

def heads():
    return [ops.ProjHead(pw, b1, drop_p=0.25, drop_seed=777), ops.ProjHead(pws, b1s, drop_p=0.25, drop_seed=778, want_dact=True)]
hA, hB = heads(), heads()
def A_():
    ops.bag_project(x, hA, act=act, drop_tick=tick)
    ops.abmil_pool_fwd(sc, hA[0].out, None, wp=wp, bp=bp)
def B_():
    ops.bag_project(x, hB, act=act, drop_tick=tick, score0=buf)
    buf.finalize(bp)
def P_():
    ops.bag_project(x, hA, act=act, drop_tick=tick)
xs = [torch.from_numpy(synth.bag(5200 + j, n, d)).to(DEV) for j in range(8)]
cnt = [0]
def PC_():                                   # a different (cold) bag per launch
    cnt[0] += 1
    ops.bag_project(xs[cnt[0] % 8], hA, act=act, drop_tick=tick)
scrub = torch.empty(96 << 20, device=DEV)    # 384 MB
def PS_():                                   # the same bag, 384 MB of other traffic in between
    scrub.add_(1.0)
    ops.bag_project(x, hA, act=act, drop_tick=tick)
def S_():
    scrub.add_(1.0)
def time(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print(f"{name}: {best:.1f} us")
for _ in range(2):
    time(P_, "projection only (2 heads)")
    time(PC_, "projection, rotating bags")
    time(S_, "scrub only")
    time(PS_, "scrub + projection")
    time(A_, "projection + one-pass scorer")
    time(B_, "scored projection + finalize")
