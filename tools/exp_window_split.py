"""Accumulation window as THREE kinds of graphs instead of one graph with branches (round 6 experiment):
  A   (main stream)      preparation + both projections of all k bags in one launch
  B_j (stream j, j < k)  bag j's middle: scorers, select, Merge, head, backward up to the dPre image (its weight gradient parked)
  C   (main stream)      ONE weight-gradient launch over the k images, reductions, the queries' EMA chain, Adam + EMA
The one-graph window (FusedTrainer.capture_window) never runs more than ~2 of its branches at a time (profiles/r06_window_timeline.md);
here every B_j is a graph of its own launched on its own stream.     python tools/exp_window_split.py [k=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MHIMX_WINDOW_PROJECT", "1")
os.environ.setdefault("MHIMX_WINDOW_WGRAD", "1")
import torch
import bench as B
from mhim_mil_amd import ops, mhim as mh
from mhim_mil_amd.engine import FusedTrainer

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
g = torch.Generator(device=dev); g.manual_seed(2000)
bags = [torch.randn(B.N_INST, B.D_IN, device=dev, generator=g).abs_() for _ in range(k)]
labels = [torch.tensor([i % 2], device=dev) for i in range(k)]

student, teacher, _ = B.make_models(dev, "auto")
tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997, accumulation_steps=k)
s, fl = tr.s, tr.flat
xs = [s._check_x(b) for b in bags]
st = tr._window_state(k, dev)
streams = [torch.cuda.Stream() for _ in range(k)]
km, E = s.merge.k, s.mlp_dim
S = {}


def part_a():
    S["prep_t"], S["preps"] = tr._nat_prep(xs, None, with_opt_tick=True)
    S["proj"] = [tr._nat_heads(x, S["prep_t"], pp) for x, pp in zip(xs, S["preps"])]
    ops.bag_project_multi(xs, [h for _, h in S["proj"]], act=mh.L.act_code(s.act, mh._FEATURE_ACTS), drop_tick=tr.tick)
    S["q_new"] = torch.empty((k, km, E), device=dev)
    S["park"] = [[] for _ in range(k)]
    S["tokens"] = [None] * k


def part_b(j):
    keep = tr._defer
    tr._defer = keep if j == 0 else st["defers"][j - 1]
    gv = fl.grad_views if j == 0 else st["views"][j - 1]
    try:
        tr._nat_bag(xs[j], labels[j], S["prep_t"], S["preps"][j], gv, accumulate=False, i=None, q_out=S["q_new"][j], slot=j,
                    wgrad_park=S["park"][j], projected=S["proj"][j])
        S["tokens"][j] = tr.last["tokens"]
    finally:
        tr._defer = keep


def part_c():
    park = [im for p in S["park"] for im in p]
    ops.bag_wgrad_multi(park, fl.grad_views["feature.0.weight"], accumulate=False, defer=tr._defer)
    ops.reduce_flush(tr._defer)
    tr._g_extra = st["slabs"][:k - 1]
    mm = float(s.merge.g_q_mm)
    w = ((1.0 - mm) * mm ** torch.arange(k - 1, -1, -1, device=dev, dtype=torch.float64)).float() if "w" not in S else S["w"]
    S["w"] = w
    Z = torch.stack(S["tokens"])
    q = s.merge.global_q_mm.data.view(km, E)
    q.mul_(mm ** k).add_((Z * w.view(k, 1, 1)).sum(0))
    tr._micro = k
    tr.update()


def eager():
    part_a()
    for j in range(k):
        part_b(j)
    part_c()


for _ in range(2):
    eager()
torch.cuda.synchronize()
pool = torch.cuda.graph_pool_handle()
cs = torch.cuda.Stream()
tr._capturing = True
gA = torch.cuda.CUDAGraph()
with torch.cuda.graph(gA, pool=pool, stream=cs):
    part_a()
gB = []
for j in range(k):
    gj = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gj, pool=pool, stream=cs):
        part_b(j)
    gB.append(gj)
gC = torch.cuda.CUDAGraph()
with torch.cuda.graph(gC, pool=pool, stream=cs):
    part_c()
tr._capturing = False
torch.cuda.synchronize()
evA = torch.cuda.Event()
evB = [torch.cuda.Event() for _ in range(k)]


def window(ns):
    main = torch.cuda.current_stream()
    gA.replay()
    evA.record(main)
    for j in range(k):
        sj = streams[j % ns]
        sj.wait_event(evA)
        with torch.cuda.stream(sj):
            gB[j].replay()
    for j in range(ns):
        evB[j].record(streams[j])
        main.wait_event(evB[j])
    gC.replay()


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for ns in (1, 2, 4, 8):
    if ns > k:
        continue
    dt = timeit(lambda: window(ns), 50)
    print(f"split window k={k} streams={ns}: {1e3 * dt:.4f} ms/window = {1e3 * dt / k:.4f} ms/bag = {B.N_INST * k / dt / 1e6:.1f} M inst/s", flush=True)
t0 = time.perf_counter()
for _ in range(50):
    window(8)
host = (time.perf_counter() - t0) / 50
torch.cuda.synchronize()
print(f"host issue time per window (8 streams): {1e3 * host:.4f} ms", flush=True)
