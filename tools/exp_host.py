"""How long does the HOST take to enqueue one train step (Python + ctypes + torch allocator)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mhim_mil_amd.engine import FusedTrainer
dev = torch.device("cuda", 0)
s, t, base = bench.make_models(dev, "auto")
tr = FusedTrainer(s, t)
bags = [torch.randn(10000, 1024, device=dev).abs_() for _ in range(4)]
lab = torch.tensor([1], device=dev)
for i in range(5): tr.train_step(bags[i % 4], lab)
torch.cuda.synchronize()
# enqueue-only time: few steps so the queue never fills
ts = []
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(bags[rep % 4], lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("enqueue ms / total ms per step:", [(round(a * 1e3, 3), round(b * 1e3, 3)) for a, b in ts])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20): tr.train_step(bags[i % 4], lab)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
