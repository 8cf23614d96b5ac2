import torch
x = torch.randn(4096, 4096, device="cuda")
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print("no external kw:", ex); raise SystemExit
y = x @ x
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a = x * 2
    e0.record()
    y = x @ x
    e1.record()
    b = y + 1
for _ in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("elapsed ms inside graph:", e0.elapsed_time(e1))
