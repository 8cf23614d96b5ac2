"""Host-side cost of the exchanges of the multi-GPU steps, measured on ONE GPU (an 8-GPU node is not ours to launch on): a one-rank
RCCL group still walks the whole enqueue path of torch.distributed + RCCL (argument checks, stream bookkeeping, kernel / copy launch), which
is what sits on the step's critical path between two graph segments.  Prints host microseconds per call (time.perf_counter around the
call, no device sync inside the loop) and device microseconds per call (events) for the buffers the steps exchange:
  c4 data parallel : the flat gradient (6.6 MB), in two pieces (4.6 MB + 2 MB)
  c5 sharded bag   : pool partial (E+2 floats) all-gather x2, scores all-gather (N floats), [R,E] merge block all-reduce, [k,E] token
                     gradient all-reduce, flat gradient all-reduce
usage: python tools/exp_allreduce_host.py   (one process, one GPU)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = "cuda"


def bench(name, fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:58s} host {1e6 * (t1 - t0) / iters:7.1f} us/call   device {1e3 * e0.elapsed_time(e1) / iters:7.1f} us/call")


for label, nfl in (("all_reduce flat gradient 6.6 MB", 1650000), ("all_reduce gradient piece 4.6 MB", 1150000), ("all_reduce gradient piece 2 MB", 524288),
                   ("all_reduce merge block [19700, 512] (c5)", 19700 * 512), ("all_reduce token gradient [5, 512]", 5 * 512)):
    t = torch.zeros(nfl, device=dev)
    bench(label, lambda t=t: dist.all_reduce(t))
for label, nfl in (("all_gather pool partial 514 floats", 514), ("all_gather scores 25 000 floats / rank (c5)", 25000)):
    src, out = torch.zeros(nfl, device=dev), torch.zeros(nfl, device=dev)
    bench(label, lambda src=src, out=out: dist.all_gather_into_tensor(out, src))
g = torch.cuda.CUDAGraph()
x = torch.zeros(1024, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    x.add_(1.0)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    x.add_(1.0)
bench("hipGraph launch (one tiny kernel) for comparison", g.replay)
dist.destroy_process_group()
