"""Does the order in which a multi-branch hipGraph's kernel nodes were CREATED (captured) decide how the branches overlap at replay?
Four branches of K kernels of ~20 us each (one workgroup-limited kernel per node), captured branch after branch vs round-robin."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda"
K, S = int(os.environ.get("K", 34)), 4
xs = [torch.randn(64, 4096, device=dev) for _ in range(S)]          # a small kernel: few workgroups, ~10+ us through repetition
streams = [torch.cuda.Stream() for _ in range(S)]


def work(x):
    torch.sin_(x)


def capture(interleaved):
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        for x in xs: work(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cs):
            ev = torch.cuda.Event(); ev.record(cs)
            for s in streams: s.wait_event(ev)
            if interleaved:
                for k in range(K):
                    for s, x in zip(streams, xs):
                        with torch.cuda.stream(s): work(x)
            else:
                for s, x in zip(streams, xs):
                    with torch.cuda.stream(s):
                        for k in range(K): work(x)
            for s in streams:
                e = torch.cuda.Event(); e.record(s); cs.wait_event(e)
    return g


for name, il in (("branch after branch", False), ("round-robin", True), ("branch after branch", False), ("round-robin", True)):
    g = capture(il)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print(f"{name:20s}: {(time.perf_counter() - t) / 20 * 1e6:8.1f} us per replay ({S} branches x {K} kernels)")
# one branch alone, for scale
g = torch.cuda.CUDAGraph(); cs = torch.cuda.Stream()
with torch.cuda.stream(cs):
    with torch.cuda.graph(g, stream=cs):
        for k in range(K): work(xs[0])
for _ in range(3): g.replay()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print(f"one branch alone    : {(time.perf_counter() - t) / 20 * 1e6:8.1f} us per replay")
