"""PCIe-inclusive rate of the c2 step: the reference's trainer hands each bag over from host memory.  Measures (a) the plain
H2D copy of one 41 MB bag from pinned memory, (b) steps with the NEXT bag's copy overlapped on a second stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mhim_mil_amd.engine import FusedTrainer
dev = torch.device("cuda", 0)
student, teacher, _ = bench.make_models(dev, "auto")
tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997)
N, D = bench.N_INST, bench.D_IN
host = [torch.randn(N, D).abs_().pin_memory() for _ in range(4)]
dbuf = [torch.empty(N, D, device=dev) for _ in range(2)]
lab = torch.tensor([1], device=dev)
graphs = [tr.capture(dbuf[i], lab, warmup=1) for i in range(2)]
copy_s = torch.cuda.Stream()
# (a) copy alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): dbuf[i % 2].copy_(host[i % 4], non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"H2D of one {N*D*4/1e6:.1f} MB bag (pinned): {dt*1e3:.3f} ms = {N*D*4/dt/1e9:.1f} GB/s")
# (b) double-buffered: copy bag i+1 on the copy stream while the graph of bag i replays
ev_copy = [torch.cuda.Event() for _ in range(2)]; ev_done = [torch.cuda.Event() for _ in range(2)]
main = torch.cuda.current_stream()
steps = 200
dbuf[0].copy_(host[0]); ev_copy[0].record(main)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    b = i % 2
    with torch.cuda.stream(copy_s):
        copy_s.wait_event(ev_done[1 - b])                       # the other buffer's last step has finished
        dbuf[1 - b].copy_(host[(i + 1) % 4], non_blocking=True)
        ev_copy[1 - b].record(copy_s)
    main.wait_event(ev_copy[b])
    graphs[b].replay()
    ev_done[b].record(main)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"step with the next bag's H2D overlapped: {dt*1e3:.3f} ms/step = {N/dt/1e6:.2f} M patch-instances/s (PCIe-inclusive)")
