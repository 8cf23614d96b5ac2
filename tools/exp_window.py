"""Accumulation-window timing on the GPU box: ms per bag of FusedTrainer.window_step (captured) for several stream counts,
beside the one-bag step.   python tools/exp_window.py [k=8] [streams list]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from mhim_mil_amd.engine import FusedTrainer

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
g = torch.Generator(device=dev); g.manual_seed(2000)
bags = [torch.randn(B.N_INST, B.D_IN, device=dev, generator=g).abs_() for _ in range(2 * k)]
labels = [torch.tensor([i % 2], device=dev) for i in range(2 * k)]


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


student, teacher, _ = B.make_models(dev, "auto")
tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997)
gs = [tr.capture(bags[i], labels[i], warmup=1) for i in range(4)]
it = [0]
def one():
    gs[it[0] % 4].replay(); it[0] += 1
print(f"one-bag graph step: {1e3 * timeit(one, 200):.4f} ms/bag", flush=True)
for S in streams:
    student, teacher, _ = B.make_models(dev, "auto")
    tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997, accumulation_steps=k)
    try:
        wins = [tr.capture_window(bags[w * k:(w + 1) * k], labels[w * k:(w + 1) * k], warmup=1, n_streams=S) for w in range(2)]
    except Exception as e:
        print(f"streams={S}: capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
        continue
    it = [0]
    def win():
        wins[it[0] % 2].replay(); it[0] += 1
    dt = timeit(win, 50)
    print(f"window k={k} streams={S}: {1e3 * dt:.4f} ms/window = {1e3 * dt / k:.4f} ms/bag = {B.N_INST * k / dt / 1e6:.1f} M inst/s", flush=True)
    if os.environ.get("EAGER"):
        dt = timeit(lambda: tr.window_step(bags[:k], labels[:k], n_streams=S), 10)
        print(f"   eager: {1e3 * dt / k:.4f} ms/bag", flush=True)
