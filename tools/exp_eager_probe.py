import os, sys, time, types
sys.path.insert(0, "/root/repo")
import torch
import bench as B
from mhim_mil_amd.engine import FusedTrainer, CommonMIL
from mhim_mil_amd.optim import FusedAdamEMA
dev = torch.device("cuda", 0)
bags = [torch.randn(B.N_INST, B.D_IN, device=dev).abs_() for _ in range(4)]
label = torch.tensor([1], device=dev)
def timed(fn, name, n=50):
    for i in range(5): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); print(f"{name:50s} {(time.perf_counter()-t0)/n*1e3:.3f} ms", flush=True)
s, t, _ = B.make_models(dev, "auto")
tr = FusedTrainer(s, t, aux_alpha=0.5)
timed(lambda i: tr.train_step(bags[i % 4], label), "A: fresh FusedTrainer eager")
if os.environ.get("MODE") == "opt":
    s2, t2, _ = B.make_models(dev, "auto")
    opt = FusedAdamEMA(s2, t2)
    timed(lambda i: (opt.trainer.forward_backward(bags[i % 4], label), opt.step()), "B: FusedAdamEMA.trainer fb + step")
    timed(lambda i: tr.train_step(bags[i % 4], label), "A again after B")
s3, t3, _ = B.make_models(dev, "auto")
tr3 = FusedTrainer(s3, t3, aux_alpha=0.5)
timed(lambda i: tr3.train_step(bags[i % 4], label), "C: second fresh FusedTrainer eager")
timed(lambda i: tr.train_step(bags[i % 4], label), "A again")
