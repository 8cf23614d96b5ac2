"""A few eager windows (8 bags, c2 shape) through FusedTrainer.window_step - the command the window's counter passes profile
(tools/pmc.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).   python tools/exp_window_eager.py [windows=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from mhim_mil_amd.engine import FusedTrainer

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 6
k = 8
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
g = torch.Generator(device=dev); g.manual_seed(2000)
bags = [torch.randn(B.N_INST, B.D_IN, device=dev, generator=g).abs_() for _ in range(k)]
labels = [torch.tensor([i % 2], device=dev) for i in range(k)]
student, teacher, _ = B.make_models(dev, "auto")
tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997, accumulation_steps=k)
for _ in range(n_win):
    tr.window_step(bags, labels)
torch.cuda.synchronize()
print("windows:", n_win, "batched:", tr.last.get("ws") is not None)
