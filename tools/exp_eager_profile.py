"""Host-side profile of the eager native step (FusedTrainer.train_step without graphs): where the ~0.45 ms of Python go."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import FusedTrainer
N, D = 10000, 1024
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
dev = torch.device("cuda", 0)
base = synth.mhim_state(7, input_dim=D, merge_k=5)
def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline="attn", **CFG)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.to(dev).train()
g = torch.Generator(device=dev); g.manual_seed(5)
bags = [torch.randn(N, D, device=dev, generator=g).abs_() for _ in range(4)]
label = torch.tensor([1], device=dev)
tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
for i in range(10):
    tr.train_step(bags[i % 4], label)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    tr.train_step(bags[i % 4], label)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
t0 = time.perf_counter()
for i in range(200):
    tr.train_step(bags[i % 4], label)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue time alone: {(t1 - t0) / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    tr.train_step(bags[i % 4], label)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
