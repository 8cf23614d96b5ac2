"""Time one MHIM(TransMIL) train step at BASELINE config c3 (N=50 000, D=1024) — eager launches."""
import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import FusedTrainer

N = int(os.environ.get("N", 50000)); D = 1024; STEPS = int(os.environ.get("STEPS", 10))
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
dev = torch.device("cuda", 0)
base = synth.mhim_state(7, input_dim=D, merge_k=5, baseline="selfattn")
def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline="selfattn", **CFG)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.to(dev).train()
s, t = mk(), mk()
tr = FusedTrainer(s, t, aux_alpha=0.5)
g = torch.Generator(device=dev); g.manual_seed(5)
bags = [torch.randn(N, D, device=dev, generator=g).abs_() for _ in range(2)]
lab = torch.tensor([1], device=dev)
GRAPH = int(os.environ.get("GRAPH", 0))
if GRAPH:
    graphs = [tr.capture(bags[i], lab, warmup=2) for i in range(2)]
    step = lambda i: graphs[i % 2].replay()
else:
    step = lambda i: tr.train_step(bags[i % 2], lab)
for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
print(f"c3 N={N} graph={GRAPH}: {dt*1e3:.2f} ms/step  {N/dt/1e6:.2f} M inst/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
