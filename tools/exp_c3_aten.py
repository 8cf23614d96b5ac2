"""Which torch (aten) operators still launch kernels in one eager c3 train step, with shapes and the python frame that called them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import FusedTrainer

N, D = int(os.environ.get("N", 50000)), 1024
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
dev = torch.device("cuda", 0)
base = synth.mhim_state(7, input_dim=D, merge_k=5, baseline="selfattn")
def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline="selfattn", **CFG)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.to(dev).train()
tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
g = torch.Generator(device=dev); g.manual_seed(5)
bag = torch.randn(N, D, device=dev, generator=g).abs_()
lab = torch.tensor([1], device=dev)
for _ in range(2):
    tr.train_step(bag, lab)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=True) as prof:
    tr.train_step(bag, lab)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if e.name.startswith("aten::") and t > 0:
        st = [f for f in (e.stack or []) if "mhim_mil_amd" in f or "tools/" in f]
        rows.append((t, e.name, "", st[0] if st else "?"))
rows.sort(reverse=True)
for t, n, s, f in rows[:40]:
    print(f"{t:8.1f} us  {n:28s} {f}")
print("total aten device time", sum(r[0] for r in rows))
