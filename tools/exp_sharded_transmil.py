"""World-1 timing of the sharded MHIM(TransMIL) step (sharded_transmil.py) beside FusedTrainer's TransMIL step at c3 size: what the
sequence-parallel path costs per rank before any exchange."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import FusedTrainer
from mhim_mil_amd.sharded import ShardedBagTrainer

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


g = torch.Generator(device=dev); g.manual_seed(5)
bags = [torch.randn(N, D, device=dev, generator=g).abs_() for _ in range(2)]
label = torch.tensor([1], device=dev)


def timed(fn, name, steps=6):
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:60s} {dt * 1e3:8.3f} ms/step", flush=True)


if not os.environ.get("SHARDED_ONLY"):
    tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
    timed(lambda i: tr.train_step(bags[i % 2], label), "FusedTrainer.train_step (TransMIL), eager")
    g = [tr.capture(b, label, warmup=2) for b in bags]
    timed(lambda i: g[i % 2].replay(), "FusedTrainer (TransMIL), hipGraph replay")
st = ShardedBagTrainer(mk(), mk(), aux_alpha=0.5)
timed(lambda i: st.train_step(bags[i % 2], label), "ShardedBagTrainer.train_step (TransMIL), world 1, eager")
