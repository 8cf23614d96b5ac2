"""Runs the c2-shaped FusedTrainer for a few steps several times and compares the final parameters bit for bit (run it while another
process loads the GPU: timing-dependent results show up as differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhim_mil_amd import synth
from mhim_mil_amd.engine import FusedTrainer
from mhim_mil_amd.mhim import MHIM
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
base = synth.mhim_state(7, input_dim=D, merge_k=5)
g = torch.Generator(device="cuda").manual_seed(3)
bags = [torch.randn(N, D, device="cuda", generator=g).abs_() for _ in range(3)]
def run():
    def mk():
        m = MHIM(input_dim=D, n_classes=2, baseline="attn", **cfg)
        sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        m = m.cuda().train(); m.merge.dropout = 0.0
        return m
    torch.manual_seed(5)
    tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
    for i in range(6):
        tr.train_step(bags[i % 3], torch.tensor([i % 2], device="cuda"))
    torch.cuda.synchronize()
    return tr.flat.student.cpu().numpy(), tr.flat.grad.cpu().numpy()
ref = run()
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    out = run()
    d = np.abs(out[0] - ref[0])
    print(f"run {k}: params differ in {(d > 0).sum()} of {d.size} elements, max {d.max():.3e}")
