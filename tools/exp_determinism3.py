"""One forward_backward of the MHIM(TransMIL) trainer many times: bit-reproducibility of every gradient (run under GPU contention too)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhim_mil_amd import synth
from mhim_mil_amd.engine import FusedTrainer
from mhim_mil_amd.mhim import MHIM
D, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bl = sys.argv[4] if len(sys.argv) > 4 else "selfattn"
cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
base = synth.mhim_state(7, input_dim=D, merge_k=5, baseline=bl)
g = torch.Generator(device="cuda").manual_seed(3)
bag = torch.randn(N, D, device="cuda", generator=g).abs_()
def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline=bl, **cfg)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    m = m.cuda().train(); m.merge.dropout = 0.0
    return m
def run():
    torch.manual_seed(5)
    tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
    tr.forward_backward(bag, torch.tensor([1], device="cuda"))
    torch.cuda.synchronize()
    out = {"logits": tr.last["logits"].cpu().numpy().copy()}
    for n, v in tr.flat.grad_views.items():
        out["g:" + n] = v.cpu().numpy().copy()
    return out
ref = run()
bad = {}
for k in range(R):
    o = run()
    for key in ref:
        if not np.array_equal(o[key], ref[key]):
            bad.setdefault(key, []).append((k, float(np.abs(o[key].astype(np.float64) - ref[key]).max())))
print(bl, "differing:", {k: v[:2] for k, v in bad.items()} if bad else "none", "in", R, "runs")
