"""One forward_backward of the small-shape FusedTrainer many times: which intermediate differs run to run under GPU contention?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhim_mil_amd import synth
from mhim_mil_amd.engine import FusedTrainer
from mhim_mil_amd.mhim import MHIM
D, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
base = synth.mhim_state(7, input_dim=D, merge_k=5)
g = torch.Generator(device="cuda").manual_seed(3)
bag = torch.randn(N, D, device="cuda", generator=g).abs_()
def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline="attn", **cfg)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    m = m.cuda().train(); m.merge.dropout = 0.0
    return m
def run():
    torch.manual_seed(5)
    tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
    tr.forward_backward(bag, torch.tensor([1], device="cuda"))
    torch.cuda.synchronize()
    names = list(tr.flat.grad_views.keys())
    out = {"logits": tr.last["logits"].cpu().numpy().copy(), "score": tr.last["score"].cpu().numpy().copy(),
           "rows": tr.last["rows"].cpu().numpy().copy(), "gq": tr.s.merge.global_q_mm.detach().cpu().numpy().copy()}
    for n in names:
        out["g:" + n] = tr.flat.grad_views[n].cpu().numpy().copy()
    return out
ref = run()
bad = {}
for k in range(R):
    o = run()
    for key in ref:
        if o[key].shape != ref[key].shape or not np.array_equal(o[key], ref[key]):
            d = float(np.abs(o[key].astype(np.float64) - ref[key]).max()) if o[key].shape == ref[key].shape else -1
            bad.setdefault(key, []).append((k, d))
            if o[key].shape == ref[key].shape and o[key].ndim == 2 and len(bad[key]) <= 3:
                rr, cc = np.nonzero(o[key] != ref[key])
                print(key, "run", k, "rows", sorted(set(rr.tolist()))[:40], "n_rows", len(set(rr.tolist())), "cols", len(set(cc.tolist())), "min/max col", cc.min(), cc.max())
print("differing:", {k: v[:3] for k, v in bad.items()} if bad else "none", "in", R, "runs")
