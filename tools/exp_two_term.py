"""can the STUDENT half of the projection (and dW1) run with TWO bf16/fp16 terms instead of three?
Measures the forward errors of the 2-term form the library has (F16S: activation in one fp16 term, weight hi + lo) against the default
3-term bf16 form and fp64, on the quantities the tolerances are stated on: bag logits (1e-4), the pooled bag feature, instance scores
(they feed a top-k).  Diffuse attention (plain init) and peaked attention (scorer sharpened x20: synth.spread_teacher)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from oracle import mhim_oracle as O

n, d = 10000, 1024
cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5, merge_mm=0.9999,
           merge_ratio=0.9, temp_t=0.1, dropout=0.0)
base = synth.mhim_state(7, input_dim=d, merge_k=5)
x = torch.from_numpy(synth.bag(2000 + n, n, d))
for name, sd in (("diffuse (plain init)", base), ("peaked (scorer x20, predictor x50)", synth.spread_teacher(base))):
    p64 = {k: torch.as_tensor(np.asarray(v)).double() for k, v in sd.items()}
    feat64, score64 = O.forward_teacher(x.double(), p64, O.Cfg(**cfg))
    logits64 = O.forward_test(x.double(), p64, O.Cfg(**cfg))
    for prec in ("auto", "f16s"):
        m = MHIM(input_dim=d, n_classes=2, baseline="attn", prec=prec, **cfg)
        s2 = dict(sd); s2["merge.global_q"] = s2["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in s2.items()})
        m = m.cuda().eval()
        feat, score = m.forward_teacher(x.cuda())
        logits = m.forward_test(x.cuda())
        ef = (feat.cpu().double().view(-1) - feat64.view(-1)).abs().max().item() / feat64.abs().max().item()
        es = (score.cpu().double().view(-1) - score64.view(-1)).abs().max().item()
        el = (logits.cpu().double().view(-1) - logits64.view(-1)).abs().max().item()
        k = 600
        top = set(np.argsort(-score64.view(-1).numpy(), kind="stable")[:k].tolist())
        got = set(np.argsort(-score.cpu().view(-1).double().numpy(), kind="stable")[:k].tolist())
        print(f"{name:38s} prec={prec:5s}: logits err {el:.2e}  bag-feature rel err {ef:.2e}  score err {es:.2e}  top-{k} set differs in {len(top - got)} ids")
