import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from mhim_mil_amd import synth
from mhim_mil_amd.engine import FusedTrainer
import test_window_gpu as TW
DEV = "cuda"
n, d, acc = 1500, 256, 8
base = synth.mhim_state(7, input_dim=d, merge_k=5)
cfg = dict(TW.V2, dropout=0.25)
xs = [torch.from_numpy(synth.bag(900 + j, n, d)).to(DEV)[None] for j in range(acc)]
ls = [torch.tensor([j % 2], device=DEV) for j in range(acc)]
res = []
for rep in range(6):
    torch.manual_seed(5)
    s, t = TW._mk(base, d, **cfg), TW._mk(synth.spread_teacher(base), d, **cfg)
    tr = FusedTrainer(s, t, accumulation_steps=acc)
    logits, _ = tr.window_step(xs, ls, n_streams=1, update=False)
    torch.cuda.synchronize()
    res.append(tr.flat.grad.cpu().clone())
    if rep:
        dd = (res[rep] - res[0]).abs()
        idx = torch.nonzero(dd > 1e-6 * res[0].abs().max()).flatten().tolist()
        names = []
        for i in idx[:10]:
            for nm, off in tr.flat.offsets.items():
                pn = dict(s.named_parameters())[nm].numel() if nm in dict(s.named_parameters()) else 0
                if off <= i < off + pn:
                    names.append((nm, i - off, float(res[0][i]), float(res[rep][i])))
        print(rep, len(idx), names, flush=True)
