import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
torch.manual_seed(0)
for (N, n_rows, E, D) in ((64, 64, 128, 256), (640, 640, 128, 256), (2000, 1777, 512, 1024)):
    x = torch.randn(N, D, device="cuda"); dH = torch.randn(N, E, device="cuda"); dact = torch.ones(N, E, device="cuda").half()
    dW, db = ops.bag_wgrad(dH, dact, x, None, n_rows)
    ref = dH[:n_rows].double().t() @ x[:n_rows].double()
    err = (dW.double() - ref).abs()
    print((N, n_rows, E, D), "err", float(err.max()), "scale", float(ref.abs().max()), "nan", int(torch.isnan(dW).sum()), "bad", int((err > 1e-3 * ref.abs().max()).sum()))
N, E, D = 64, 128, 256
for l in (0, 3, 4, 9, 31, 32, 40, 63):
    dH2 = torch.zeros(N, E, device="cuda"); dH2[l, 3] = 1.0
    x2 = torch.zeros(N, D, device="cuda"); x2[:, 9] = torch.arange(N, device="cuda").float() + 1; x2[:, 200] = 100 + torch.arange(N, device="cuda").float()
    o, _ = ops.bag_wgrad(dH2, torch.ones(N, E, device="cuda").half(), x2, None, N)
    print("l", l, "->", float(o[3, 9]), float(o[3, 200]), "expect", l + 1, 100 + l, "nonzero", (o.abs() > 0.5).nonzero().tolist()[:6])
