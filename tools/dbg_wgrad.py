import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from mhim_mil_amd import ops, _lib as L
torch.manual_seed(0)
N, n_rows, E, D = 64, 64, 128, 256
x = torch.randn(N, D, device="cuda")
dH = torch.randn(N, E, device="cuda")
dact = torch.ones(N, E, device="cuda").half()
lib = L.lib()
img = torch.zeros(lib.mhimx_wgrad_image_bytes(n_rows, E) // 4, device="cuda")
ob = torch.empty(E, device="cuda"); ws_b = torch.empty(2 * 2 * E, device="cuda")
L.check(lib.mhimx_rows_dpre_image(None, dH.data_ptr(), dact.data_ptr(), None, n_rows, E, img.data_ptr(), ob.data_ptr(), 0, ws_b.data_ptr(), ws_b.numel() * 4, None), "img")
torch.cuda.synchronize()
b = img.view(torch.bfloat16).view(n_rows // 32, E // 128, 4, 2, 128, 8).float()    # [ks][it][koct][hl][slot][8]
val = b[:, :, :, 0] + b[:, :, :, 1]                                                # [ks][it][koct][slot][8]
dec = torch.empty(n_rows, E, device="cuda")
for i in range(E):
    it, il = i // 128, i % 128
    slot = (il % 4) * 32 + il // 4
    dec[:, i] = val[:, it, :, slot, :].reshape(-1)
print("image err", float((dec - dH).abs().max()), "colsum err", float((ob - dH.sum(0)).abs().max()))
dW, db = ops.bag_wgrad(dH, dact, x, None, n_rows)
ref = dH.double().t() @ x.double()
err = (dW.double() - ref).abs()
print("wgrad err", float(err.max()), "scale", float(ref.abs().max()))
bad = (err > 1e-3).nonzero()
print("bad count", bad.shape[0], "of", E * D, "first", bad[:10].tolist())
# structure probe: which (i,n) does output (i', n') correspond to?  x = one-hot columns
for (i_probe, n_probe) in ((0, 0), (1, 0), (0, 1), (5, 7), (64, 130), (127, 255)):
    dH2 = torch.zeros(N, E, device="cuda"); dH2[:, i_probe] = 1.0
    x2 = torch.zeros(N, D, device="cuda"); x2[:, n_probe] = 1.0
    o, _ = ops.bag_wgrad(dH2, dact, x2, None, n_rows)
    nz = (o.abs() > 0.5).nonzero().tolist()
    print((i_probe, n_probe), "->", nz[:6], [float(o[a, b2]) for a, b2 in nz[:3]])
# k probe: only row l nonzero
for l in (0, 1, 4, 8, 31, 32, 63):
    dH2 = torch.zeros(N, E, device="cuda"); dH2[l, 3] = 1.0
    x2 = torch.zeros(N, D, device="cuda"); x2[:, 9] = torch.arange(N, device="cuda").float() + 1
    o, _ = ops.bag_wgrad(dH2, dact, x2, None, n_rows)
    print("l", l, "->", float(o[3, 9]), "expect", l + 1)
