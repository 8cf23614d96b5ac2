"""Phase timing of mca_fwd_part_kernel (experiments only; build with MHIMX_EXTRA_FLAGS=-DMHIMX_MCA_PROF)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mhim_mil_amd import _lib as L, ops, synth
from mhim_mil_amd.mhim import MHIM

dev = "cuda"
m = MHIM(input_dim=1024, n_classes=2, merge_enable=True, merge_k=5).to(dev).train()
H = torch.randn(985, 512, device=dev).abs()
mw = m._merge_w(None)
lib = L.lib()
lib.mhimx_mca_prof_read.argtypes = [C.c_void_p]
for it in range(4):
    z, _, mws = ops.merge_fwd(mw, H, update_q=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mhimx_mca_prof_read(C.cast(buf, C.c_void_p))
    t = list(buf)
    names = ["issue loads", "loads land", "row loop", "partials out"]
    print(" ".join(f"{n}={(t[i+1]-t[i])/100:.2f}us" for i, n in enumerate(names)), f"total={(t[4]-t[0])/100:.2f}us")
