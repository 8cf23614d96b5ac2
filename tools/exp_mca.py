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
lib.mhimx_mf_prof_read.argtypes = [C.c_void_p]
wkv = m.merge.attn.to_kv.weight.data
frag = torch.empty_like(wkv)
ops.prep_batch([(ops.PREP_FRAG, wkv, frag)])
mw = m._merge_w(None, wkv_frag=frag)
for it in range(4):
    z, _, mws = ops.merge_fwd(mw, H, update_q=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mhimx_mf_prof_read(C.cast(buf, C.c_void_p))
    t = list(buf)
    names = ["load", "gemm", "kv out", "dots", "softmax", "o sums"]
    print(" ".join(f"{n}={(t[i+1]-t[i])/100:.2f}us" for i, n in enumerate(names)), f"total={(t[6]-t[0])/100:.2f}us")
