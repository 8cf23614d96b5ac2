"""Phase timing of scorer_fused_kernel (experiments only; build with MHIMX_EXTRA_FLAGS=-DMHIMX_SF_PROF)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from mhim_mil_amd import _lib as L, ops

dev = "cuda"
torch.manual_seed(0)
M, E, A = 10000, 512, 128
H = torch.randn(M, E, device=dev).abs()
wa = torch.randn(A, E, device=dev) * 0.05
frag = torch.empty_like(wa)
ops.prep_batch([(ops.PREP_FRAG, wa, frag)])
sc = ops.ScorerW(wa, torch.randn(1, A, device=dev) * 0.1, L.ACT["tanh"], ba=torch.zeros(A, device=dev),
                 bc=torch.zeros(1, device=dev), prec="bf16x3", wa_frag=frag)
wp = torch.randn(2, E, device=dev) * 0.05
lib = L.lib()
lib.mhimx_sf_prof_read.argtypes = [C.c_void_p]
for it in range(4):
    st = ops.abmil_pool_fwd(sc, H, wp=wp)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mhimx_sf_prof_read(C.cast(buf, C.c_void_p))
    t = list(buf)
    names = ["load", "gemm", "epilogue", "pool", "cproj"]
    print(" ".join(f"{n}={(t[i+1]-t[i])/100:.2f}us" for i, n in enumerate(names)), f"total={(t[5]-t[0])/100:.2f}us")
