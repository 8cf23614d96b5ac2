"""Phase timing of select_small_kernel (experiments only): build with
   MHIMX_LIB_NAME=libmhimx_selprof.so MHIMX_EXTRA_FLAGS=-DMHIMX_SEL_PROF python -m mhimx_mil_amd.build, run with MHIMX_LIB_NAME set."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from mhim_mil_amd import _lib as L, ops

dev = "cuda"
torch.manual_seed(0)
N = 10000
logit = torch.randn(N, device=dev) * 0.3
score = torch.softmax(logit, 0)
tick = torch.zeros(1, dtype=torch.int64, device=dev)
k, n_sel = 300, 150
R = (N - n_sel) - int((N - n_sel) * 0.9)
lib = L.lib()
lib.mhimx_sel_prof_read.argtypes = [C.c_void_p]
for it in range(5):
    rows = ops.select_rows(score, k, n_sel, R, 1234, tick=tick, merge_first=True)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.mhimx_sel_prof_read(C.cast(buf, C.c_void_p))
    t = list(buf)
    order = [8, 0, 1, 2, 3, 4, 5, 6, 10, 7]
    names = ["load", "radix1", "gather", "sort", "flags", "compact", "radix2", "scan", "stores"]
    print(" ".join(f"{n}={(t[b]-t[a])/100:.2f}us" for n, a, b in zip(names, order[:-1], order[1:])), f"total={(t[7]-t[8])/100:.2f}us")
