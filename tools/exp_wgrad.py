"""Timing of the projection's weight gradient at c2 size: the two kernels of the dedicated pair (wgrad.hip) and the generic pair."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mhim_mil_amd import _lib as L
from mhim_mil_amd import ops

N, L_, E, D = 10000, 9705, 512, 1024
torch.manual_seed(0)
x = torch.randn(N, D, device="cuda").abs()
dH = torch.randn(N + 6, E, device="cuda") * 1e-3
dact = torch.randn(N + 6, E, device="cuda").half()
rows = torch.randperm(N, device="cuda")[:L_].sort().values
ow, ob = torch.empty(E, D, device="cuda"), torch.empty(E, device="cuda")
lib = L.lib()
img = torch.zeros(lib.mhimx_wgrad_image_bytes(L_, E) // 4 + 1024, device="cuda")      # (+ room for the -DWG_PROF stamps behind the image)
ws_b = torch.empty(2 * -(-L_ // 32) * E, device="cuda")
ws = torch.empty(lib.mhimx_wgrad_ws_floats(L_, E, D), device="cuda")
g = L.BagWgrad(img=img.data_ptr(), X=x.data_ptr(), ldx=D, n_bag_rows=N, rows=rows.data_ptr(), L=L_, E=E, D=D, C=ow.data_ptr(), ldc=D,
               accumulate=0, ws=ws.data_ptr(), ws_floats=ws.numel(), defer=None)
lst = ops.ReduceList()                       # queue the final reductions (never flushed: the kernels alone are timed)


def image():
    lst.c.n = 0
    L.check(lib.mhimx_rows_dpre_image(None, dH.data_ptr(), dact.data_ptr(), rows.data_ptr(), L_, E, img.data_ptr(), ob.data_ptr(), 0,
                                      ws_b.data_ptr(), ws_b.numel() * 4, lst.ptr()), "image")


def wgrad():
    lst.c.n = 0
    g.defer = lst.ptr()
    L.check(lib.mhimx_bag_wgrad(None, C.byref(g)), "wgrad")


def old():
    dpre, _ = ops.rows_dpre(dH, dact, rows, L_, colsum_out=ob)
    ops.gemm_tn(dpre, x, out=ow, rows=rows, splits=8, prec="bf16x3", M=L_)


keep = torch.zeros(N, dtype=torch.uint8, device="cuda"); keep[rows] = 1
ximg = torch.empty(ops.bag_ximage_floats(x), device="cuda")
img2 = torch.zeros(lib.mhimx_wgrad_image_bytes(N, E) // 4, device="cuda")
ws2 = torch.empty(lib.mhimx_wgrad_ws_floats(N, E, D), device="cuda")
g2 = L.BagWgrad(img=img2.data_ptr(), X=x.data_ptr(), ldx=D, n_bag_rows=N, rows=None, L=N, E=E, D=D, C=ow.data_ptr(), ldc=D,
                accumulate=0, ws=ws2.data_ptr(), ws_floats=ws2.numel(), defer=None, ximg=ximg.data_ptr())


def ximage():
    ops.prep_batch([(ops.PREP_XIMG, x, ximg)])


def image_k():
    lst.c.n = 0
    L.check(lib.mhimx_rows_dpre_image_k(None, dH.data_ptr(), dact.data_ptr(), keep.data_ptr(), N, E, img2.data_ptr(), ob.data_ptr(), 0,
                                        ws_b.data_ptr(), ws_b.numel() * 4, lst.ptr()), "image_k")


def wgrad_dma():
    lst.c.n = 0
    g2.defer = lst.ptr()
    L.check(lib.mhimx_bag_wgrad(None, C.byref(g2)), "wgrad_dma")


which = sys.argv[1:] or ["image", "wgrad", "ximage", "image_k", "wgrad_dma", "image", "wgrad", "wgrad_dma"]
for name in which:
    fn = {"image": image, "wgrad": wgrad, "old": old, "ximage": ximage, "image_k": image_k, "wgrad_dma": wgrad_dma}[name]
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-6s %.1f us per call (back-to-back launches)" % (name, e0.elapsed_time(e1) * 1e3 / 100))

if os.environ.get("WG_PROF"):
    wgrad(); torch.cuda.synchronize()
    tail_ = img[lib.mhimx_wgrad_image_bytes(L_, E) // 4:][:64].cpu()
    st = tail_[:32].view(8, 4).tolist()
    if float(tail_[32]) > 0:
        print("workgroup 0: %.0f shader cycles in %.2f us (100 MHz clock) -> %.3f GHz" % (float(tail_[33]), float(tail_[32]) / 100, float(tail_[33]) / float(tail_[32]) / 10))
    for w in (0, 4):
        print("wave %d: entry -> loop %.0f, loop %.0f (%d k-steps), epilogue %.0f shader cycles" % (w, st[w][0], st[w][1], st[w][3], st[w][2]))
