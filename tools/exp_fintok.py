"""Phase stamps of pool_finalize_tok_kernel (block 0) inside a c2 train step ("find the 7.5 us"): the stamped
profile build (mhim_mil_amd/libmhimx_prof.so: python -m mhim_mil_amd.build --prof), run with MHIMX_LIB_NAME=libmhimx_prof.so."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import _lib as L, synth
from mhim_mil_amd.engine import FusedTrainer
from mhim_mil_amd.mhim import MHIM

D = 1024
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
dev = torch.device("cuda", 0)
base = synth.mhim_state(7, input_dim=D, merge_k=5)


def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline="attn", **CFG)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.to(dev).train()


tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
g = torch.Generator(device=dev); g.manual_seed(5)
bags = [torch.randn(10000, D, device=dev, generator=g).abs_() for _ in range(8)]
label = torch.tensor([1], device=dev)
lib = L.lib()
lib.mhimx_ft_prof_read.argtypes = [C.c_void_p]
names = ["entry -> token rows in LDS + the 64 weight loads landed", "token products + chunk sums (-> us)", "u_pre / token scores", "statistics",
         "weighted sum of the pooled rows + z"]
acc = [0.0] * 5
n = 0
for it in range(24):
    tr.train_step(bags[it % 8], label)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mhimx_ft_prof_read(C.cast(buf, C.c_void_p))
    t = list(buf)
    if it >= 4:
        for j in range(5):
            acc[j] += (t[j + 1] - t[j]) / 100.0
        n += 1
print("pool_finalize_tok_kernel, block 0, c2 step (eager executor), mean of %d launches (us, 100 MHz clock):" % n)
for nm, a in zip(names, acc):
    print("  %-62s %.2f" % (nm, a / n))
print("  %-62s %.2f" % ("first stamp -> last stamp", sum(acc) / n))
