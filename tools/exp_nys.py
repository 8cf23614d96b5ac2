"""Time the streamed Nystrom attention entry points on their own at the c3 size (T = 50 176 tokens): forward pair and the two backward
entry points (each = one token-column kernel + one landmark-column kernel + small reductions).  Environment switches select kernel
forms (MHIMX_NYS_BWD_T_V1, MHIMX_NYS_BWD_Q_V1, MHIMX_NYS_STAGGER ...): run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops

T = int(os.environ.get("T", 50176)); REP = int(os.environ.get("REP", 20))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
qkv = torch.randn(T, 1536, device=dev, generator=g) * 0.5
lm = torch.randn(256, 1024, device=dev, generator=g) * 0.5
no = ops.NysOperands(qkv, lm, 0.125)
a3v, lse3 = ops.nys_a3v_fwd(no)
w2 = torch.randn(8, 256, 64, device=dev, generator=g) * 0.1
out, lse1 = ops.nys_out_fwd(no, w2)
dout = torch.randn(T, 512, device=dev, generator=g) * 0.1
dqkv = torch.zeros_like(qkv); dlm = torch.zeros_like(lm)
da3v = torch.randn(8, 256, 64, device=dev, generator=g) * 0.1


def timed(name, fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / REP * 1e3:8.1f} us", flush=True)


timed("a3v_fwd", lambda: ops.nys_a3v_fwd(no))
timed("out_fwd", lambda: ops.nys_out_fwd(no, w2, out))
timed("out_fwd+=", lambda: ops.nys_out_fwd(no, w2, out, accumulate=True))
timed("out_bwd", lambda: ops.nys_out_bwd(no, w2, dout, lse1, dqkv, dlm))
timed("a3v_bwd", lambda: ops.nys_a3v_bwd(no, a3v, da3v, lse3, dqkv, dlm, True))
print("checks", float(dqkv.abs().sum()), float(dlm.abs().sum()))
u = torch.randn(8, 256, device=dev, generator=g)
timed("cls_attn", lambda: ops.nys_cls_attn(no, lse3, u))
