"""Times the streamed Nystrom attention entry points at T = 50 176 (back-to-back launches, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
torch.manual_seed(0)
qkv = torch.randn(T, 1536, device="cuda")
lm = qkv[:, :1024].reshape(256, T // 256, 1024).mean(1).contiguous()
o = ops.NysOperands(qkv, lm, 0.125)
w2 = torch.randn(8, 256, 64, device="cuda")
dout = torch.randn(T, 512, device="cuda")
dqkv, dlm = torch.empty_like(qkv), torch.empty_like(lm)
a3v, lse3 = ops.nys_a3v_fwd(o)
out, lse1 = ops.nys_out_fwd(o, w2)
da = torch.randn(8, 256, 64, device="cuda")
u = torch.randn(8, 256, device="cuda")
def t(name, fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:12s} {1e3 * e0.elapsed_time(e1) / n:8.1f} us")
t("a3v_fwd", lambda: ops.nys_a3v_fwd(o))
t("out_fwd", lambda: ops.nys_out_fwd(o, w2, out=out))
t("out_bwd", lambda: ops.nys_out_bwd(o, w2, dout, lse1, dqkv, dlm))
t("a3v_bwd", lambda: ops.nys_a3v_bwd(o, a3v, da, lse3, dqkv, dlm, False))
t("cls_attn", lambda: ops.nys_cls_attn(o, lse3, u))
