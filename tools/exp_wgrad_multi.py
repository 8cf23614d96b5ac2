"""mhimx_bag_wgrad_multi against the per-bag launches: values and time (8 bags of the c2 shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import ops
dev = "cuda"
N, D, E, NB = 10000, 1024, 512, int(os.environ.get("NB", 8))
g = torch.Generator(device=dev); g.manual_seed(3)
Lr = 9700
bags = [torch.randn(N, D, device=dev, generator=g).abs_() for _ in range(NB)]
dHs = [torch.randn(N, E, device=dev, generator=g) * 0.01 for _ in range(NB)]
dacts = [(torch.rand(N, E, device=dev, generator=g) * 1.3).half() for _ in range(NB)]
rows = [torch.randperm(N, device=dev, generator=g)[:Lr].contiguous() for _ in range(NB)]


def per_bag():
    w, b = torch.empty(E, D, device=dev), torch.empty(E, device=dev)
    for j in range(NB):
        ops.bag_wgrad(dHs[j], dacts[j], bags[j], rows[j], Lr, out_w=w, out_b=b, accumulate=j > 0)
    return w, b


def multi():
    w, b = torch.empty(E, D, device=dev), torch.empty(E, device=dev)
    ims = [ops.bag_wgrad_image(dHs[j], dacts[j], bags[j], rows[j], Lr, out_b=b, accumulate=j > 0) for j in range(NB)]
    ops.bag_wgrad_multi(ims, w)
    return w, b


w0, b0 = per_bag(); w1, b1 = multi(); torch.cuda.synchronize()
ref = sum((dHs[j][rows[j]].double() * dacts[j][rows[j]].double()).t() @ bags[j][rows[j]].double() for j in range(NB))
rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
print("per-bag vs fp64 %.2e   multi vs fp64 %.2e   bias equal: %s" % (rel(w0, ref), rel(w1, ref), torch.equal(b0, b1)))
for name, fn in (("per-bag launches", per_bag), ("one multi launch", multi)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-18s %.1f us per window of %d bags (images + products + slab sums)" % (name, e0.elapsed_time(e1) * 1e3 / 20, NB))
ims = [ops.bag_wgrad_image(dHs[j], dacts[j], bags[j], rows[j], Lr) for j in range(NB)]
w = torch.empty(E, D, device=dev)
for name, fn in (("images only", lambda: [ops.bag_wgrad_image(dHs[j], dacts[j], bags[j], rows[j], Lr) for j in range(NB)]),
                 ("product only (multi)", lambda: ops.bag_wgrad_multi(ims, w))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-22s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / 20))
