"""Determinism soak of the batched accumulation window (mhimx_window_run): the SAME window (same state, same seeds, same device tick) run
REPS times eagerly - logits, row lists, the summed gradient and the chained queries must be bit-identical every time (gates, arrival
counters and the bag planes' relocated pointers leave no room for an order-dependent sum).   python tools/exp_window_soak.py [reps=300]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from mhim_mil_amd.engine import FusedTrainer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = 8
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
g = torch.Generator(device=dev); g.manual_seed(2000)
bags = [torch.randn(B.N_INST, B.D_IN, device=dev, generator=g).abs_() for _ in range(k)]
labels = [torch.tensor([i % 2], device=dev) for i in range(k)]
student, teacher, _ = B.make_models(dev, "auto")
tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997, accumulation_steps=k)
snap = (tr.flat.student.clone(), tr.flat.teacher.clone(), tr.tick.clone(), tr.opt_step.clone(), student._step, teacher._step)


def digest():
    h = hashlib.sha256()
    per = tr.last["bags"]
    for b in per:
        h.update(b["logits"].cpu().numpy().tobytes()); h.update(b["rows"].cpu().numpy().tobytes())
    h.update(tr.flat.grad.cpu().numpy().tobytes())
    h.update(student.merge.global_q_mm.detach().cpu().numpy().tobytes())
    return h.hexdigest()


seen = {}
for r in range(reps):
    tr.flat.student.copy_(snap[0]); tr.flat.teacher.copy_(snap[1]); tr.tick.copy_(snap[2]); tr.opt_step.copy_(snap[3])
    student._step, teacher._step = snap[4], snap[5]
    tr.flat.grad.zero_(); tr._micro = 0
    tr.window_step(bags, labels, update=False)
    torch.cuda.synchronize()
    assert tr.last.get("ws") is not None, "the window did not take the batched form"
    d = digest()
    seen[d] = seen.get(d, 0) + 1
print(f"{reps} runs of one window: {len(seen)} distinct result(s): {sorted(seen.values(), reverse=True)}")
assert torch.isfinite(tr.flat.grad).all()
