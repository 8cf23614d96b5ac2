"""What the step's preparation launch (mhimx_prep_batch, ~13 us at c2) spends its time on: the launch with subsets of its jobs, each
captured 20x into a hipGraph and replayed (experiments only)."""
import sys
import torch
sys.path.insert(0, ".")
import bench as B
from mhim_mil_amd import ops
from mhim_mil_amd.engine import FusedTrainer

dev = torch.device("cuda", 0)
torch.manual_seed(0)
student, teacher, _ = B.make_models(dev, "auto")
tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997)
x = torch.randn(B.N_INST, B.D_IN, device=dev).abs_()
lab = torch.tensor([1], device=dev)
tr.train_step(x, lab)
torch.cuda.synchronize()

captured = {}
orig = ops.prep_batch


def spy(jobs):
    captured["jobs"] = list(jobs)
    orig(jobs)


ops.prep_batch = spy
import mhim_mil_amd.engine as E
prep_t, preps = tr._nat_prep([x], None, with_opt_tick=True)
ops.prep_batch = orig
jobs = captured["jobs"]
names = {0: "transpose", 1: "pair", 2: "copy", 3: "tick", 4: "frag", 5: "frag_t", 6: "merge", 7: "pair_t"}
for k, src, dst in jobs:
    print(names[k], None if src is None or not hasattr(src, "shape") else tuple(src.shape))


def time_jobs(js, reps=20, loops=30):
    if not js:
        return 0.0
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.prep_batch(js)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                ops.prep_batch(js)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(loops):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * loops)


print("all jobs           %.2f us" % time_jobs(jobs))
for kind in sorted({k for k, _, _ in jobs}):
    print("only %-10s    %.2f us" % (names[kind], time_jobs([j for j in jobs if j[0] == kind])))
    print("all but %-10s %.2f us" % (names[kind], time_jobs([j for j in jobs if j[0] != kind])))
pairs = [j for j in jobs if j[0] == 1]
print("first pair only    %.2f us" % time_jobs(pairs[:1]))
print("ticks + pairs      %.2f us" % time_jobs([j for j in jobs if j[0] in (1, 3)]))
