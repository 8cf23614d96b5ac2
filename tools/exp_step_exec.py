"""The step behind the C-ABI (mhimx_step_run) against the Python orchestration on bags whose size NEVER repeats (nothing can be captured):
ms per step (device-complete), host enqueue time alone, and the drop-in loop (CommonMIL(fused=) under the reference trainer's loop body)
over a 200-bag epoch of distinct sizes, first epoch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types
import torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import FusedTrainer, CommonMIL
from mhim_mil_amd.optim import FusedAdamEMA

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


NB = int(os.environ.get("NB", 200))
g = torch.Generator(device=dev); g.manual_seed(5)
sizes = [9000 + 10 * j for j in range(NB)]                       # 200 distinct sizes around the c2 bag, mean 9 995 rows
if os.environ.get("WIDE"):
    # round 6: whole-slide bags - 200 distinct sizes between 9 000 and 60 000 rows, shuffled; above 16 384 rows the executor
    # issues the multi-workgroup select.  Eager (nothing captured) against the hipGraph replay of the same step on a sample of the sizes.
    sizes = [9000 + 255 * j for j in range(NB)]
    import random
    random.Random(3).shuffle(sizes)
    tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
    label = torch.tensor([1], device=dev)
    x0 = torch.randn(max(sizes), D, device=dev, generator=g).abs_()
    for n in sizes[:4]:
        tr.train_step(x0[:n], label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in sizes:
        tr.train_step(x0[:n], label)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"C executor, {NB} bags of distinct sizes 9 000 .. {max(sizes)} (mean {sum(sizes) / NB:.0f}), eager: {(t2 - t0) / NB * 1e3:.3f} ms/step "
          f"({sum(sizes) / (t2 - t0) / 1e6:.1f} M inst/s), host enqueue alone {(t1 - t0) / NB * 1e3:.3f} ms/step")
    sample = sizes[::10]
    te = tg = 0.0
    for n in sample:
        xb = x0[:n]
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        for _ in range(2):
            tr.train_step(xb, label)
        e0.record()
        for _ in range(5):
            tr.train_step(xb, label)
        e1.record()
        gr = tr.capture(xb, label, warmup=1)
        gr.replay(); gr.replay()
        e2.record()
        for _ in range(5):
            gr.replay()
        e3.record()
        torch.cuda.synchronize()
        te += e0.elapsed_time(e1) / 5
        tg += e2.elapsed_time(e3) / 5
        del gr
    print(f"sample of {len(sample)} of those sizes, 5 steps each: eager executor {te / len(sample):.3f} ms/step, hipGraph replay {tg / len(sample):.3f} ms/step "
          f"-> eager / replay = {te / tg:.3f}")
    sys.exit(0)
bags = [torch.randn(n, D, device=dev, generator=g).abs_() for n in sizes]
label = torch.tensor([1], device=dev)
inst = sum(sizes)

for name, use in (("C executor (mhimx_step_run)", True), ("Python orchestration", False)):
    tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
    tr.use_executor = use
    for b in bags[:5]:
        tr.train_step(b, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bags:
        tr.train_step(b, label)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {NB} bags of distinct sizes, eager: {(t2 - t0) / NB * 1e3:.3f} ms/step ({inst / (t2 - t0) / 1e6:.1f} M inst/s), "
          f"host enqueue alone {(t1 - t0) / NB * 1e3:.3f} ms/step")

# one call per chunk of bags
tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
tr.run_steps(bags[:8], [label] * 8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in range(0, NB, 8):
    tr.run_steps(bags[c:c + 8], [label] * len(bags[c:c + 8]))
torch.cuda.synchronize()
print(f"run_steps (8 bags per C call): {(time.perf_counter() - t0) / NB * 1e3:.3f} ms/step")

# the drop-in loop: base_engine.py:76-167's body around CommonMIL(fused=optimizer), first epoch over distinct sizes (nothing to replay)
args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5, main_alpha=1.0)
for gc in (0, 4):
    model, model_ema = mk(), mk()
    opt = FusedAdamEMA(model, model_ema, lr=2e-4, mm=0.9997, aux_alpha=0.5)
    eng = CommonMIL(args, fused=opt, graph_cache=gc)
    crit = torch.nn.CrossEntropyLoss()
    def epoch(bs):
        for it, b in enumerate(bs):
            logits, lab, aux, pn, kn, _, _ = eng.forward_func(args, model, model_ema, b.unsqueeze(0), label, crit, 1, it, 0, it, None)
            loss = args.main_alpha * crit(logits.view(1, -1), lab) + args.aux_alpha * aux
            loss.backward()
            opt.step()
            opt.zero_grad()
    epoch(bags[:5])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    epoch(bags)
    torch.cuda.synchronize()
    print(f"reference loop body, CommonMIL(fused=, graph_cache={gc}), first epoch over {NB} distinct sizes: {(time.perf_counter() - t0) / NB * 1e3:.3f} ms/step")
    opt.close()
