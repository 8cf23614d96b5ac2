"""The drop-in path of INTEGRATION.md section 2 under the reference trainer's loop (engines/base_engine.py:46-167 with
CommonMIL.forward_func, common_mil.py:14-48): MHIM.forward_teacher + MHIM.forward + criterion + loss.backward() + torch.optim.Adam.step +
the per-parameter EMA update - eager launches, PyTorch autograd around the kernel-backed Functions - timed at BASELINE's c2 size beside
the native FusedTrainer step (which bench.py times).  VERDICT r2 weak item 10."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhim_mil_amd import synth
from mhim_mil_amd.mhim import MHIM
from mhim_mil_amd.engine import CommonMIL, FusedTrainer

N, D = int(os.environ.get("N", 10000)), 1024
BASE = os.environ.get("BASELINE_MODEL", "attn")
STEPS = int(os.environ.get("STEPS", 50))
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
           merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
dev = torch.device("cuda", 0)
base = synth.mhim_state(7, input_dim=D, merge_k=5, baseline=BASE)


def mk():
    m = MHIM(input_dim=D, n_classes=2, baseline=BASE, **CFG)
    sd = dict(base); sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.to(dev).train()


g = torch.Generator(device=dev); g.manual_seed(5)
bags = [torch.randn(1, N, D, device=dev, generator=g).abs_() for _ in range(4)]
label = torch.tensor([1], device=dev)
args = types.SimpleNamespace(model="mhim", baseline=BASE, aux_alpha=0.5, main_alpha=1.0)

# ---- the reference trainer's loop on the drop-in classes
model, ema = mk(), mk()
for p in ema.parameters():
    p.requires_grad_(False)
engine = CommonMIL(args)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=1e-5)
crit = torch.nn.CrossEntropyLoss()
mm = 0.9999


def ref_step(i):
    bag = bags[i % 4]
    logits, lab, aux_loss, patch_num, keep_num, _, _ = engine.forward_func(args, model, ema, bag, label, crit, 1, i, 0, i, None)
    loss = args.main_alpha * crit(logits.view(1, -1), lab) + args.aux_alpha * aux_loss
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    with torch.no_grad():                                    # the reference's EMA: one update per parameter (utils.py EMA.update)
        for pe, ps in zip(ema.parameters(), model.parameters()):
            pe.mul_(mm).add_(ps.detach(), alpha=1.0 - mm)
    return loss


def timed(fn, name):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        fn(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f"{name:46s} {dt * 1e3:8.3f} ms/step  {N / dt / 1e6:6.2f} M inst/s", flush=True)
    return dt


timed(ref_step, f"drop-in classes under the reference loop ({BASE})")

# ---- the same loop with the fused optimiser (optim.FusedAdamEMA: Adam + every parameter's EMA in ONE launch; the loop's own EMA finds
# no teacher parameter left to update)
from mhim_mil_amd.optim import FusedAdamEMA
model, ema = mk(), mk()
opt = FusedAdamEMA(model, ema, lr=2e-4, weight_decay=1e-5, mm=mm)
timed(ref_step, f"the same loop with optim.FusedAdamEMA ({BASE})")
model, ema = mk(), mk()
opt = FusedAdamEMA(model, ema, lr=2e-4, weight_decay=1e-5, mm=mm)
engine = CommonMIL(args, fused=opt)                       # forward_func runs the native forward + backward; the loop is unchanged
timed(ref_step, f"... + CommonMIL(args, fused=optimizer) ({BASE})")
model, ema = mk(), mk()
opt = FusedAdamEMA(model, ema, lr=2e-4, weight_decay=1e-5, mm=mm)
engine = CommonMIL(args, fused=opt, graph_cache=4)        # ... and replays a captured graph for a bag shape it has seen twice
timed(ref_step, f"... + graph_cache=4 ({BASE})")
tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
timed(lambda i: tr.train_step(bags[i % 4][0], label), "FusedTrainer.train_step, eager")
if os.environ.get("PROFILE_EAGER") == "1":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(20):
        tr.train_step(bags[i % 4][0], label)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
graphs = [tr.capture(bags[i][0], label, warmup=2) for i in range(4)]
timed(lambda i: graphs[i % 4].replay(), "FusedTrainer, hipGraph replay")
