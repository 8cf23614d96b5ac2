"""bench.py — MHIM(ABMIL) train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete train step on one synthetic bag per rank (N=10 000 instances, D=1024, config c2):
teacher forward + hard-instance select + student forward + CE/distillation head + backward + [RCCL all-reduce of
the flat gradient buffer] + fused Adam + EMA teacher.  Bags are resident in HBM before the timed region (8 distinct
bags per rank = 328 MB > the 256 MB Infinity Cache, rotated).  Rank 0 prints ONE JSON line.

The default one-GPU run also carries, inside the same JSON line (each with its own short, fixed step count):
  "accumulate8"      the same step with --accumulation_steps 8 as ONE captured window (FusedTrainer.capture_window): ms per bag;
  "hbm_copy"         the on-box float4 stream-copy rate (GB/s read + written) beside the nominal 8 TB/s;
  "other_workloads"  c3 (MHIM(TransMIL) N=50k) and c5 (one bag N=200k D=1536, the sharded code path at world 1), each with its
                     roofline figure and a one-step cpu_baseline.   --no-extras skips all three.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_INST, D_IN, N_BAGS = 10000, 1024, 8
CFG = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
           merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
ALGO_BYTES_PER_INST_STEP = 3 * D_IN * 4 + 4 * (1 + 2) + 8          # 12 308 B at D=1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--prec", default="auto", help="auto|bf16x3|f16s|f32 (matrix-core form of the feature GEMM)")
    ap.add_argument("--cpu-steps", type=int, default=400, help="oracle steps for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--dp-graph", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--dp-eager", action="store_true",
                    help="N > 1: eager steps whose all-reduce overlaps the backward instead of graph(fwd+bwd) | all-reduce | graph(Adam+EMA)")
    ap.add_argument("--no-repeats", action="store_true", help="skip the five repeats of the timed region (region_repeats)")
    ap.add_argument("--no-extras", action="store_true", help="c2, one GPU: skip the accumulate8 / hbm_copy / other_workloads legs")
    ap.add_argument("--window-streams", type=int, default=4, help="HIP streams of the accumulate-8 window")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="c2, one GPU: consecutive complete train steps (one bag, one update each) captured per hipGraph (FusedTrainer.capture_steps); "
                         "0 = the largest divisor of gcd(steps, warmup) that is <= 8, 1 = one graph per bag")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "c2-dsmil", "c3-sharded"],
                    help="c2 (default, BASELINE.json's metric): MHIM(ABMIL) N=10k D=1024, one bag per GPU per step; "
                         "c3: MHIM(TransMIL) N=50k D=1024 (replicas); c5: ONE bag N=200k D=1536 instance-sharded over the GPUs; "
                         "c2-dsmil: MHIM(DSMIL) N=10k D=1024 (scope row N1)")
    return ap.parse_args()


def make_models(dev, prec):
    from mhim_mil_amd import synth
    from mhim_mil_amd.mhim import MHIM
    base = synth.mhim_state(7, input_dim=D_IN, merge_k=5)

    def mk(sd):
        m = MHIM(input_dim=D_IN, n_classes=2, baseline="attn", prec=prec, **CFG)
        sd = dict(sd)
        sd["merge.global_q"] = sd["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        return m.to(dev).train()

    return mk(base), mk(base), base           # teacher = deepcopy(student) at init (modules/__init__.py:176-214)


def cpu_model():
    """The host CPU's model string ("print CPU model and core count")."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(steps, base):
    """The oracle (a port of the reference's math; the reference itself cannot travel) on the host cores."""
    import numpy as np
    from mhim_mil_amd import synth
    from oracle import mhim_oracle as O
    cfg = O.Cfg(**CFG)                       # dropout 0.25 as BASELINE.md section 3 has it: the masks are drawn by torch's CPU dropout
    O.DRAW_DROPOUT = True
    stu, tea, opt = O.as_torch(base), O.as_torch(base), {}
    x = torch.from_numpy(synth.bag(4242, N_INST, D_IN))
    k, n_sel, _ = O.mask_count(N_INST, CFG["mask_ratio_h"], CFG["mask_ratio_hr"])
    perm, shuf = synth.permutation(1, k), synth.permutation(2, N_INST - n_sel)
    # torch's CPU ops scale poorly past a few dozen threads on ops this small (one oracle step at all 256 hardware
    # threads of the GPU box takes ~28 s): take the best of a short thread sweep, then time the sample with it.
    ncpu = os.cpu_count() or 1
    best = (None, 1e30)
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        O.train_step(x, 1, stu, tea, opt, cfg, 1, perm=perm, ids_shuffle=shuf)        # warm-up at this width
        t0 = time.perf_counter()
        O.train_step(x, 1, stu, tea, opt, cfg, 1, perm=perm, ids_shuffle=shuf)
        dt1 = time.perf_counter() - t0
        if dt1 < best[1]:
            best = (th, dt1)
        if dt1 > 5.0:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    done = 0
    for s in range(steps):
        stu, tea, opt, _ = O.train_step(x, s % 2, stu, tea, opt, cfg, s + 1, perm=perm, ids_shuffle=shuf)
        done += 1
        if time.perf_counter() - t0 > 15.0:
            break
    dt = time.perf_counter() - t0
    return {"value": N_INST * done / dt, "unit": "patch-instances/s", "cores": cores, "cpu_model": cpu_model(), "host_threads": ncpu, "kind": "port",
            "sample": f"{done} oracle train steps (torch CPU fp32, {cores} threads = best of an 8/16/32/64 sweep on a "
                      f"{ncpu}-thread host, dropout 0.25 drawn by torch) on one N={N_INST} D={D_IN} bag, {dt:.1f} s"}


def cpu_baseline_other(workload, base, n, d, bl):
    """c3 / c5 / c2-dsmil: ONE oracle train step of the same configuration on the host cores (a bounded sample: 10-40 s)."""
    from mhim_mil_amd import synth
    from oracle import mhim_oracle as O
    cfg = O.Cfg(**{**CFG, "baseline": bl})
    O.DRAW_DROPOUT = True
    stu, tea = O.as_torch(base), O.as_torch(base)
    x = torch.from_numpy(synth.bag(4242, n, d))
    k, n_sel, _ = O.mask_count(n, CFG["mask_ratio_h"], CFG["mask_ratio_hr"])
    perm, shuf = synth.permutation(1, k), synth.permutation(2, n - n_sel)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    opt = {}
    t0 = time.perf_counter()
    stu, tea, opt, _ = O.train_step(x, 1, stu, tea, opt, cfg, 1, perm=perm, ids_shuffle=shuf)     # warm-up (untimed)
    warm = time.perf_counter() - t0
    # a bounded sample: three timed steps unless one step already takes more than ~8 s (then as many as fit ~20 s, at least one)
    n_timed = 3 if warm < 8.0 else max(1, int(20.0 // warm))
    t0 = time.perf_counter()
    for q in range(n_timed):
        stu, tea, opt, _ = O.train_step(x, q % 2, stu, tea, opt, cfg, q + 2, perm=perm, ids_shuffle=shuf)
    dt = (time.perf_counter() - t0) / n_timed
    return {"value": n / dt, "unit": "patch-instances/s", "cores": cores, "cpu_model": cpu_model(), "host_threads": os.cpu_count() or 1, "kind": "port",
            "sample": f"{n_timed} oracle train steps after one warm-up step (torch CPU fp32, {cores} threads, dropout 0.25 drawn by torch) on one "
                      f"N={n} D={d} bag, {dt:.1f} s per step"}


def sustained_clock():
    """The shader clock the projection launch holds, measured INSIDE the kernel (s_memtime shader cycles against the constant 100 MHz
    s_memrealtime over one workgroup's life; 100 back-to-back launches of the c2 shape, stamps of the last): the stamped profile build of the
    same sources (mhim_mil_amd/libmhimx_prof.so, __graft_entry__.build()) in a process of its own.  None when that library is absent."""
    import re
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "mhim_mil_amd", "libmhimx_prof.so")):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_proj_prof.py")], capture_output=True, text=True, timeout=180,
                           env={**os.environ, "MHIMX_LIB_NAME": "libmhimx_prof.so", "REPS": "100"})
        m = re.search(r"wave 0: (\d+) shader cycles in ([0-9.]+) us -> ([0-9.]+) GHz", r.stdout)
        if not m:
            return None
        return {"GHz": float(m.group(3)), "shader_cycles": int(m.group(1)), "us": float(m.group(2)),
                "source": "tools/exp_proj_prof.py on libmhimx_prof.so (-DPW_PROF=2): workgroup 0 of the 100th back-to-back projection launch, c2 shape"}
    except Exception:      # noqa: BLE001 - a calibration extra: never fails the bench
        return None


def timed(a, world, dev, step):
    """W warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks."""
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def other_workload(a, world, rank, dev, embedded=False):
    """c3 / c5: parity-test configurations of BASELINE.json (not the headline metric).  Returns rank 0's result dict (None elsewhere);
    ``embedded``: called from the default c2 run with its own short step counts."""
    from mhim_mil_amd import synth
    from mhim_mil_amd.mhim import MHIM
    c3 = a.workload in ("c3", "c2-dsmil")                 # replicas driven by FusedTrainer's autograd path
    bl = {"c3": "selfattn", "c5": "attn", "c2-dsmil": "dsmil", "c3-sharded": "selfattn"}[a.workload]
    n_total, d = {"c3": (50000, 1024), "c5": (200000, 1536), "c2-dsmil": (N_INST, D_IN), "c3-sharded": (50000, 1024)}[a.workload]
    base = synth.mhim_state(7, input_dim=d, merge_k=5, baseline=bl)

    def mk():
        m = MHIM(input_dim=d, n_classes=2, baseline=bl, prec=a.prec, **CFG)
        sd = dict(base)
        sd["merge.global_q"] = sd["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        return m.to(dev).train()

    student, teacher = mk(), mk()
    g = torch.Generator(device=dev)
    lab = torch.tensor([1], device=dev)
    if c3:
        from mhim_mil_amd.engine import FusedTrainer
        g.manual_seed(3000 + rank)
        tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997)
        bags = [torch.randn(n_total, d, device=dev, generator=g).abs_() for _ in range(2)]
        graphs = None if a.no_graph else [tr.capture(b, lab, warmup=2) for b in bags]
        step = (lambda i: tr.train_step(bags[i % 2], lab)) if graphs is None else (lambda i: graphs[i % 2].replay())
        per_step, scaling, par = n_total * world, "weak", f"dp{world} (replicas, one bag per GPU per step)"
        name = (f"c3: MHIM(TransMIL/Nystrom)" if bl == "selfattn" else "MHIM(DSMIL)") + f" train step, one bag N={n_total} D={d} per GPU per step"
    else:
        from mhim_mil_amd.sharded import ShardedBagTrainer
        assert n_total % world == 0
        n = n_total // world
        g.manual_seed(5000 + rank)
        tr = ShardedBagTrainer(student, teacher, counts=[n] * world, seed=11, aux_alpha=0.5, mm=0.9997)
        bags = [torch.randn(n, d, device=dev, generator=g).abs_() for _ in range(2)]
        step = lambda i: tr.train_step(bags[i % 2], lab)
        c5_launch = "eager"
        if bl == "selfattn":
            c5_launch = "eager (the sequence-parallel TransMIL step plans its exchanges on the host every step: sharded_transmil.py)"
        elif not a.no_graph:
            # graph | exchange | graph ...: one set of segments per resident shard buffer
            try:
                replays = [tr.capture(b, lab, warmup=2) for b in bags]
                step = lambda i, replays=replays: replays[i % 2]()
                c5_launch = "hipGraph segments between the exchanges (graph | collective | graph ...), one set per resident shard"
            except Exception as e:  # noqa: BLE001 - reported in the JSON line
                c5_launch = f"eager (segment capture failed: {type(e).__name__}: {str(e)[:160]})"
        per_step, scaling, par = n_total, "strong", f"instance-sharded over {world} GPU(s), {n} rows each"
        name = (f"c5: MHIM(ABMIL) train step on ONE bag N={n_total} D={d} sharded by rows" if bl == "attn" else
                f"c3-sharded: MHIM(TransMIL/Nystrom) train step on ONE bag N={n_total} D={d} sharded by rows, sequence-parallel encoder")
    dt = timed(a, world, dev, step)
    if rank == 0:
        algo = 3 * d * 4 + 4 * (1 + 2) + 8                   # three passes over X + score / ids, per instance per step
        extra = {"whole_step_hbm_roofline": {"algorithmic_bytes_per_instance": algo,
                                             "achieved_GBps": per_step * a.steps / dt / world * algo / 1e9,
                                             "frac_of_8TBps": per_step * a.steps / dt / world * algo / 1e9 / HBM_PEAK_GBS,
                                             "note": "step-level figure (no single dominant kernel; the attention matrices are streamed, "
                                                     "never materialised): the TransMIL step is matrix-core work, see `roofline`"
                                             if bl == "selfattn" else "step-level figure"}}
        if bl == "selfattn":
            # ~38 MFLOP per instance per train step (fp32-equivalent: every product runs as 3 bf16 MFMA terms)
            tf = per_step * a.steps / dt / world * 38e6 / 1e12
            extra["roofline"] = {"kernel": "whole step (flat profile: GEMMs 32 %, streamed Nystrom kernels 21 %, pseudo-inverse 13 %)",
                                 "bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                                 "frac_counting_3_bf16_terms": 3 * tf / 2500.0, "traffic": None,
                                 "algorithmic_flops_per_instance": 38e6}
        else:
            hb = extra["whole_step_hbm_roofline"]
            extra["roofline"] = {"kernel": "whole step (projection + weight-gradient GEMMs 63 %, scorer passes, select, Merge)", "bound": "hbm",
                                 "achieved": hb["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb["frac_of_8TBps"], "traffic": None,
                                 "algorithmic_bytes_per_instance": algo}
        # step-level HBM-side bytes from the committed rocprofv3 --pmc passes of this same command, eager (tools/pmc_step.py: FETCH_SIZE x2 +
        # WRITE_SIZE summed over every launch of one step; counters cannot be read from inside the process being timed)
        for pmc_name in (f"r05_pmc_traffic_{a.workload}.json",):
            pmc = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc) and "roofline" in extra:
                pj = json.load(open(pmc))
                extra["roofline"]["traffic"] = pj["step_traffic_bytes"]
                extra["roofline"]["traffic_unit"] = "bytes per step (all launches)"
                extra["roofline"]["traffic_source"] = f"profiles/{pmc_name}: read {pj['step_read_bytes'] / 1e6:.0f} MB + write {pj['step_write_bytes'] / 1e6:.0f} MB per step"
        if world == 1 and a.cpu_steps > 0:
            extra["cpu_baseline"] = cpu_baseline_other(a.workload, base, n_total, d, bl)
        return {
            **extra,
            "metric": f"patch-instances/sec through MHIM fwd+bwd ({a.workload})", "value": per_step * a.steps / dt,
            "unit": "patch-instances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic |N(0,1)| bags resident in HBM, random-init weights (reference init law)",
            "config": {"workload": name, "parallelism": par, "dropout": CFG["dropout"],
                       "launch": ("hipGraph replay" if not a.no_graph else "eager") if c3 else c5_launch}}
    return None


def hbm_copy_rate(dev):
    """On-box stream-copy microbenchmark ('confirm with a copy microbenchmark on the box'): 1 GiB float4 copy through the
    library's own kernel, HIP events, best of 12; bytes = read + written."""
    from mhim_mil_amd import ops
    n = 1 << 28
    src, dst = torch.empty(n, device=dev).normal_(), torch.empty(n, device=dev)
    best = 1e30
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.stream_copy(src, dst)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return {"GBps": 2 * n * 4 / (best * 1e-3) / 1e9, "bytes_moved": 2 * n * 4, "best_ms": best, "nominal_peak_GBps": HBM_PEAK_GBS,
            "kernel": "mhimx_stream_copy (float4 grid-stride copy, 1 GiB -> 1 GiB: far beyond the 256 MB Infinity Cache)"}


def accumulate8(a, dev, bags, labels):
    """--accumulation_steps 8 (base_engine.py:29,100-119; SURVEY 8(d) c4: 'also report accumulate-8'): one optimiser update per 8 bags,
    the window captured as ONE hipGraph (FusedTrainer.capture_window).  Round 6: the window is ONE call of mhimx_window_run - every
    launch between the projections and the weight gradient covers all 8 bags (gridDim.z = bag), ~20 launches per window;
    MHIMX_WINDOW_BATCHED=0: the bags on HIP streams (~127 launches), as rounds 3-5 had it."""
    from mhim_mil_amd.engine import FusedTrainer
    K, reps, warm = 8, 12, 3
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    more = [torch.randn(N_INST, D_IN, device=dev, generator=g).abs_() for _ in range(K)]     # second window: 16 distinct bags = 655 MB
    sets = [(bags[:K], labels[:K]), (more, labels[:K])]
    res = _accumulate8_form(a, dev, sets, K, reps, warm, batched=None)
    was_batched = res.pop("_batched", False)
    if not a.no_graph and was_batched:
        # the same two windows in the stream form of rounds 3-5, on this box, right behind the batched form: the A/B in the line itself
        st = _accumulate8_form(a, dev, sets, K, reps, warm, batched=False)
        st.pop("_batched", None)
        res["stream_form_same_box"] = {k: st[k] for k in ("ms_per_bag", "ms_per_window", "launch")}
    return res


def _accumulate8_form(a, dev, sets, K, reps, warm, batched):
    from mhim_mil_amd.engine import FusedTrainer
    student, teacher, _ = make_models(dev, a.prec)
    tr = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997, accumulation_steps=K)
    if batched is not None:
        tr.window_batched = bool(batched)
    batched = False
    if a.no_graph:
        run = [lambda b=b, l=l: tr.window_step(b, l, n_streams=a.window_streams) for b, l in sets]
        launch = "eager"
    else:
        wins = [tr.capture_window(b, l, warmup=1, n_streams=a.window_streams) for b, l in sets]
        run = [w.replay for w in wins]
        batched = tr._exec_window_ok([b[0] if b.dim() == 3 else b for b in sets[0][0]], sets[0][1])
        launch = ("ONE hipGraph per window: mhimx_window_run, every launch over all 8 bags (one projection launch, the step's middle with one grid plane "
                  "per bag, one weight-gradient launch, one Adam + EMA)" if batched else "ONE hipGraph per window, the HIP streams as its branches")
    for i in range(warm):
        run[i % 2]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        run[i % 2]()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"_batched": batched, "value": N_INST * K / dt, "unit": "patch-instances/s", "ms_per_bag": 1e3 * dt / K, "ms_per_window": 1e3 * dt, "bags_per_update": K,
            "windows_timed": reps, "warmup_windows": warm, "streams": a.window_streams, "launch": launch,
            "whole_step_hbm_roofline_frac_of_8TBps": N_INST * K / dt * ALGO_BYTES_PER_INST_STEP / 1e9 / HBM_PEAK_GBS,
            "semantics": "reference --accumulation_steps 8: loss/8 per bag, gradients summed, one Adam + EMA per window; Merge's query EMA "
                         "chained over the window's tokens (every bag attends with the window's first queries: second order in 1 - merge_mm)"}


def reference_loop(a, dev, bags, labels):
    """The reference trainer's OWN loop body (engines/base_engine.py:76-167: forward_func, criterion, loss.backward(), optimizer.step(),
    zero_grad, the per-parameter EMA loop) on the drop-in classes with optim.FusedAdamEMA and CommonMIL(fused=, graph_cache=): what a user
    gets from the two-line swap of INTEGRATION.md section 2, beside the native FusedTrainer step the headline times."""
    import types
    from mhim_mil_amd.engine import CommonMIL
    from mhim_mil_amd.optim import FusedAdamEMA
    steps, warm, mm = 48, 8, 0.9997
    model, ema, _ = make_models(dev, a.prec)
    args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5, main_alpha=1.0)
    opt = FusedAdamEMA(model, ema, lr=2e-4, weight_decay=1e-5, mm=mm)
    engine = CommonMIL(args, fused=opt, graph_cache=0 if a.no_graph else 2)
    crit = torch.nn.CrossEntropyLoss()

    def step(i):
        bag, label = bags[i % len(bags)][None], labels[i % len(bags)]
        logits, lab, aux, _, _, _, _ = engine.forward_func(args, model, ema, bag, label, crit, 1, i, 0, i, None)
        loss = args.main_alpha * crit(logits.view(1, -1), lab) + args.aux_alpha * aux
        loss.backward()
        opt.step()
        opt.zero_grad()
        for pq, pk in zip(model.parameters(), ema.parameters()):            # base_engine.py:166-167 (an adopted teacher yields nothing)
            pk.data.mul_(mm).add_(pq.data, alpha=1. - mm)

    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": N_INST / dt, "unit": "patch-instances/s", "ms_per_step": 1e3 * dt, "steps": steps, "warmup": warm,
            "loop": "the call sequence of engines/base_engine.py:76-167 (forward_func, nn.CrossEntropyLoss, loss.backward(), optimizer.step(), "
                    "zero_grad(), per-parameter EMA loop) with optim.FusedAdamEMA + CommonMIL(args, fused=optimizer, graph_cache=2)",
            "launch": "eager native step" if a.no_graph else "the native forward + backward replayed as a hipGraph per bag shape; bag copied into the graph's buffer"}


def run_extras(a, dev, bags, labels):
    """The legs the default one-GPU run adds to the headline line; each catches its own failure so the headline always prints."""
    import copy
    out = {}
    for name, fn in (("hbm_copy", lambda: hbm_copy_rate(dev)), ("accumulate8", lambda: accumulate8(a, dev, bags, labels)),
                     ("reference_loop", lambda: reference_loop(a, dev, bags, labels))):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        torch.cuda.synchronize()
    others = {}
    for wl, steps, warm in (("c3", 20, 3), ("c5", 24, 4)):
        b = copy.copy(a)
        b.workload, b.steps, b.warmup = wl, steps, warm
        try:
            torch.cuda.empty_cache()
            r = other_workload(b, 1, 0, dev, embedded=True)
            others[wl] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "roofline", "whole_step_hbm_roofline",
                                            "cpu_baseline", "config") if k in r}
        except Exception as e:  # noqa: BLE001
            others[wl] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        torch.cuda.synchronize()
    out["other_workloads"] = others
    return out


def self_launch(a):
    """`python bench.py --gpus N` without a launcher (the shape of the driver's one-GPU command): re-exec under torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1 (the reference's own multi-process entry is torchrun too: options.py:181,287)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def pick_collective(a, trainer, graphs, bags, labels, world, dev):
    """N > 1: time a few steps of every form the gradient exchange can take and run the timed region on the fastest (the
    ring all-reduce is estimated at 4.2x at 8 GPUs, the mesh form at 6.5x - profiles/r03_dp_wire_time.md; the box decides).  Returns
    (step function, name, {name: ms per step or error})."""
    from mhim_mil_amd import comm as CM
    results, forms = {}, {}

    def timed_ms(step, n=5, warm=2):
        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(warm + i)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        return 1e3 * float(tt.item()) / n

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item())

    if graphs is not None:
        forms["graph | torch.distributed all_reduce (RCCL picks ring / tree) | graph"] = (None, lambda i: graphs[i % N_BAGS].replay())
        native, err = None, None
        if os.environ.get("MHIMX_BENCH_SELFTEST") == "1":
            err = "skipped: the self-test's ranks share one device (RCCL refuses two ranks on a device)"
        else:
            try:
                native = CM.NativeComm(int(os.environ.get("RANK", "0")), world, mode=1)
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {str(e)[:120]}"
        if agree(native is not None):
            forms["graph | mhimx_comm_allreduce mode 1 (reduce-scatter + all-gather over the xGMI mesh) | graph"] = (native, lambda i: graphs[i % N_BAGS].replay())
        else:
            results["mhimx_comm_allreduce mode 1"] = err or "another rank could not create the communicator"
    forms["eager steps, all-reduce in two pieces, the first under the dW1 GEMM"] = ("eager", lambda i: trainer.train_step(bags[i % N_BAGS], labels[i % N_BAGS]))
    for name, (cm, step) in forms.items():
        trainer.comm = cm if (cm is not None and cm != "eager") else None
        try:
            ms = timed_ms(step)
            ok = True
        except Exception as e:  # noqa: BLE001
            ms, ok = f"{type(e).__name__}: {str(e)[:120]}", False
        if agree(ok):
            results[name] = ms
        else:
            results[name] = ms if not ok else "failed on another rank"
    # the bare exchange of every form: the flat gradient buffer all-reduced alone, bus bandwidth = 2 (W - 1) / W x bytes / time (the figure
    # rccl-tests prints: what a link carries) - printed beside the step times so that a slow step can be told from a slow wire
    buf = trainer.flat.grad[:trainer.flat.n_train]
    nbytes = buf.numel() * 4
    bus = {}

    def wire(name, fn):
        try:
            keep = buf.clone()
            ms = timed_ms(lambda i: fn(), n=10, warm=3)
            buf.copy_(keep)
            ok = True
        except Exception as e:  # noqa: BLE001
            ms, ok = f"{type(e).__name__}: {str(e)[:120]}", False
        if agree(ok):
            bus[name] = {"bytes": nbytes, "ms": ms, "bus_GBps": 2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9}
        else:
            bus[name] = ms if not ok else "failed on another rank"

    wire("torch.distributed all_reduce", lambda: torch.distributed.all_reduce(buf))
    for name, (cm, _) in forms.items():
        if cm is not None and cm != "eager":
            wire("mhimx_comm_allreduce mode 0 (ncclAllReduce)", lambda cm=cm: cm.allreduce(buf, mode=0))
            wire("mhimx_comm_allreduce mode 1 (reduce-scatter + all-gather)", lambda cm=cm: cm.allreduce(buf, mode=1))
    results["flat_gradient_all_reduce_alone"] = bus
    best = min((n for n in forms if isinstance(results.get(n), float)), key=lambda n: results[n])
    cm, step = forms[best]
    trainer.comm = cm if (cm is not None and cm != "eager") else None
    return step, best, results


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # MHIMX_BENCH_SELFTEST=1 (one-GPU boxes only): every rank on cuda:0 over gloo - exercises the multi-rank control flow of this
    # script (barriers, max-over-ranks timing, rank-0 JSON, the overlapped all-reduce); its numbers mean nothing
    selftest = os.environ.get("MHIMX_BENCH_SELFTEST") == "1"
    if selftest:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.workload != "c2":
        res = other_workload(a, world, rank, dev)
        if res is not None:
            print(json.dumps(res), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from mhim_mil_amd import ops
    from mhim_mil_amd.engine import FusedTrainer

    torch.manual_seed(1234 + rank)
    student, teacher, base = make_models(dev, a.prec)
    trainer = FusedTrainer(student, teacher, aux_alpha=0.5, mm=0.9997)
    # distinct synthetic bags per rank, resident in HBM: |N(0,1)| patch features (post-ReLU-like)
    g = torch.Generator(device=dev)
    g.manual_seed(1000 * 2 + rank)
    bags = [torch.randn(N_INST, D_IN, device=dev, generator=g).abs_() for _ in range(N_BAGS)]
    labels = [torch.tensor([(i + rank) % 2], device=dev) for i in range(N_BAGS)]

    # per-kernel HIP events on the launch stream for the dominant kernel (the N x D -> E feature projection)
    ev = []
    spin = [False]                             # (only in the dedicated eager pass: never under capture, never in a timed region)
    hook = None
    if not a.no_kernel_events:
        def hook(tag, M, N, K):
            if tag == "bag_project" and M >= N_INST and K == D_IN:                # the teacher + student projection launch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev.append((e0, e1))
                # eager launches are host-bound (~20 us of Python per launch): with an idle stream the first event is processed long
                # before the kernel's packet arrives and the interval contains that host gap (it read 69 us for a kernel the trace times
                # at 60).  ~100 us of spin queued first keeps event, kernel, event back to back in the queue.
                if spin[0]:
                    torch.cuda._sleep(200000)
                return e0, e1
            return None

    graphs, graph_note = None, None
    # steps per captured graph (one GPU): every chunk is `chunk` complete steps; warm-up and timed region are whole numbers of chunks
    chunk = 1
    if world == 1 and not a.no_graph:
        g0 = math.gcd(a.steps, a.warmup) if a.warmup > 0 else a.steps
        chunk = a.steps_per_graph if a.steps_per_graph > 0 else max(d for d in range(1, 9) if g0 % d == 0)
        if a.steps % chunk or a.warmup % chunk:
            chunk = 1
    # N > 1 defaults to graph(fwd+bwd) | eager RCCL all-reduce | graph(Adam+EMA): no measurement on more than one GPU exists yet that
    # shows the eager form (all-reduce of every gradient but the projection's started in the middle of the backward,
    # FusedTrainer._mid_hook: hides up to ~80 us of collective, pays ~3 % of eager launches at N = 1) winning; --dp-eager selects it.
    # The launch form used is printed in config.launch.
    if not a.no_graph and (world == 1 or not a.dp_eager):
        # one captured hipGraph per resident bag (the bag pointer is a kernel argument); they share one memory pool.
        # Each replay runs the complete step: prep, teacher fwd, select, student fwd, head, bwd, [all-reduce], Adam + EMA.
        # If capture is refused (e.g. a collective that cannot be captured on this RCCL build) every rank falls back to
        # eager launches together — the decision is all-reduced so that no rank replays while another launches eagerly.
        ok = 1
        try:
            if world == 1 and chunk > 1:
                # one graph per CHUNK of consecutive steps (each a complete step on its own bag, with its own update): the rotation over the
                # resident bags continues from chunk to chunk, N_BAGS / gcd(chunk, N_BAGS) distinct graphs
                n_graphs = N_BAGS // math.gcd(chunk, N_BAGS)
                graphs = [trainer.capture_steps([bags[(q * chunk + j) % N_BAGS] for j in range(chunk)],
                                                [labels[(q * chunk + j) % N_BAGS] for j in range(chunk)], warmup=1 if q == 0 else 0)
                          for q in range(n_graphs)]
            else:
                graphs = [trainer.capture(bags[i], labels[i], warmup=1) for i in range(N_BAGS)]
            for g_ in graphs:               # part of the capture: one replay per graph right after instantiation (the first launch of a
                g_.replay()                 # hipGraphExec uploads it; with W < N_BAGS warm-up steps that would fall into the timed region)
            torch.cuda.synchronize()
        except Exception as exc:            # noqa: BLE001 — any capture failure means "run eagerly"
            ok, graph_note = 0, f"{type(exc).__name__}: {str(exc)[:160]}"
            torch.cuda.synchronize()
        if world > 1:
            flag = torch.tensor([ok], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            graphs = None
            graph_note = graph_note or "another rank could not capture"

    def step(i):
        if graphs is not None:
            if chunk > 1:
                if i % chunk == 0:                          # (one replay = `chunk` steps: the calls in between are already done)
                    graphs[(i // chunk) % len(graphs)].replay()
            else:
                graphs[i % N_BAGS].replay()
        else:
            trainer.train_step(bags[i % N_BAGS], labels[i % N_BAGS])

    collective, collective_ms = None, None
    if world > 1 and not a.dp_eager and not a.no_graph:
        step, collective, collective_ms = pick_collective(a, trainer, graphs, bags, labels, world, dev)
        if collective.startswith("eager"):
            graphs = None

    events_from = "the timed region"
    ev_eager = None
    proj_ms = proj_empty_ms = None
    if graphs is not None and not a.no_kernel_events:
        # Host-side HIP event records cannot be placed between the nodes of a replayed hipGraph (ROCm rejects external
        # event nodes), so the dominant kernel is bracketed in an eager pass of the SAME steps right BEFORE the warm-up and the
        # timed region (it used to run after them; before, its ~8 ms of launches also settle the clocks that a 5 + 20-step run
        # would otherwise still be ramping through); profiles/ holds the rocprofv3 trace of the graph replays themselves.
        ev.clear()
        n_ev = min(a.steps, 20)
        if world == 1 and trainer._exec_ok(bags[0]):
            # (round 6) the SAME issue path as the captured steps: mhimx_step_run itself brackets its projection launch with a pair of HIP
            # events on the launch stream (mhimx_step_cfg.time_project).  The executor enqueues a step faster than the GPU runs it, so the
            # stream never idles and event, kernel, event sit back to back in the queue: the bracket reads within ~2 % of the rocprofv3
            # duration (the round-5 pass ran the host-bound Python orchestration with a spin kernel in front and read + 5 %).
            import ctypes as _C
            from mhim_mil_amd import _lib as _L
            trainer.time_project = True
            for i in range(4 + n_ev):
                trainer.train_step(bags[i % N_BAGS], labels[i % N_BAGS])
            torch.cuda.synchronize()
            trainer.time_project = False
            buf, buf_e = (_C.c_float * 256)(), (_C.c_float * 256)()
            n_got = _L.lib().mhimx_step_project_ms(buf, buf_e, 256)
            proj_ms = [float(buf[j]) for j in range(max(0, n_got))][4:]          # (the first four steps settle clocks and caches)
            proj_empty_ms = [float(buf_e[j]) for j in range(max(0, n_got))][4:]
            events_from = (f"an eager pass of {n_ev} steps through mhimx_step_run right before the warm-up steps, the projection launch bracketed by "
                           f"the executor itself (graph nodes cannot carry host events)")
        else:
            ops.KERNEL_EVENT_HOOK, spin[0] = hook, True
            for i in range(n_ev):
                trainer.train_step(bags[i % N_BAGS], labels[i % N_BAGS])
            torch.cuda.synchronize()
            ops.KERNEL_EVENT_HOOK, spin[0] = None, False
            ev_eager = list(ev)
            events_from = f"an eager pass of {n_ev} steps right before the warm-up steps (graph nodes cannot carry host events)"
        ops.KERNEL_EVENT_HOOK = None
    if graphs is None:
        ops.KERNEL_EVENT_HOOK = hook            # eager steps: the events of the timed steps themselves
    dt = timed(a, world, dev, step)
    # (round 6) the line calibrates itself: the SAME region - K steps between two synchronisations - five more times right
    # after the headline one.  `value` stays the first region (the contract: exactly K timed steps after W warm-up steps); the spread of the
    # repeats is what a claimed same-box gain has to exceed.
    repeats = None
    if world == 1 and not a.no_repeats:
        rep = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(a.warmup + i)
            torch.cuda.synchronize()
            rep.append((time.perf_counter() - t0) / a.steps * 1e3)
        srt = sorted(rep)
        repeats = {"n": len(rep), "steps_each": a.steps, "ms_per_step": [round(v, 5) for v in rep], "min": srt[0], "median": srt[len(srt) // 2], "max": srt[-1],
                   "spread_pct": 100.0 * (srt[-1] - srt[0]) / srt[len(srt) // 2],
                   "note": "the headline region repeated right after it (not part of `value`): min / median / max ms per step"}
    if graphs is None:
        ev[:] = ev[-a.steps:]                   # eager: keep the events of the timed steps only
    elif ev_eager is not None:
        ev[:] = ev_eager
    ops.KERNEL_EVENT_HOOK = None

    extras = {}
    if world == 1 and not a.no_extras:
        extras = run_extras(a, dev, bags, labels)

    if rank == 0:
        value = N_INST * a.steps * world / dt
        out = {
            "metric": "patch-instances/sec through MHIM fwd+bwd, N=10k D=1024", "value": value, "unit": "patch-instances/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic |N(0,1)| bags resident in HBM, random-init weights (reference init law)",
            "config": {"workload": "c2: MHIM(ABMIL) train step, one bag N=10000 D=1024 per GPU per step "
                                   "(teacher fwd + select + student fwd + bwd + Adam + EMA"
                                   + (" + RCCL all-reduce of the 6.6 MB flat gradient" if world > 1 else "") + ")",
                       "bags_per_step": world, "rotating_bags_per_gpu": N_BAGS, "matrix_core_form": student._feature_prec(N_INST),
                       "dropout": CFG["dropout"], "parallelism": f"dp{world}",
                       "step_issued_by": ("mhimx_step_run (csrc/step.hip: the step's launches enqueued by ONE C call; captured into the graphs)"
                                          if trainer._exec is not None else "the Python orchestration of engine.py (one ctypes call per launch)"),
                       **({"collective": collective, "collective_candidates_ms_per_step": collective_ms} if collective else {}),
                       "launch": (("eager" if world == 1 else "eager, gradient all-reduce in two pieces, the first overlapped with the dW1 GEMM of the backward")
                                  + (f" (graph capture failed: {graph_note})" if graph_note else "")) if graphs is None
                                 else ((f"hipGraph replay, {chunk} consecutive complete steps (one bag + one update each) per graph" if chunk > 1 else
                                        "hipGraph replay, one graph per resident bag") if world == 1 else
                                       "hipGraph replay per resident bag: graph(fwd+bwd) | eager RCCL all-reduce | graph(Adam+EMA)")},
            "whole_step_hbm_roofline": {"algorithmic_bytes_per_instance": ALGO_BYTES_PER_INST_STEP,
                                        "achieved_GBps": value / world * ALGO_BYTES_PER_INST_STEP / 1e9,
                                        "frac_of_8TBps": value / world * ALGO_BYTES_PER_INST_STEP / 1e9 / HBM_PEAK_GBS},
        }
        if repeats is not None:
            out["region_repeats"] = repeats
        if ev or proj_ms:
            ms = proj_ms if proj_ms else [e0.elapsed_time(e1) for e0, e1 in ev]
            avg_bracket = sum(ms) / len(ms)
            # what the event pair itself reads: an EMPTY bracket recorded right behind every timed one (round 6; the bracket read + 5 % over
            # the rocprofv3 duration of the same launch in rounds 4-5).  avg = bracket - empty bracket: the kernel's own time
            empty = sum(proj_empty_ms) / len(proj_empty_ms) if proj_empty_ms else 0.0
            # MEASURED (profiles/r06_kernel_timing.md): bracket 68.2 us, empty bracket 5.3 us, rocprofv3 of the same launch 65.2 us - the
            # truth lies between bracket and bracket - empty.  `avg_kernel_ms` (what `frac` is computed from) stays the BRACKET: the
            # conservative reading; the corrected one and the committed rocprofv3 average sit beside it.
            avg = avg_bracket
            rocprof_us = None
            rp = os.path.join(ROOT, "profiles", "r06_kernel_avg.json")
            if os.path.exists(rp):
                rocprof_us = json.load(open(rp)).get("bag_project_ws_kernel_avg_us")
            # The binding roofline of this kernel is the MATRIX CORE, not HBM: it issues 3 bf16 MFMA terms per product (fp32-class accuracy
            # for the instance scores that feed a top-k) = 3 x 2 x N x D x 1024 flop per launch against 2.5 PFLOP/s dense bf16; its HBM
            # floor (X read ONCE + 2 weight images + H_teacher, H_student fp32 + fp16 d out/d pre written) is ~12 us of the ~72.
            flops_bf16 = 3 * 2.0 * N_INST * D_IN * 1024
            tf = flops_bf16 / (avg * 1e-3) / 1e12
            read_once = N_INST * D_IN * 4 + 2 * 512 * D_IN * 4
            written = 2 * N_INST * 512 * 4 + N_INST * 512 * 2
            budget = 2 * N_INST * D_IN * 4        # SURVEY 8(d) credits this launch with TWO of the step's three passes over X
            # HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes of this same command
            # (tools/pmc.sh + tools/pmc_project.py; counters cannot be read from inside the process being timed)
            traffic, tsrc = None, None
            for pmc_name in ("r06_pmc_bag_project.json", "r05_pmc_bag_project.json", "r04_pmc_bag_project.json", "r03_pmc_bag_project.json", "r02_pmc_bag_project.json"):
                pmc = os.path.join(ROOT, "profiles", pmc_name)
                if os.path.exists(pmc):
                    pj = json.load(open(pmc))
                    traffic, tsrc = pj["traffic_bytes"], (f"profiles/{pmc_name}: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, "
                                                          f"KiB, separate --pmc passes; read {pj['fetch_bytes'] / 1e6:.1f} MB + write "
                                                          f"{pj['write_bytes'] / 1e6:.1f} MB per launch")
                    break
            step_traffic = None
            pmc = next((q for q in (os.path.join(ROOT, "profiles", nm) for nm in ("r06_pmc_traffic_c2.json", "r05_pmc_traffic_c2.json")) if os.path.exists(q)), "")
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                step_traffic = {"bytes_per_step": pj["step_traffic_bytes"], "read": pj["step_read_bytes"], "write": pj["step_write_bytes"],
                                "over_algorithmic": pj["traffic_over_algorithmic"],
                                "source": f"profiles/{os.path.basename(pmc)[:-5]}.md: FETCH_SIZE x2 + WRITE_SIZE over every launch of one step (two --pmc passes)"}
            # SURVEY 8(d) / BASELINE.md section 4: the HBM fraction on the ALGORITHMIC bytes is the headline figure of the dominant kernel - this
            # launch stands for two of the step's three passes over X (4096 B per instance each); the matrix-core figures sit beside it.
            # attainable: every product runs as 3 bf16 terms (section 3 of DESIGN.md: two terms miss the 1e-4 logit bound under peaked
            # attention), so the kernel cannot run faster than its issued flop at the dense peak: 63 GF / 2.5 PF = 25 us = 0.41 of the HBM
            # roofline on this basis; the whole step's 3-term floor (118 GF) is 47 us = 0.33 of the step-level roofline - the 50 % target
            # of north_star is above what these numerics can reach on this chip.
            mfma_floor_ms = flops_bf16 / 2500e12 * 1e3
            out["roofline"] = {"kernel": "bag_project_ws_kernel (teacher AND student feature projection X[N,D] -> 2 x H[N,512] in one pass over the raw fp32 bag, 3-term bf16 MFMA, fused bias+GELU+dropout, fp16 d out/d pre)",
                               "bound": "hbm", "achieved": budget / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": budget / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "basis": "SURVEY 8(d) algorithmic bytes: two passes over X (teacher fwd + student fwd), 2 x 4096 B per instance per launch",
                               "mfma_issued_TFLOPs": tf, "mfma_issued_frac": tf / 2500.0, "mfma_useful_frac": tf / 3.0 / 2500.0,
                               "attainable_ceiling": {"kernel_frac_of_hbm_roofline": (budget / 1e9 / HBM_PEAK_GBS * 1e3) / mfma_floor_ms,
                                                      "kernel_mfma_floor_ms": mfma_floor_ms,
                                                      "step_frac_of_hbm_roofline": (N_INST * ALGO_BYTES_PER_INST_STEP / 1e9 / HBM_PEAK_GBS * 1e3) / (3 * 39.3e9 / 2500e12 * 1e3),
                                                      "step_mfma_floor_ms": 3 * 39.3e9 / 2500e12 * 1e3,
                                                      "why": "3 bf16 MFMA terms per product (fp32-class instance scores feed a top-k; 2 terms miss the 1e-4 logit bound under peaked attention: profiles/r03_two_term.md): the dense-peak time of the issued flop is the floor"},
                               "flops_note": "issued = bf16 MFMA flop actually issued: 3 terms (hi*hi + hi*lo + lo*hi) x 2 N D 1024; useful = fp32-equivalent = a third",
                               "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": tsrc, "step_traffic": step_traffic, "avg_kernel_ms": avg,
                               "avg_bracket_ms": avg_bracket, "empty_bracket_ms": empty, "avg_kernel_ms_minus_empty_bracket": avg_bracket - empty,
                               "rocprofv3_avg_kernel_ms": None if rocprof_us is None else rocprof_us * 1e-3,
                               "frac_from_rocprofv3": None if rocprof_us is None else budget / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "rocprofv3_source": "profiles/r06_kernel_avg.json <- profiles/r06_kernel_trace_bench_c2_graph.md (rocprofv3 --kernel-trace --stats of this command)",
                               "launches_timed": len(ms), "hip_events_over": events_from,
                               "hbm": {"basis": "bytes the launch must move: X read once + both weight images + H_teacher, H_student (fp32) and d out/d pre (fp16) written",
                                       "bytes_per_launch": read_once + written, "achieved_GBps": (read_once + written) / (avg * 1e-3) / 1e9,
                                       "frac_of_8TBps": (read_once + written) / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "read_only_bytes": read_once, "read_only_frac_of_8TBps": read_once / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS},
                               "hbm_budget_basis": {"basis": "SURVEY 8(d) algorithmic budget: two of the step's three passes over X (teacher fwd + student fwd, 4096 B/instance each) are this one launch",
                                                    "bytes_per_launch": budget, "achieved_GBps": budget / (avg * 1e-3) / 1e9,
                                                    "frac_of_8TBps": budget / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS},
                               "hbm_copy_peak_GBps_measured": (extras.get("hbm_copy") or {}).get("GBps")}
        if world == 1 and not a.no_extras and "roofline" in out:
            clk = sustained_clock()
            if clk:
                # the matrix pipe's ceiling at the clock the launch really holds (the dense peak is quoted at 2.4 GHz)
                peak_here = 2500.0 * clk["GHz"] / 2.4
                clk["mfma_peak_TFLOPs_at_this_clock"] = peak_here
                clk["mfma_issued_frac_at_this_clock"] = out["roofline"]["mfma_issued_TFLOPs"] / peak_here
                out["roofline"]["sustained_clock_GHz"] = clk
        out.update(extras)
        if world == 1 and a.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(a.cpu_steps, base)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
