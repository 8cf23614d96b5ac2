"""Round-4 regressions (GPU box): stale step images after a data-parallel update, the query chain when the ranks of one
update take different step paths, the ABI version gate."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests.test_sharded_gpu import V2, D, N, DEV, _models

pytestmark = pytest.mark.gpu


def test_apply_drops_the_step_images():
    """FusedTrainer._apply changes the weights: whatever weight images the step installed (ops.step_images) must be gone afterwards, also
    when the update does not pass through update() (the captured data-parallel step calls _apply directly)."""
    from mhim_mil_amd import ops
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models()
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999)
    x = torch.from_numpy(synth.bag(5, N, D)).to(DEV)
    tr.forward_backward(x, torch.tensor([1], device=DEV))
    w = s.feature[0].weight.data
    ops.step_images({(w.data_ptr(), False): torch.zeros_like(w)})
    assert ops.pair_planes(w).abs().max().item() == 0.0            # (the table is what pair_planes hands out)
    tr._apply(1.0)
    assert not ops._STEP_IMAGES
    assert ops.pair_planes(w).abs().max().item() > 0.0
    torch.cuda.synchronize()


# one rank's bag is too small for the single-pass step (bag_ordered_ok needs >= 64 rows): it takes the generic step
N_SMALL = 48


def _mixed_bag(step, rank):
    n = N_SMALL if rank == 0 else N
    return torch.from_numpy(synth.bag(2700 + 10 * step + rank, n, D)).to(DEV), torch.tensor([(step + rank) % 2], device=DEV)


def _mixed_draws(step, rank):
    n = N_SMALL if rank == 0 else N
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    return (torch.from_numpy(synth.permutation(150 + 2 * step + rank, k)).to(DEV),
            torch.from_numpy(synth.permutation(170 + 2 * step + rank, n - n_sel)).to(DEV))


def _mixed_worker(rank, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models()
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999)
    assert tr._chain is not None
    paths = []
    orig = tr._forward_backward_nat
    tr._forward_backward_nat = lambda *a, **k: (paths.append("nat"), orig(*a, **k))[1]
    for step in range(2):
        x, lab = _mixed_bag(step, rank)
        perm, shuf = _mixed_draws(step, rank)
        tr.train_step(x, lab, perm=perm, ids_shuffle=shuf)
    torch.cuda.synchronize()
    assert len(paths) == (0 if rank == 0 else 2), paths            # rank 0: the generic step, rank 1: the single-pass step
    torch.save({"stu": {k: v.detach().cpu() for k, v in s.state_dict().items()},
                "tea": {k: v.detach().cpu() for k, v in t.state_dict().items()}}, os.path.join(out, f"mx{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_ranks_on_different_step_paths_keep_one_query_chain(tmp_path):
    """Two ranks of one update, one on the generic ABMIL step (a 48-row bag) and one on the single-pass step: both put their Merge tokens
    into the chain, the replicas stay bit-identical (merge.global_q_mm included) and equal the single process with accumulation_steps = 2."""
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models()
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999, accumulation_steps=2)
    for step in range(2):
        for rank in range(2):
            x, lab = _mixed_bag(step, rank)
            perm, shuf = _mixed_draws(step, rank)
            tr.train_step(x, lab, perm=perm, ids_shuffle=shuf)
    torch.cuda.synchronize()
    s_ref = {k: v.detach().cpu() for k, v in s.state_dict().items()}
    port = 41500 + (os.getpid() % 1000)
    mp.spawn(_mixed_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(tmp_path, f"mx{r}.pt")) for r in range(2)]
    for k in res[0]["stu"]:
        assert torch.equal(res[0]["stu"][k], res[1]["stu"][k]), k
        assert torch.equal(res[0]["tea"][k], res[1]["tea"][k]), k
    for k, v in s_ref.items():
        err = (res[0]["stu"][k].double() - v.double()).abs()
        if "global_q" in k:
            assert err.max().item() <= 3e-6, (k, err.max().item())
        else:
            assert err.mean().item() <= 2e-6 and err.max().item() <= 2 * 4.1e-4, (k, err.mean().item(), err.max().item())


@pytest.mark.parametrize("n,ratio_h,ratio_hr", [(200, 0.06, 0.5), (700, 0.02, 0.25), (96, 0.25, 0.5)])
def test_select_rows_draws_are_fair_on_small_lists(n, ratio_h, ratio_hr):
    """mhimx_select_rows itself (feistel_small: 4 rounds on domains of a few bits), both draws - the n_sel-subset of the k candidates
    (masking.py:66-71) and the R-subset of the kept rows (merge.py:158-176): over T seeds every candidate is masked n_sel / k of the time
    and every kept row merged R / L of the time, within 4.5 sigma of the binomial; the LEAN and the mask-id form draw the same rows."""
    from mhim_mil_amd import ops
    s = ((synth.permutation(15, n) + 0.25 * synth.uniform(16, (n,))) / n).astype(np.float32)
    k, n_sel, _ = O.mask_count(n, ratio_h, ratio_hr)
    L_ = n - n_sel
    Lk = int(L_ * 0.9)
    R = L_ - Lk
    assert 1 <= n_sel < k and R >= 1
    top = np.array(sorted(O.topk_indices(s, k, True).tolist()))
    sd = torch.from_numpy(s).to(DEV)
    T = 600
    cnt_mask, cnt_merge, kept_seen = np.zeros(n), np.zeros(n), np.zeros(n)
    for seed in range(T):
        r = ops.select_rows(sd, k, n_sel, R, 7919 * seed + 3)
        if seed < 5:
            r2, _ = ops.select_rows(sd, k, n_sel, R, 7919 * seed + 3, want_mask_ids=True)
            assert torch.equal(r, r2)
        r = r.cpu().numpy()
        m = np.ones(n, bool); m[r] = False
        assert m.sum() == n_sel and set(np.nonzero(m)[0].tolist()) <= set(top.tolist())
        cnt_mask[m] += 1
        cnt_merge[r[Lk:]] += 1
        kept_seen[r] += 1
    p = n_sel / k
    f = cnt_mask[top] / T
    assert abs(f.mean() - p) < 1e-9
    assert np.abs(f - p).max() < 4.5 * np.sqrt(p * (1 - p) / T), (f.min(), f.max(), p)
    never = np.setdiff1d(np.arange(n), top)                       # rows that are kept under every seed: merged R / L of the time
    q = R / L_
    g = cnt_merge[never] / T
    assert np.abs(g - q).max() < 4.5 * np.sqrt(q * (1 - q) / T) + 1e-9, (g.min(), g.max(), q)


# ------------------------------------------------------------------------------------------------------------------------------
# the reference trainer's loop with the fused optimiser, and the factory's teacher seam
# ------------------------------------------------------------------------------------------------------------------------------
def _ref_loop_steps(model, ema, optimizer, own_ema, mm, steps, fused=None):
    """The body of base_engine.py:76-167 for args.model == 'mhim' (forward_func, criterion, backward, optimizer.step, zero_grad, the
    per-parameter EMA loop) on injected draws."""
    import types
    from mhim_mil_amd.engine import CommonMIL
    args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5, main_alpha=1.0)
    eng = CommonMIL(args, fused=fused)
    crit = torch.nn.CrossEntropyLoss()
    out = []
    for step in range(steps):
        x = torch.from_numpy(synth.bag(3300 + step, N, D)).to(DEV)[None]
        label = torch.tensor([step % 2], device=DEV)
        k, n_sel, _ = O.mask_count(N, V2["mask_ratio_h"], V2["mask_ratio_hr"])
        perm = torch.from_numpy(synth.permutation(250 + step, k)).to(DEV)
        shuf = torch.from_numpy(synth.permutation(270 + step, N - n_sel)).to(DEV)
        logits, lab, aux, pn, kn, _, _ = eng.forward_func(args, model, ema, x, label, crit, 1, step, 0, step, None, perm=perm, ids_shuffle=shuf)
        loss = args.main_alpha * crit(logits.view(1, -1), lab) + args.aux_alpha * aux
        loss.backward()
        optimizer.step()
        eng.after_backward_func(args, model=model, others=None, num_updates=step)
        optimizer.zero_grad()
        n_updated = 0
        for pq, pk in zip(model.parameters(), ema.parameters()):                # base_engine.py:166-167
            pk.data.mul_(mm).add_(pq.data, alpha=1. - mm)
            n_updated += 1
        assert (n_updated == 0) == (not own_ema)
        out.append(float(loss))
    torch.cuda.synchronize()
    return out


def test_fused_adam_ema_under_the_reference_loop_equals_torch_adam_plus_ema_loop():
    from mhim_mil_amd.optim import FusedAdamEMA
    mm = 0.99
    s1, t1 = _models()
    for p in t1.parameters():
        p.requires_grad_(False)
    opt1 = torch.optim.Adam([p for p in s1.parameters() if p.requires_grad], lr=2e-4, weight_decay=1e-5)
    l1 = _ref_loop_steps(s1, t1, opt1, True, mm, 3)
    s2, t2 = _models()
    opt2 = FusedAdamEMA(s2, t2, lr=2e-4, weight_decay=1e-5, mm=mm)
    assert list(t2.parameters()) == [] and len(list(t2.named_parameters())) == len(list(s2.named_parameters()))
    l2 = _ref_loop_steps(s2, t2, opt2, False, mm, 3)
    np.testing.assert_allclose(l1, l2, rtol=0, atol=2e-5)
    for (n, a), (_, b) in zip(s1.state_dict().items(), s2.state_dict().items()):
        if "global_q" in n:
            continue                                                            # (Merge's in-forward EMA of the queries: both loops do it)
        err = (a.double() - b.double()).abs()
        assert err.mean().item() <= 2e-6 and err.max().item() <= 3 * 4.1e-4, (n, err.mean().item(), err.max().item())
    for (n, a), (_, b) in zip(t1.state_dict().items(), t2.state_dict().items()):
        err = (a.double() - b.double()).abs().max().item()
        assert err <= 3e-5, ("teacher", n, err)
    # the flat gradient is zero after an update, .grad still bound to it
    assert float(opt2.flat.grad.abs().max()) == 0.0
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in opt2._views)
    # CommonMIL(args, fused=optimizer): forward_func runs the native forward + backward, the loop's criterion / backward see two leaves
    s3, t3 = _models()
    opt3 = FusedAdamEMA(s3, t3, lr=2e-4, weight_decay=1e-5, mm=mm)
    calls = []
    orig = opt3.trainer.forward_backward
    opt3.trainer.forward_backward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    l3 = _ref_loop_steps(s3, t3, opt3, False, mm, 3, fused=opt3)
    assert len(calls) == 3
    np.testing.assert_allclose(l1, l3, rtol=0, atol=2e-5)
    for (n, a), (_, b) in zip(s1.state_dict().items(), s3.state_dict().items()):
        if "global_q" in n:
            continue
        err = (a.double() - b.double()).abs()
        assert err.mean().item() <= 2e-6 and err.max().item() <= 3 * 4.1e-4, ("native", n, err.mean().item(), err.max().item())
    for (n, a), (_, b) in zip(t1.state_dict().items(), t3.state_dict().items()):
        assert (a.double() - b.double()).abs().max().item() <= 3e-5, ("native teacher", n)


def test_factory_teacher_seam_on_the_device():
    """modules/__init__.py:176-214 on device-resident models: deepcopy of a student whose parameters are views of a trainer's flat buffer,
    --teacher_init of an mhim_pure checkpoint saved from DistributedDataParallel (module. keys, no merge.*) with strict=False,
    merge_test False, others['model_ema'] / ['mm_sche']; the pair then trains under FusedTrainer."""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.standalone import build_teacher
    s, t0 = _models()
    tr0 = FusedTrainer(s, t0, aux_alpha=0.5, mm=0.999)                           # (s's parameters are views of tr0's flat buffer now)
    x = torch.from_numpy(synth.bag(77, N, D)).to(DEV)
    tr0.train_step(x, torch.tensor([1], device=DEV))
    others = {}
    pure = {("module." + k): (v.detach().clone() + 0.25) for k, v in s.state_dict().items() if not k.startswith("merge.")}
    tea = build_teacher(s, others, teacher_init={"model": pure}, mm_sche=None)
    assert others["model_ema"] is tea and others["mm_sche"] is None and tea.merge_test is False and tea is not s
    info = others["teacher_init_info"]
    assert sorted(info.missing_keys) == sorted(k for k in s.state_dict() if k.startswith("merge.")) and not info.unexpected_keys
    for k, v in s.state_dict().items():
        tv = tea.state_dict()[k]
        assert tv.is_cuda and tv.data_ptr() != v.data_ptr()
        want = v if k.startswith("merge.") else v + 0.25
        assert torch.equal(tv, want), k
    before = {k: v.clone() for k, v in tea.state_dict().items()}
    s.feature[0].weight.data.add_(1.0)                                          # the copy does not alias the student's flat buffer
    assert torch.equal(tea.state_dict()["feature.0.weight"], before["feature.0.weight"])
    s.feature[0].weight.data.sub_(1.0)
    same = build_teacher(s, {}, tea_type="same")
    assert same is s and s.merge_test is False
    tr = FusedTrainer(s, tea, aux_alpha=0.5, mm=0.999)
    logits, losses = tr.train_step(x, torch.tensor([0], device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and torch.isfinite(losses).all()
    # the teacher moved by the EMA of the update: (1 - mm) of the distance to the student
    d = (tea.state_dict()["feature.0.bias"] - before["feature.0.bias"]).abs().max().item()
    assert 0 < d < 1e-3


# ------------------------------------------------------------------------------------------------------------------------------
# Merge over the shards of an instance-sharded bag (mhimx_merge_fwd_part / _finish, mhimx_merge_bwd with own rows): W shards run one after
# another in ONE process against the fp64 oracle of the whole row block
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,W,order", [(970, 2, "ascending"), (1000, 4, "shuffled"), (19400, 8, "ascending"), (45, 3, "shuffled")])
def test_merge_sharded_rows_equal_the_whole_block(R, W, order):
    from mhim_mil_amd import ops
    from tests.test_ops_gpu import rnd
    E, k = 512, 5
    Nbag = 4 * R + 7 * W                                           # the bag: the merge list picks R of its rows
    sd = synth.mhim_state(7, input_dim=64, merge_k=k)
    p = {kk: torch.from_numpy(v).double() for kk, v in sd.items() if kk.startswith("merge.")}
    p["merge.norm.weight"] = p["merge.norm.weight"] + rnd(161, (E,), std=0.1).double()
    p["merge.norm.bias"] = rnd(162, (E,), std=0.1).double()
    p["merge.attn.to_out.0.bias"] = rnd(163, (E,), std=0.1).double()
    for kk in p:
        p[kk].requires_grad_(kk != "merge.global_q_mm")
    Hbag = (rnd(164, (Nbag, E)).abs() * 0.7)
    ids = np.sort(synth.permutation(165, Nbag)[:R]) if order == "ascending" else synth.permutation(165, Nbag)[:R]
    ids = np.ascontiguousarray(ids).astype(np.int64)
    X = Hbag[torch.from_numpy(ids)].double().requires_grad_(True)
    z, q_ref = O.merge_tokens(X, p, 0.9999, True)
    dz = rnd(166, (k, E), std=0.1)
    (z * dz.double()).sum().backward()
    f32 = lambda t: t.detach().float().contiguous().to(DEV)
    tr = (ops.transpose(f32(p["merge.attn.to_kv.weight"])), ops.transpose(f32(p["merge.attn.to_q.weight"])),
          ops.transpose(f32(p["merge.attn.to_out.0.weight"])))
    rows = torch.from_numpy(ids).to(DEV)
    bounds = [Nbag * r // W for r in range(W + 1)]                  # contiguous shards of the bag's rows (ragged)
    mws, parts, wss, Hloc = [], [], [], []
    for r in range(W):
        lo, n = bounds[r], bounds[r + 1] - bounds[r]
        mw = ops.MergeW(f32(p["merge.global_q_mm"]).reshape(k, E), f32(p["merge.norm.weight"]), f32(p["merge.norm.bias"]),
                        f32(p["merge.attn.to_kv.weight"]), f32(p["merge.attn.to_q.weight"]), f32(p["merge.attn.to_out.0.weight"]),
                        f32(p["merge.attn.to_out.0.bias"]), 0.9999, transposes=tr, x_rows=rows, own=(lo, n), rep=1.0 if r == 0 else 0.0)
        h = Hbag[lo:lo + n].to(DEV).contiguous()
        part, ws = ops.merge_fwd_part(mw, h)
        mws.append(mw); parts.append(part); wss.append(ws); Hloc.append(h)
    allparts = torch.stack(parts).contiguous()
    zs = [ops.merge_fwd_finish(mws[r], allparts, wss[r]) for r in range(W)]
    for zr, qr in zs:                                              # every shard finishes alike, bit for bit
        assert torch.equal(zr, zs[0][0]) and torch.equal(qr, zs[0][1])
    np.testing.assert_allclose(zs[0][0].cpu().numpy(), z.detach().float().numpy(), atol=2e-5, rtol=4e-5)
    np.testing.assert_allclose(zs[0][1].cpu().numpy(), q_ref.detach().float().numpy(), atol=1e-7, rtol=1e-6)
    names = {"d_ln_w": "merge.norm.weight", "d_ln_b": "merge.norm.bias", "d_wkv": "merge.attn.to_kv.weight", "d_wq": "merge.attn.to_q.weight",
             "d_wo": "merge.attn.to_out.0.weight", "d_bo": "merge.attn.to_out.0.bias"}
    total = {kk: 0 for kk in names}
    dX = torch.zeros((R, E))
    for r in range(W):
        lo, n = bounds[r], bounds[r + 1] - bounds[r]
        dH = torch.full((n, E), float("nan"), device=DEV)           # only this shard's merge rows may be written
        g = ops.merge_bwd(mws[r], Hloc[r], dz.to(DEV), wss[r], grads={"dX": dH})
        torch.cuda.synchronize()
        own = (ids >= lo) & (ids < lo + n)
        written = ~torch.isnan(dH[:, 0]).cpu().numpy()
        assert set(np.nonzero(written)[0].tolist()) == set((ids[own] - lo).tolist())
        dX[torch.from_numpy(np.nonzero(own)[0])] = dH[torch.from_numpy(ids[own] - lo).to(DEV)].cpu()
        for kk in names:
            total[kk] = total[kk] + g[kk].double().cpu()

    def close(name, got, ref):
        ref = ref.float().numpy()
        np.testing.assert_allclose(np.asarray(got, dtype=np.float32).reshape(ref.shape), ref, atol=2e-4 * (np.abs(ref).max() + 1e-30), rtol=1.2e-3,
                                   err_msg=name)

    close("dX", dX.numpy(), X.grad)
    for kk, nm in names.items():
        close(kk, total[kk].numpy(), p[nm].grad)


def test_capture_steps_replays_the_same_steps_as_train_step():
    """FusedTrainer.capture_steps: ONE hipGraph of consecutive complete steps == the same steps called one by one (injected draws, no
    dropout: the steps do not depend on the seed counters the capture advanced)."""
    from mhim_mil_amd.engine import FusedTrainer
    k, n_sel, _ = O.mask_count(N, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    perm = torch.from_numpy(synth.permutation(350, k)).to(DEV)
    shuf = torch.from_numpy(synth.permutation(370, N - n_sel)).to(DEV)
    bags = [torch.from_numpy(synth.bag(4100 + j, N, D)).to(DEV) for j in range(3)]
    labels = [torch.tensor([j % 2], device=DEV) for j in range(3)]
    sa, ta = _models()
    tra = FusedTrainer(sa, ta, aux_alpha=0.5, mm=0.999)
    for b, l in zip(bags, labels):
        tra.train_step(b, l, perm=perm, ids_shuffle=shuf)
    sb, tb = _models()
    trb = FusedTrainer(sb, tb, aux_alpha=0.5, mm=0.999)
    snap = [trb.flat.student.clone(), trb.flat.teacher.clone(), trb.flat.m.clone(), trb.flat.v.clone(), trb.opt_step.clone(), trb.tick.clone()]
    g = trb.capture_steps(bags, labels, warmup=1, perm=perm, ids_shuffle=shuf)
    trb.flat.student.copy_(snap[0]); trb.flat.teacher.copy_(snap[1]); trb.flat.m.copy_(snap[2]); trb.flat.v.copy_(snap[3])
    trb.opt_step.copy_(snap[4]); trb.tick.copy_(snap[5]); trb.flat.step = 0
    trb.flat.grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    for (n, a), (_, b) in zip(sa.state_dict().items(), sb.state_dict().items()):
        assert torch.equal(a, b), n
    for (n, a), (_, b) in zip(ta.state_dict().items(), tb.state_dict().items()):
        assert torch.equal(a, b), ("teacher", n)


# ------------------------------------------------------------------------------------------------------------------------------
# the teacher's scorer inside the projection's epilogue (mhimx_proj_score) == projection + one-pass scorer + finalize
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,p,two_heads", [(10000, 0.0, True), (1000, 0.25, True), (330, 0.25, False), (160, 0.0, False)])
def test_scored_projection_equals_projection_plus_scorer(n, p, two_heads):
    from mhim_mil_amd import ops
    from mhim_mil_amd import _lib as L
    from tests.test_ops_gpu import rnd
    d, E, A, Cc = 1024, 512, 128, 2
    x = torch.from_numpy(synth.bag(5100, n, d)).to(DEV)
    w1 = (rnd(201, (E, d), std=0.03)).to(DEV); b1 = rnd(202, (E,), std=0.1).to(DEV)
    w1s = (rnd(203, (E, d), std=0.03)).to(DEV); b1s = rnd(204, (E,), std=0.1).to(DEV)
    wa = rnd(205, (A, E), std=0.05).to(DEV); wc = (rnd(206, (1, A), std=0.3) * 5).to(DEV)
    wp = rnd(207, (Cc, E), std=0.05).to(DEV); bp = rnd(208, (Cc,), std=0.1).to(DEV)
    act = L.ACT["gelu"]
    sact = L.ACT["relu"]
    tick = torch.zeros(1, dtype=torch.int64, device=DEV)

    def heads():
        hs = [ops.ProjHead(ops.pair_planes(w1), b1, drop_p=p, drop_seed=777)]
        if two_heads:
            hs.append(ops.ProjHead(ops.pair_planes(w1s), b1s, drop_p=p, drop_seed=778, want_dact=True))
        return hs
    # A: rows to HBM, then the one-pass scorer + finalize
    ha = heads()
    ops.bag_project(x, ha, act=act, drop_tick=tick)
    frag = torch.empty_like(wa)
    ops.prep_batch([(ops.PREP_FRAG, wa, frag)])
    sc = ops.ScorerW(wa, wc, sact, wa_frag=frag)
    st = ops.abmil_pool_fwd(sc, ha[0].out, None, wp=wp, bp=bp)
    # B: scored in the epilogue
    img = torch.empty(144 * E, device=DEV)
    ops.prep_batch([(ops.PREP_FRAG16, wa, img), (ops.PREP_FRAG16, wp, img[128 * E:])])
    for keep in (True, False):
        buf = ops.ProjScoreBuf(n, img, wc.view(-1).contiguous(), sact, Cc, x.device)
        hb = heads()
        ops.bag_project(x, hb, act=act, drop_tick=tick, score0=buf, keep_rows0=keep)
        buf.finalize(bp)
        torch.cuda.synchronize()
        if keep:
            assert torch.equal(hb[0].out, ha[0].out)                      # the same rows, bit for bit (same activation, same dropout stream)
        else:
            assert hb[0].out is None
        if two_heads:
            assert torch.equal(hb[1].out, ha[1].out) and torch.equal(hb[1].dact, ha[1].dact)
        sa, sb = st.s.double().cpu(), buf.s.double().cpu()
        scale = sa.abs().max().item()
        assert (sa - sb).abs().max().item() <= 3e-5 * scale, ((sa - sb).abs().max().item(), scale)
        ca, cb = st.cproj.double().cpu(), buf.cproj.double().cpu()
        assert (ca - cb).abs().max().item() <= 3e-5 * ca.abs().max().item()
        za, zb = st.z.double().cpu(), buf.z.double().cpu()
        assert (za - zb).abs().max().item() <= 2e-4 * za.abs().max().item(), (za - zb).abs().max().item()
        # the statistics against fp64 of the device's own scores
        mref = sb.max().item()
        lref = torch.exp(sb - mref).sum().item()
        assert abs(buf.stats[0].item() - mref) <= 1e-6 * max(1.0, abs(mref)) and abs(buf.stats[1].item() - lref) <= 2e-5 * lref
        pa, pb = st.pscore.double().cpu(), buf.pscore.double().cpu()
        assert (pa - pb).abs().max().item() <= 2e-5
    # replaying the launch: the pair counters only count up
    for _ in range(3):
        buf2 = ops.ProjScoreBuf(n, img, wc.view(-1).contiguous(), sact, Cc, x.device)
        ops.bag_project(x, heads(), act=act, drop_tick=tick, score0=buf2)
        buf2.finalize(bp)
    torch.cuda.synchronize()
    assert torch.equal(buf2.s, buf.s) and torch.equal(buf2.z, buf.z) and torch.equal(buf2.pscore, buf.pscore)


def test_common_mil_graph_cache_replays_the_native_step():
    """CommonMIL(args, fused=optimizer, graph_cache=K): the second bag of a shape is captured, later ones replay - the same numbers as the
    eager native path when that path is given the seeds the capture froze (the draw streams then advance through the device tick alone)."""
    import types
    from mhim_mil_amd.engine import CommonMIL
    from mhim_mil_amd.optim import FusedAdamEMA
    args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5, main_alpha=1.0)
    crit = torch.nn.CrossEntropyLoss()
    mm, steps = 0.999, 6
    bags = [torch.from_numpy(synth.bag(4100 + i, N, D)).to(DEV)[None] for i in range(steps)]
    other = torch.from_numpy(synth.bag(4200, N - 200, D)).to(DEV)[None]          # another shape in between: eager, its own cache entry later

    def run(cache):
        torch.manual_seed(11)
        s, t = _models()
        opt = FusedAdamEMA(s, t, lr=2e-4, weight_decay=1e-5, mm=mm)
        eng = CommonMIL(args, fused=opt, graph_cache=cache)
        out, frozen = [], None
        seq = [(bags[i], i % 2) for i in range(steps)]
        seq.insert(3, (other, 1))
        for step, (x, lab) in enumerate(seq):
            label = torch.tensor([lab], device=DEV)
            if not cache and x.shape[1] == N:
                if step == 1:
                    frozen = (s._step, t._step)                       # what the capture (second bag of the shape) bakes in
                if step >= 1:
                    s._step, t._step = frozen
            logits, lb, aux, pn, kn, _, _ = eng.forward_func(args, s, t, x, label, crit, 1, step, 0, step, None)
            loss = args.main_alpha * crit(logits.view(1, -1), lb) + args.aux_alpha * aux
            loss.backward()
            opt.step()
            opt.zero_grad()
            out.append((float(loss), int(pn), int(kn)))
        torch.cuda.synchronize()
        return out, {k: v.detach().clone() for k, v in s.state_dict().items()}, {k: v.detach().clone() for k, v in t.state_dict().items()}, eng

    a, sa, ta, eng = run(2)
    sg = eng.fused.trainer._shape_graphs
    assert len(sg["graphs"]) == 1 and sg["seen"][next(iter(sg["graphs"]))] == 2
    assert sg["arena"] is not None and sg["arena"].numel() == N * D               # (the graphs' shared bag buffer: the largest bag captured)
    b, sb, tb, _ = run(0)
    for (la, pa, ka), (lb_, pb, kb) in zip(a, b):
        assert pa == pb and ka == kb
        if pa == N:
            assert la == lb_, (la, lb_)
    # the other-shaped bag ran eagerly in both runs but with different host seeds (run b's counters were rewound): compare the bags up to it
    assert [x[0] for x in a[:3]] == [x[0] for x in b[:3]]


def test_common_mil_graph_cache_many_shapes_share_one_bag_buffer():
    """A dataset of bags of different sizes under graph_cache: every shape is captured at its second visit (one shared bag buffer: the graphs
    never run concurrently), replayed from the third on; the oldest shape leaves when the cache is full."""
    import types
    from mhim_mil_amd.engine import CommonMIL
    from mhim_mil_amd.optim import FusedAdamEMA
    args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5, main_alpha=1.0)
    crit = torch.nn.CrossEntropyLoss()
    s, t = _models()
    opt = FusedAdamEMA(s, t, lr=2e-4, weight_decay=1e-5, mm=0.999)
    eng = CommonMIL(args, fused=opt, graph_cache=2)
    sizes = [N, N - 160, N - 320]
    bags = [torch.from_numpy(synth.bag(4300 + i, n, D)).to(DEV)[None] for i, n in enumerate(sizes)]
    w0 = s.feature[0].weight.detach().clone()
    losses = []
    for epoch in range(3):
        for x in bags:
            label = torch.tensor([epoch % 2], device=DEV)
            logits, lb, aux, pn, kn, _, _ = eng.forward_func(args, s, t, x, label, crit, 1, epoch, 0, epoch, None)
            assert pn == x.shape[1]
            loss = args.main_alpha * crit(logits.view(1, -1), lb) + args.aux_alpha * aux
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    sg = opt.trainer._shape_graphs
    assert all(np.isfinite(losses)) and len(sg["graphs"]) == 2                    # three shapes, room for two
    assert sg["arena"].numel() == N * D and all(e[1].data_ptr() == sg["arena"].data_ptr() for e in sg["graphs"].values())
    assert float((s.feature[0].weight.detach() - w0).abs().max()) > 0


def test_trainer_shape_cached_train_step_equals_eager_steps_with_the_capture_seeds():
    """FusedTrainer.shape_cached('train_step'): complete steps (update and folded reductions included) replayed per bag shape."""
    from mhim_mil_amd.engine import FusedTrainer
    bags = [torch.from_numpy(synth.bag(4400 + i, N, D)).to(DEV) for i in range(5)]

    def run(cached):
        torch.manual_seed(13)
        s, t = _models()
        tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999)
        out, frozen = [], None
        for step, x in enumerate(bags):
            label = torch.tensor([step % 2], device=DEV)
            if cached:
                lg, ls, pn, kn = tr.shape_cached("train_step", x, label, cache=4)
            else:
                if step == 1:
                    frozen = (s._step, t._step)
                if step >= 1:
                    s._step, t._step = frozen
                lg, ls = tr.train_step(x, label)
            out.append((lg.cpu().clone(), ls.cpu().clone()))
        torch.cuda.synchronize()
        return out, {k: v.detach().cpu().clone() for k, v in s.state_dict().items()}, tr

    a, sa, tra = run(True)
    b, sb, trb = run(False)
    assert tra.flat.step == trb.flat.step == len(bags)
    for (la, sa_), (lb, sb_) in zip(a, b):
        assert torch.equal(la, lb) and torch.equal(sa_, sb_)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_shape_cached_follows_a_ham_ratio_schedule():
    """--mrh_sche: the row counts change when the scheduled ratio crosses a rounding boundary; they are part of the cache key, so every
    plateau of the schedule gets its own graph and the masked / kept counts follow the schedule."""
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models()
    s.mrh_sche = [0.03, 0.03, 0.03, 0.02, 0.02, 0.02, 0.0295]
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999)
    keeps = []
    for step in range(len(s.mrh_sche)):
        x = torch.from_numpy(synth.bag(4500 + step, N, D)).to(DEV)
        out = tr.shape_cached("train_step", x, torch.tensor([step % 2], device=DEV), i=step, cache=4)
        assert out is not None and bool(torch.isfinite(out[1]).all())
        keeps.append(out[3])
        assert out[3] == s.v2_counts(N, step)[3] + s.merge.k
    torch.cuda.synchronize()
    assert keeps[0] == keeps[2] and keeps[3] == keeps[5] and keeps[0] != keeps[3]
    assert len(tr._shape_graphs["graphs"]) == 2                     # one per plateau seen twice (the last ratio was seen once: eager)
    assert tr.flat.step == len(s.mrh_sche)
