"""Accumulation windows (--accumulation_steps k: base_engine.py:29,47-49,100-119,146-167) on the fused trainer — GPU box only.

(1) the sequential form (k train_step calls, one update) against the REFERENCE fixture g18 (accumulation 8, two updates);
(2) the batched form (FusedTrainer.window_step: one preparation, the k bags issued over several HIP streams, one slab sum, one Adam + EMA)
    against the oracle's window step on the row sets and teacher scores the device drew — per-bag logits, the accumulated gradients
    before the update, parameters and global queries after it; one stream == several streams on identical draws;
(3) the captured window (ONE hipGraph with the streams as branches), replayed, against the oracle.
"""
import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G
from tests.test_single_pass_gpu import _draws_from_device

pytestmark = pytest.mark.gpu
DEV = "cuda"
V2 = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
          merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)


def _mk(sd, d, **cfg):
    from mhim_mil_amd.mhim import MHIM
    m = MHIM(input_dim=d, n_classes=2, baseline="attn", **cfg)
    sd = dict(sd)
    sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    m = m.to(DEV).train()
    m.merge.dropout = 0.0
    return m


def _check_params(mdl, ref, mean_tol, max_tol, what):
    sd = mdl.state_dict()
    for name, r in ref.items():
        if name == "merge.global_q":
            continue
        err = (sd[name].detach().cpu().double() - r.double()).abs()
        assert err.mean().item() <= mean_tol and err.max().item() <= max_tol, (what, name, err.mean().item(), err.max().item())


def test_sequential_accumulation_vs_reference_fixture():
    """FusedTrainer(accumulation_steps=8) stepping bag after bag == the reference modules with --accumulation_steps 8 (fixture g18 from the
    reference import: 16 bags, two optimiser updates, injected draws; Merge's query EMA runs bag after bag as in the reference)."""
    from mhim_mil_amd.engine import FusedTrainer
    meta, a = G.load("g18_train_accum8")
    d, n, acc = meta["d"], meta["n"], meta["accum"]
    base = synth.mhim_state(meta["seed"], input_dim=d, merge_k=meta["merge_k"])
    cfg = {k: meta[k] for k in V2}
    s, t = _mk(base, d, **cfg), _mk(synth.spread_teacher(base), d, **cfg)
    tr = FusedTrainer(s, t, lr=meta["lr"], weight_decay=meta["wd"], mm=meta["mm"], aux_alpha=meta["aux_alpha"], accumulation_steps=acc)
    for b in range(meta["updates"] * acc):
        x = torch.from_numpy(synth.bag(int(a["xseeds"][b]), n, d)).to(DEV)
        logits, losses = tr.train_step(x[None], torch.tensor([b % 2], device=DEV), perm=torch.from_numpy(a[f"perm{b}"]).to(DEV),
                                       ids_shuffle=torch.from_numpy(a[f"shuf{b}"]).to(DEV))
        np.testing.assert_allclose(logits.cpu().numpy().ravel(), a["logits"][b].ravel(), atol=1e-4, rtol=0)
        assert abs(float(losses[0]) - float(a["losses"][b])) < 3e-4, (b, float(losses[0]), a["losses"][b])
    assert tr.flat.step == meta["updates"] and tr._micro == 0
    for tag, mdl in (("stu", s), ("tea", t)):
        sd = mdl.state_dict()
        for k, exp in G.tagged(a, tag).items():
            got = sd[k].detach().cpu().numpy().astype(np.float64).reshape(-1)
            want = exp["full"].astype(np.float64).reshape(-1) if "full" in exp else exp["sample"].astype(np.float64)
            got = got if "full" in exp else got[::int(exp["stride"])][:want.shape[0]]
            err = np.abs(got - want)           # (Adam: a rounding-level difference on a near-zero gradient moves that element by ~lr)
            assert err.mean() <= 5e-6 and err.max() <= 4.1e-4, (tag, k, err.mean(), err.max())


def _window_vs_oracle(tr, s, t, bags, labels, stu, tea, opt, ocfg, step, n, k, n_sel, run):
    """run() executes one window on the device; the draws it made are read back and handed to the oracle."""
    run()
    torch.cuda.synchronize()
    per = tr.last["bags"]
    perms, shufs, scores = [], [], []
    for j in range(len(bags)):
        rows, score = per[j]["rows"].cpu().numpy(), per[j]["score"].cpu().numpy()
        p, sh = _draws_from_device(score, rows, per[j]["R"], n, k, n_sel)
        perms.append(p); shufs.append(sh); scores.append(torch.from_numpy(score))
    stu2, tea2, opt2, info = O.train_window(bags, labels, stu, tea, opt, ocfg, step, perms=perms, shuffles=shufs, q_ema="window",
                                            score_overrides=scores, mm=tr.mm)
    for j in range(len(bags)):
        np.testing.assert_allclose(per[j]["logits"].cpu().numpy().ravel(), info["logits"][j].numpy().ravel(), atol=1e-4, rtol=0)
        assert abs(float(per[j]["losses"][0]) - info["loss"][j]) < 3e-4
    return stu2, tea2, opt2, info


@pytest.fixture(params=["per-bag GEMM launches", "one launch per GEMM", "every launch over all bags"])
def window_gemms(request):
    """The window's bags on HIP streams with the projections / weight-gradient products per bag or as ONE launch each
    (mhimx_bag_project_multi / mhimx_bag_wgrad_multi; opt-in: MHIMX_WINDOW_PROJECT / MHIMX_WINDOW_WGRAD) - or (round 6, the default for
    same-shaped bags) the whole window as ONE call of mhimx_window_run: every launch covers all the bags."""
    from mhim_mil_amd import engine as EN
    old = EN._WINDOW_PROJECT, EN._WINDOW_WGRAD, EN._WINDOW_BATCHED
    EN._WINDOW_PROJECT = EN._WINDOW_WGRAD = request.param == "one launch per GEMM"
    EN._WINDOW_BATCHED = request.param == "every launch over all bags"
    yield request.param
    EN._WINDOW_PROJECT, EN._WINDOW_WGRAD, EN._WINDOW_BATCHED = old


@pytest.mark.parametrize("n_streams", [1, 3])
def test_window_step_vs_oracle(n_streams, window_gemms):
    from mhim_mil_amd.engine import FusedTrainer
    n, d, acc = 2100, 256, 4
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)
    s, t = _mk(base, d, **V2), _mk(tsd, d, **V2)
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997, accumulation_steps=acc)
    ocfg = O.Cfg(**V2)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    stu, tea, opt = O.as_torch(base), O.as_torch(tsd), {}
    for u in range(2):
        bags = [torch.from_numpy(synth.bag(500 + 10 * u + j, n, d)) for j in range(acc)]
        labels = [(u + j) % 2 for j in range(acc)]
        xs = [b.to(DEV)[None] for b in bags]
        ls = [torch.tensor([l], device=DEV) for l in labels]
        stu, tea, opt, info = _window_vs_oracle(tr, s, t, bags, labels, stu, tea, opt, ocfg, u + 1, n, k, n_sel,
                                                lambda: tr.window_step(xs, ls, n_streams=n_streams, update=False))
        gv = tr.flat.grad_views
        for name, ref in info["grads"].items():                          # the accumulated gradient of the window, before the update
            g, r = gv[name].cpu().numpy(), ref.numpy()
            np.testing.assert_allclose(g, r.reshape(g.shape), atol=2e-3 * (np.abs(r).max() + 1e-30), rtol=2e-3, err_msg=name)
        tr.update()
        torch.cuda.synchronize()
        _check_params(s, stu, 3e-6, 4.1e-4 * (u + 1), f"student, window {u}")
        _check_params(t, tea, 1e-6, 2e-6, f"teacher, window {u}")
        np.testing.assert_allclose(s.merge.global_q_mm.detach().cpu().numpy(), stu["merge.global_q_mm"].numpy(), atol=2e-6, rtol=0)
        # keep the two sides on the oracle's trajectory (Adam's first steps amplify rounding on a few elements)
        s.load_state_dict({**stu, "merge.global_q": stu["merge.global_q_mm"]})
        t.load_state_dict({**tea, "merge.global_q": tea["merge.global_q_mm"]})


def test_batched_window_at_baseline_size_vs_oracle():
    """BASELINE c2's shape under --accumulation_steps 8: eight bags of N = 10 000, D = 1024 through mhimx_window_run (captured, then replayed
    from the restored state) against the oracle's window step on the draws the device made: per-bag logits 1e-4, the parameters after the
    update, the chained global queries."""
    from mhim_mil_amd import engine as EN
    from mhim_mil_amd.engine import FusedTrainer
    if not EN._WINDOW_BATCHED:
        pytest.skip("MHIMX_WINDOW_BATCHED=0: the stream form is selected")
    n, d, acc = 10000, 1024, 8
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)
    s, t = _mk(base, d, **V2), _mk(tsd, d, **V2)
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997, accumulation_steps=acc)
    bags = [torch.from_numpy(synth.bag(1300 + j, n, d)) for j in range(acc)]
    labels = [j % 2 for j in range(acc)]
    xs = [b.to(DEV)[None] for b in bags]
    ls = [torch.tensor([l], device=DEV) for l in labels]
    assert tr._exec_window_ok([x[0] for x in xs], ls), "the window did not take the batched form"
    snap = [tr.flat.student.clone(), tr.flat.teacher.clone(), tr.flat.m.clone(), tr.flat.v.clone(), tr.opt_step.clone(), tr.tick.clone(), tr.flat.step]
    g = tr.capture_window(xs, ls, warmup=1)
    tr.flat.student.copy_(snap[0]); tr.flat.teacher.copy_(snap[1]); tr.flat.m.copy_(snap[2]); tr.flat.v.copy_(snap[3])
    tr.opt_step.copy_(snap[4]); tr.tick.copy_(snap[5]); tr.flat.step = snap[6]
    tr.flat.grad.zero_()
    ocfg = O.Cfg(**V2)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    stu, tea, opt, _ = _window_vs_oracle(tr, s, t, bags, labels, O.as_torch(base), O.as_torch(tsd), {}, ocfg, 1, n, k, n_sel, g.replay)
    _check_params(s, stu, 3e-6, 4.1e-4, "student after the replayed window")
    _check_params(t, tea, 1e-6, 2e-6, "teacher after the replayed window")
    np.testing.assert_allclose(s.merge.global_q_mm.detach().cpu().numpy(), stu["merge.global_q_mm"].numpy(), atol=2e-6, rtol=0)


def test_window_streams_do_not_change_the_result():
    """The same window on 1 stream and on 4: identical draws (the seeds do not depend on the stream), bit-identical per-bag logits, the
    accumulated gradient equal up to the order of the slab sum."""
    from mhim_mil_amd.engine import FusedTrainer
    n, d, acc = 1500, 256, 8
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    cfg = dict(V2, dropout=0.25)
    xs = [torch.from_numpy(synth.bag(900 + j, n, d)).to(DEV)[None] for j in range(acc)]
    ls = [torch.tensor([j % 2], device=DEV) for j in range(acc)]
    res = []
    for S in (1, 4):
        torch.manual_seed(5)
        s, t = _mk(base, d, **cfg), _mk(synth.spread_teacher(base), d, **cfg)
        tr = FusedTrainer(s, t, accumulation_steps=acc)
        logits, _ = tr.window_step(xs, ls, n_streams=S, update=False)
        torch.cuda.synchronize()
        res.append((torch.stack(logits).cpu(), [b["rows"].cpu() for b in tr.last["bags"]], tr.flat.grad.cpu().clone(),
                    s.merge.global_q_mm.detach().cpu().clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    g0, g1 = res[0][2].numpy(), res[1][2].numpy()
    np.testing.assert_allclose(g0, g1, atol=2e-6 * np.abs(g0).max(), rtol=1e-4)
    assert torch.equal(res[0][3], res[1][3])


def test_batched_window_equals_the_stream_window():
    """mhimx_window_run (every launch over all the bags, round 6) against the same window on HIP streams with per-bag launches of the same
    kernels: the two forms draw the same seeds, so per-bag scores, row lists and logits are bit-identical; the accumulated gradient and the
    updated parameters agree up to the slab counts that follow a launch's size; dropout 0.25, 8 bags, then a second window."""
    from mhim_mil_amd import engine as EN
    from mhim_mil_amd.engine import FusedTrainer
    n, d, acc = 1500, 256, 8
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    cfg = dict(V2, dropout=0.25)
    xs = [torch.from_numpy(synth.bag(900 + j, n, d)).to(DEV)[None] for j in range(acc)]
    ls = [torch.tensor([j % 2], device=DEV) for j in range(acc)]
    old = EN._WINDOW_PROJECT, EN._WINDOW_BATCHED
    res = []
    try:
        for batched in (False, True):
            EN._WINDOW_PROJECT, EN._WINDOW_BATCHED = True, batched
            torch.manual_seed(5)
            s, t = _mk(base, d, **cfg), _mk(synth.spread_teacher(base), d, **cfg)
            tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997, accumulation_steps=acc)
            assert tr._exec_window_ok([x[0] for x in xs], ls) == batched
            out = []
            for w in range(2):
                logits, losses = tr.window_step(xs, ls, n_streams=2, update=False)
                torch.cuda.synchronize()
                per = tr.last["bags"]
                out.append((torch.stack([l.reshape(-1) for l in logits]).cpu().clone(), torch.stack([l.reshape(-1)[:3] for l in losses]).cpu().clone(),
                            [b["rows"].cpu().clone() for b in per], [b["score"].cpu().clone() for b in per], [b["tokens"].cpu().clone() for b in per],
                            tr.flat.grad.cpu().clone(), s.merge.global_q_mm.detach().cpu().clone()))
                tr.update()
                torch.cuda.synchronize()
                out.append(tr.flat.student.cpu().clone())
            res.append(out)
    finally:
        EN._WINDOW_PROJECT, EN._WINDOW_BATCHED = old
    a, b = res
    for w in (0, 2):
        if w == 0:                                                         # (the second window starts from parameters that differ in the last bits)
            assert torch.equal(a[w][0], b[w][0]) and torch.equal(a[w][1], b[w][1])
            assert all(torch.equal(x, y) for x, y in zip(a[w][2], b[w][2]))
            assert all(torch.equal(x, y) for x, y in zip(a[w][3], b[w][3]))
            assert all(torch.equal(x, y) for x, y in zip(a[w][4], b[w][4]))
        else:
            np.testing.assert_allclose(a[w][0].numpy(), b[w][0].numpy(), atol=2e-5, rtol=0)
        g0, g1 = a[w][5].numpy(), b[w][5].numpy()
        np.testing.assert_allclose(g0, g1, atol=3e-6 * np.abs(g0).max(), rtol=1e-4)
        np.testing.assert_allclose(a[w][6].numpy(), b[w][6].numpy(), atol=1e-6, rtol=0)
        p0, p1 = a[w + 1].numpy(), b[w + 1].numpy()
        assert np.abs(p0 - p1).mean() < 2e-6, np.abs(p0 - p1).mean()


def test_projection_and_weight_gradient_of_several_bags_in_one_launch():
    """ops.bag_project_multi / ops.bag_wgrad_multi against per-bag ops.bag_project / ops.bag_wgrad with the same seeds: bit-identical
    feature rows and d out / d pre (same tiles, same hash stream per bag), the summed weight gradient up to the slab-sum order."""
    from mhim_mil_amd import ops
    n, d, E, nb = 1300, 256, 512, 3
    g = torch.Generator(device=DEV).manual_seed(11)
    xs = [torch.randn(n, d, device=DEV, generator=g).abs_() for _ in range(nb)]
    wt, ws_ = (torch.randn(E, d, device=DEV, generator=g) * 0.05 for _ in range(2))
    bt, bs = (torch.randn(E, device=DEV, generator=g) * 0.1 for _ in range(2))
    wtp, wsp = ops.pair_planes(wt), ops.pair_planes(ws_)
    tick = torch.tensor([3], dtype=torch.int64, device=DEV)
    mk = lambda b: [ops.ProjHead(wtp, bt, drop_p=0.25, drop_seed=100 + 2 * b), ops.ProjHead(wsp, bs, drop_p=0.25, drop_seed=101 + 2 * b, want_dact=True)]
    one = [ops.bag_project(x, mk(b), act=2, drop_tick=tick) for b, x in enumerate(xs)]
    many = ops.bag_project_multi(xs, [mk(b) for b in range(nb)], act=2, drop_tick=tick)
    for a, m in zip(one, many):
        assert torch.equal(a[0].out, m[0].out) and torch.equal(a[1].out, m[1].out) and torch.equal(a[1].dact, m[1].dact)
    assert not torch.equal(many[0][1].out != 0, many[1][1].out != 0)          # every bag its own mask
    rows = [torch.randperm(n, device=DEV, generator=g)[:1100].contiguous() for _ in range(nb)]
    dHs = [torch.randn(n, E, device=DEV, generator=g) * 0.01 for _ in range(nb)]
    w0, b0 = torch.empty(E, d, device=DEV), torch.empty(E, device=DEV)
    for b in range(nb):
        ops.bag_wgrad(dHs[b], many[b][1].dact, xs[b], rows[b], 1100, out_w=w0, out_b=b0, accumulate=b > 0)
    w1, b1 = torch.empty(E, d, device=DEV), torch.empty(E, device=DEV)
    ims = [ops.bag_wgrad_image(dHs[b], many[b][1].dact, xs[b], rows[b], 1100, out_b=b1, accumulate=b > 0) for b in range(nb)]
    ops.bag_wgrad_multi(ims, w1)
    torch.cuda.synchronize()
    assert torch.equal(b0, b1)
    ref = sum((dHs[b][rows[b]].double() * many[b][1].dact[rows[b]].double()).t() @ xs[b][rows[b]].double() for b in range(nb))
    rel = lambda a: float((a.double() - ref).abs().max() / ref.abs().max())
    assert rel(w0) < 2e-5 and rel(w1) < 2e-5


def test_captured_window_replays_vs_oracle():
    """capture_window: the whole window (prep, 8 bags on 4 streams, slab sum, query chain, Adam + EMA) as ONE hipGraph; a replay from a
    restored state against the oracle's window step, then a second replay advances the state again (fresh draws: the device tick)."""
    from mhim_mil_amd.engine import FusedTrainer
    n, d, acc = 2100, 256, 8
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)
    s, t = _mk(base, d, **V2), _mk(tsd, d, **V2)
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997, accumulation_steps=acc)
    bags = [torch.from_numpy(synth.bag(700 + j, n, d)) for j in range(acc)]
    labels = [j % 2 for j in range(acc)]
    xs = [b.to(DEV)[None] for b in bags]
    ls = [torch.tensor([l], device=DEV) for l in labels]
    snap = [tr.flat.student.clone(), tr.flat.teacher.clone(), tr.flat.m.clone(), tr.flat.v.clone(), tr.opt_step.clone(), tr.tick.clone(), tr.flat.step]
    g = tr.capture_window(xs, ls, warmup=1, n_streams=4)
    tr.flat.student.copy_(snap[0]); tr.flat.teacher.copy_(snap[1]); tr.flat.m.copy_(snap[2]); tr.flat.v.copy_(snap[3])
    tr.opt_step.copy_(snap[4]); tr.tick.copy_(snap[5]); tr.flat.step = snap[6]
    tr.flat.grad.zero_()
    ocfg = O.Cfg(**V2)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    stu, tea, opt, _ = _window_vs_oracle(tr, s, t, bags, labels, O.as_torch(base), O.as_torch(tsd), {}, ocfg, 1, n, k, n_sel, g.replay)
    _check_params(s, stu, 3e-6, 4.1e-4, "student after the replayed window")
    _check_params(t, tea, 1e-6, 2e-6, "teacher after the replayed window")
    rows0 = [b["rows"].clone() for b in tr.last["bags"]]
    w0 = s.feature[0].weight.detach().clone()
    g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(w0, s.feature[0].weight.detach()) and torch.isfinite(tr.flat.student).all()
    assert any(not torch.equal(a, b["rows"]) for a, b in zip(rows0, tr.last["bags"])), "the second replay drew the same random subsets"


def test_clip_grad_and_lr_schedule_on_the_device():
    """--clip_grad (base_engine.py:115-119) and a per-update learning-rate schedule (train_utils.py:69-77, base_engine.py:152-153) inside
    the fused optimiser launch (mhimx_optim_step): three eager steps and two replays of ONE captured graph against the oracle stepped
    with the same clip value and the schedule's entries - the table is read at the DEVICE step counter, so it advances under replay."""
    from mhim_mil_amd.engine import FusedTrainer
    n, d = 1200, 256
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)
    lrs = [2e-4, 1.5e-4, 1e-4, 5e-5, 2.5e-5, 1e-5]
    ocfg = O.Cfg(**V2)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    for clip in (0.05, None):
        s, t = _mk(base, d, **V2), _mk(tsd, d, **V2)
        tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997, clip_grad=clip, lr_sche=lrs)
        stu, tea, opt = O.as_torch(base), O.as_torch(tsd), {}
        x = torch.from_numpy(synth.bag(321, n, d))
        xd, lab = x.to(DEV)[None], torch.tensor([1], device=DEV)
        graph = None
        for step in range(5):
            if step < 3:
                tr.forward_backward(xd, lab)
                torch.cuda.synchronize()
                if step == 0 and clip is not None:                      # the clip must actually bite in this test
                    gn = float(tr.flat.grad[:tr.flat.n_train].double().norm())
                    assert gn > 2 * clip, gn
                tr.update()
            else:
                if graph is None:
                    snap = [tr.flat.student.clone(), tr.flat.teacher.clone(), tr.flat.m.clone(), tr.flat.v.clone(), tr.opt_step.clone(),
                            tr.tick.clone(), tr.flat.step]
                    graph = tr.capture(xd, lab, warmup=1)
                    tr.flat.student.copy_(snap[0]); tr.flat.teacher.copy_(snap[1]); tr.flat.m.copy_(snap[2]); tr.flat.v.copy_(snap[3])
                    tr.opt_step.copy_(snap[4]); tr.tick.copy_(snap[5]); tr.flat.step = snap[6]
                    tr.flat.grad.zero_()
                graph.replay()
            torch.cuda.synchronize()
            rows, score = tr.last["rows"].cpu().numpy(), tr.last["score"].cpu().numpy()
            perm, shuf = _draws_from_device(score, rows, tr.last["R"], n, k, n_sel)
            stu, tea, opt, info = O.train_step(x, 1, stu, tea, opt, ocfg, step + 1, perm=perm, ids_shuffle=shuf, lr=lrs[step],
                                               score_override=torch.from_numpy(score), clip_grad=clip)
            # the update of this step has the schedule's size: |dw| <= lr_step (+ eps effects), so a frozen lr (2e-4) would overshoot
            _check_params(s, stu, 3e-6, 2.05 * lrs[step] + 2e-5, f"clip={clip} step {step}")
            s.load_state_dict({**stu, "merge.global_q": stu["merge.global_q_mm"]})
            t.load_state_dict({**tea, "merge.global_q": tea["merge.global_q_mm"]})
            # the Adam moments live on the device: bring them onto the oracle's trajectory too
            for name in tr.flat.train_names:
                o, nn_ = tr.flat.offsets[name], opt[name][0].numel()
                tr.flat.m[o:o + nn_].copy_(opt[name][0].reshape(-1).to(DEV))
                tr.flat.v[o:o + nn_].copy_(opt[name][1].reshape(-1).to(DEV))


def test_several_bags_entries_reject_mixed_shapes_and_take_one_bag():
    """mhimx_bag_wgrad_multi with ONE bag is mhimx_bag_wgrad; bags of different L (or E, D, ldx) are refused, so are more than 8 bags;
    mhimx_bag_project_multi refuses bags that do not share the weight image."""
    from mhim_mil_amd import ops, _lib as L
    import ctypes as C
    n, d, E = 700, 256, 512
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(n, d, device=DEV, generator=g).abs_()
    dH = torch.randn(n, E, device=DEV, generator=g) * 0.01
    dact = (torch.rand(n, E, device=DEV, generator=g) * 1.2).half()
    rows = torch.randperm(n, device=DEV, generator=g)[:600].contiguous()
    w0, _ = ops.bag_wgrad(dH, dact, x, rows, 600)
    im = ops.bag_wgrad_image(dH, dact, x, rows, 600)
    w1 = ops.bag_wgrad_multi([im], torch.empty(E, d, device=DEV))
    torch.cuda.synchronize()
    assert torch.equal(w0, w1)
    im2 = ops.bag_wgrad_image(dH, dact, x, rows[:500].contiguous(), 500)
    with pytest.raises(L.MhimxError):
        ops.bag_wgrad_multi([im, im2], torch.empty(E, d, device=DEV))
    with pytest.raises(L.MhimxError):
        ops.bag_wgrad_multi([im] * 9, torch.empty(E, d, device=DEV))
    wa, wb = (ops.pair_planes(torch.randn(E, d, device=DEV, generator=g) * 0.05) for _ in range(2))
    with pytest.raises(L.MhimxError):
        ops.bag_project_multi([x, x], [[ops.ProjHead(wa)], [ops.ProjHead(wb)]], act=0)
