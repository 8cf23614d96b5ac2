"""GPU tests of the single-pass ABMIL step: mhimx_bag_project (teacher + student projection in one launch over the raw bag) and
the row-gather forms of the pool / Merge / activation-backward kernels that read its bag-ordered buffers."""
import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from mhim_mil_amd import ops
    return ops


def rnd(seed, shape, std=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * std).astype(np.float32))


@pytest.mark.parametrize("N,D,heads", [(1, 32, 2), (79, 64, 2), (160, 1024, 1), (801, 128, 2), (10000, 1024, 2), (3333, 1536, 2)])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_bag_project_vs_fp64(N, D, heads, act):
    """H_g = act(X W_g^T + b_g) for both models from one launch, 3-term bf16 (~2^-16), and d out / d pre (fp16)."""
    ops = _ops()
    E = 512
    x = rnd(1, (N, D)).abs()
    ws = [rnd(2 + g, (E, D), std=(2.0 / (E + D)) ** 0.5) for g in range(heads)]
    bs = [rnd(7 + g, (E,), std=0.1) for g in range(heads)]
    hs = [ops.ProjHead(ops.pair_planes(w.to(DEV)), b.to(DEV), want_dact=(g == heads - 1)) for g, (w, b) in enumerate(zip(ws, bs))]
    ops.bag_project(x.to(DEV), hs, act={"gelu": 2, "relu": 1}[act], extra_rows=3)
    for h, w, b in zip(hs, ws, bs):
        assert h.out.shape == (N + 3, E)
        pre = x.double() @ w.double().t() + b.double()
        ref = O._act(pre, act)
        scale = max(1.0, pre.abs().max().item())
        np.testing.assert_allclose(h.out[:N].cpu().numpy(), ref.float().numpy(), atol=2e-5 * scale, rtol=2e-5)
    pre = x.double() @ ws[-1].double().t() + bs[-1].double()
    if act == "gelu":
        gref = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * np.pi) ** 0.5
    else:
        gref = (pre > 0).double()
    got = hs[-1].dact.double().cpu()
    near0 = pre.abs() < 1e-4 if act == "relu" else torch.zeros_like(pre, dtype=torch.bool)       # relu'(0) is a convention
    assert ((got - gref).abs() <= 1e-3)[~near0].all()


def test_bag_project_dropout_mask_and_hash():
    ops = _ops()
    N, D, E = 700, 64, 512
    x, w, b = rnd(1, (N, D)).abs(), rnd(2, (E, D), std=0.1), rnd(3, (E,), std=0.1)
    wp = ops.pair_planes(w.to(DEV))
    xd = x.to(DEV)
    full = ops.bag_project(xd, [ops.ProjHead(wp, b.to(DEV), want_dact=True)], act=2)[0]
    mask = (torch.from_numpy(synth.uniform(5, (N, E))) >= 0.25).to(torch.uint8)
    m = ops.bag_project(xd, [ops.ProjHead(wp, b.to(DEV), drop_p=0.25, drop_mask=mask.to(DEV), want_dact=True)], act=2)[0]
    keep = mask.bool()
    np.testing.assert_allclose(m.out.cpu()[keep].numpy(), (full.out.cpu()[keep] / 0.75).numpy(), rtol=1e-6)
    assert (m.out.cpu()[~keep] == 0).all() and (m.dact.cpu()[~keep] == 0).all()
    np.testing.assert_allclose(m.dact.float().cpu()[keep].numpy(), (full.dact.float().cpu()[keep] / 0.75).numpy(), rtol=2e-3, atol=1e-6)
    # hashed stream: deterministic in (seed, row, column), keep rate 1 - p, the two models of one launch draw different masks
    tick = torch.zeros(1, dtype=torch.int64, device=DEV)
    def run(seed_a, seed_b):
        hs = [ops.ProjHead(wp, b.to(DEV), drop_p=0.25, drop_seed=seed_a), ops.ProjHead(wp, b.to(DEV), drop_p=0.25, drop_seed=seed_b)]
        ops.bag_project(xd, hs, act=2, drop_tick=tick)
        return hs[0].out.cpu(), hs[1].out.cpu()
    a1, b1 = run(77, 78)
    a2, b2 = run(77, 78)
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a1 != 0, b1 != 0)
    tick += 1
    a3, _ = run(77, 78)
    assert not torch.equal(a1 != 0, a3 != 0)                   # the device step counter moves the stream (graph replays)
    kept = (a1 != 0) | (full.out.cpu() == 0)
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    assert (kept.float().mean(0) - 0.75).abs().max() < 0.08 and (kept.float().mean(1) - 0.75).abs().max() < 0.1
    np.testing.assert_allclose(a1[a1 != 0].numpy(), (full.out.cpu()[a1 != 0] / 0.75).numpy(), rtol=1e-6)


def test_bag_project_argument_errors():
    ops = _ops()
    from mhim_mil_amd import _lib as L
    x = torch.zeros(100, 48, device=DEV)                        # D % 32 != 0
    wp = torch.zeros(512, 48, device=DEV)
    with pytest.raises(L.MhimxError):
        ops.bag_project(x, [ops.ProjHead(wp)])
    with pytest.raises(L.MhimxError):
        ops.bag_project(torch.zeros(100, 64, device=DEV), [ops.ProjHead(torch.zeros(384, 64, device=DEV))])   # E % 256 != 0


def test_pool_with_gathered_rows_matches_contiguous():
    """abmil_pool_fwd / bwd with rows1: same scores, pool and gradients as on a gathered copy; gradients land at rows1."""
    ops = _ops()
    from mhim_mil_amd import _lib as L
    M, E, A, n = 1500, 512, 128, 1100
    T = rnd(1, (M, E)).to(DEV)
    rows = torch.from_numpy(synth.permutation(2, M)[:n].copy()).to(DEV)
    wa, wc = rnd(3, (A, E), std=0.05).to(DEV), rnd(4, (1, A), std=0.3).to(DEV)
    sc = ops.ScorerW(wa, wc, L.ACT["relu"], prec="bf16x3")
    st_g = ops.abmil_pool_fwd(sc, T, None, rows1=rows)
    st_c = ops.abmil_pool_fwd(sc, T[rows].contiguous(), None)
    assert torch.equal(st_g.s, st_c.s) and torch.equal(st_g.z, st_c.z) and torch.equal(st_g.stats, st_c.stats)
    g_z = rnd(5, (E,)).to(DEV)
    wa_t = ops.transpose(wa)
    dT = torch.zeros(M, E, device=DEV)
    gg = ops.abmil_pool_bwd(sc, st_g, g_z, wa_t, grads={"dT1": dT})
    gc = ops.abmil_pool_bwd(sc, st_c, g_z, wa_t)
    assert torch.equal(dT[rows], gc["dT1"])
    untouched = torch.ones(M, dtype=torch.bool, device=DEV)
    untouched[rows] = False
    assert (dT[untouched] == 0).all()
    np.testing.assert_allclose(gg["d_wa"].cpu().numpy(), gc["d_wa"].cpu().numpy(), rtol=1e-5, atol=1e-6 * gc["d_wa"].abs().max().item())
    assert torch.equal(gg["d_wc"], gc["d_wc"])


@pytest.mark.parametrize("n,n_img", [(1100, 1095), (1100, 1100), (64, 59), (37, 37)])
def test_pool_backward_writes_its_rows_of_the_dpre_image(n, n_img):
    """Round 6 (mhimx_pool_grad.img): the one-pass pool backward writes the first img_rows tokens' share of the projection's dPRE image
    itself.  Against the two-pass route on the same inputs - dT rows to memory, then mhimx_rows_dpre_image over a list in which the rows
    behind img_rows are masked out: the image bytes are IDENTICAL (the same fp32 product split into the same bf16 hi / lo), the tokens
    behind img_rows keep their dT rows bit for bit, the per-tile column sums are the bias gradient."""
    ops = _ops()
    from mhim_mil_amd import _lib as L
    M, E, A = 1500, 512, 128
    T = rnd(1, (M, E)).to(DEV)
    rows = torch.from_numpy(synth.permutation(2, M)[:n].copy()).to(DEV)
    wa, wc = rnd(3, (A, E), std=0.05).to(DEV), rnd(4, (1, A), std=0.3).to(DEV)
    sc = ops.ScorerW(wa, wc, L.ACT["relu"], prec="bf16x3")
    st = ops.abmil_pool_fwd(sc, T, None, rows1=rows)
    g_z = rnd(5, (E,)).to(DEV)
    wa_t = ops.transpose(wa)
    dact = (rnd(6, (M, E)).to(DEV) * 0.7).half()
    # reference route: fp32 gradient rows, then the image pass
    dT0 = torch.zeros(M, E, device=DEV)
    ops.abmil_pool_bwd(sc, st, g_z, wa_t, grads={"dT1": dT0})
    tiles = -(-n // 32)
    keep = torch.zeros(tiles * 32, dtype=torch.uint8, device=DEV)
    keep[:n_img] = 1
    rows_pad = torch.zeros(tiles * 32, dtype=torch.int64, device=DEV)
    rows_pad[:n] = rows
    lib = L.lib()
    img0 = torch.zeros(lib.mhimx_wgrad_image_bytes(tiles * 32, E) // 4, device=DEV)
    b0, ws0 = torch.empty(E, device=DEV), torch.empty(tiles * E, device=DEV)
    dHg = torch.zeros(tiles * 32, E, device=DEV)
    dHg[:n] = dT0[rows]
    dact_g = torch.zeros(tiles * 32, E, device=DEV, dtype=torch.float16)
    dact_g[:n] = dact[rows]
    L.check(lib.mhimx_rows_dpre_image_k(ops._stream(), dHg.data_ptr(), dact_g.data_ptr(), keep.data_ptr(), tiles * 32, E, img0.data_ptr(), b0.data_ptr(), 0,
                                        ws0.data_ptr(), ws0.numel() * 4, None), "mhimx_rows_dpre_image_k")
    # fused route
    dT1 = torch.zeros(M, E, device=DEV)
    img1 = torch.full_like(img0, float("nan"))
    part = torch.full((tiles, E), float("nan"), device=DEV)
    ops.abmil_pool_bwd(sc, st, g_z, wa_t, grads={"dT1": dT1}, img=img1, img_dact=dact, img_part=part, img_rows=n_img)
    torch.cuda.synchronize()
    assert torch.equal(img0.view(torch.int32), img1.view(torch.int32))
    tail = rows[n_img:]
    assert torch.equal(dT1[tail], dT0[tail])
    touched = torch.zeros(M, dtype=torch.bool, device=DEV)
    touched[tail] = True
    assert (dT1[~touched] == 0).all()                                      # the image rows' fp32 gradient never went to memory
    ref_b = (dT0[rows[:n_img]].double() * dact[rows[:n_img]].double()).sum(0)
    np.testing.assert_allclose(part.double().sum(0).cpu().numpy(), ref_b.cpu().numpy(), rtol=0, atol=2e-6 * float(ref_b.abs().max()) + 1e-9)
    np.testing.assert_allclose(b0.double().cpu().numpy(), ref_b.cpu().numpy(), rtol=0, atol=2e-6 * float(ref_b.abs().max()) + 1e-9)


def test_select_rows_in_image_order():
    """mhimx_select_rows_img: the same draw as mhimx_select_rows, and the kept rows once more as [stay | 0.. | merge from the offset | 0..]."""
    ops = _ops()
    N, k, n_sel, R = 5000, 150, 75, 490
    score = rnd(11, (N,)).to(DEV)
    tick = torch.tensor([5], dtype=torch.int64, device=DEV)
    rows = ops.select_rows(score, k, n_sel, R, 1234, tick=tick, merge_first=True)
    Lk = N - n_sel - R
    off = -(-(Lk + 5) // 32) * 32
    rows2, rimg = ops.select_rows_img(score, k, n_sel, R, 1234, off, tick=tick)
    torch.cuda.synchronize()
    assert torch.equal(rows, rows2)
    assert rimg.numel() == -(-(off + R) // 32) * 32
    assert torch.equal(rimg[:Lk], rows[R:]) and torch.equal(rimg[off:off + R], rows[:R])
    assert (rimg[Lk:off] == 0).all() and (rimg[off + R:] == 0).all()


def test_merge_with_gathered_rows_matches_contiguous():
    ops = _ops()
    from mhim_mil_amd import _lib as L
    M, E, k, R = 900, 512, 5, 333
    X = rnd(1, (M, E)).to(DEV)
    rows = torch.from_numpy(synth.permutation(2, M)[:R].copy()).to(DEV)
    q = rnd(3, (k, E), std=0.05).to(DEV)
    lnw, lnb = (1 + rnd(4, (E,), std=0.1)).to(DEV), rnd(5, (E,), std=0.1).to(DEV)
    wkv, wq, wo, bo = rnd(6, (1024, E), std=0.04).to(DEV), rnd(7, (512, E), std=0.04).to(DEV), rnd(8, (E, 512), std=0.04).to(DEV), rnd(9, (E,), std=0.1).to(DEV)
    tr = (ops.transpose(wkv), ops.transpose(wq), ops.transpose(wo))
    def mw(xr):
        return ops.MergeW(q, lnw, lnb, wkv, wq, wo, bo, 0.999, prec="bf16x3", transposes=tr, x_rows=xr)
    zg, _, wsg = ops.merge_fwd(mw(rows), X, update_q=False)
    zc, _, wsc = ops.merge_fwd(mw(None), X[rows].contiguous(), update_q=False)
    assert torch.equal(zg, zc)
    dz = rnd(10, (k, E)).to(DEV)
    dX = torch.zeros(M, E, device=DEV)
    gg = ops.merge_bwd(mw(rows), X, dz, wsg, grads={"dX": dX})
    gc = ops.merge_bwd(mw(None), X[rows].contiguous(), dz, wsc)
    assert torch.equal(dX[rows], gc["dX"])
    for key in ("d_ln_w", "d_ln_b", "d_wkv", "d_wq", "d_wo", "d_bo"):
        assert torch.equal(gg[key], gc[key]), key


def test_rows_dpre():
    ops = _ops()
    M, E, n = 1300, 512, 1000
    dH = rnd(1, (M, E), std=1e-3).to(DEV)
    dact = rnd(2, (M, E)).to(DEV).half()
    rows = torch.from_numpy(synth.permutation(3, M)[:n].copy()).to(DEV)
    dpre, cs = ops.rows_dpre(dH, dact, rows, n)
    ref = dH[rows].double() * dact[rows].double()
    np.testing.assert_allclose(dpre.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(cs.cpu().numpy(), ref.sum(0).float().cpu().numpy(), rtol=1e-4, atol=1e-7)
    dpre2, cs2 = ops.rows_dpre(dH, dact, None, M)
    np.testing.assert_allclose(dpre2.cpu().numpy(), (dH.double() * dact.double()).float().cpu().numpy(), rtol=1e-6)


def test_single_pass_step_equals_two_launch_step():
    """FusedTrainer: the single-pass step (bag-ordered buffers) against the two-projection step on the same draws (dropout off:
    the two forms draw different dropout streams)."""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.mhim import MHIM
    n, d = 2100, 256
    cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
               merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)

    def mk(sd):
        m = MHIM(input_dim=d, n_classes=2, baseline="attn", **cfg)
        sd = dict(sd)
        sd["merge.global_q"] = sd["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        m = m.to(DEV).train()
        m.merge.dropout = 0.0
        return m

    x = torch.from_numpy(synth.bag(99, n, d)).to(DEV)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    perm, shuf = torch.from_numpy(synth.permutation(1, k)).to(DEV), torch.from_numpy(synth.permutation(2, n - n_sel)).to(DEV)
    res = []
    for single in (True, False):
        s, t = mk(base), mk(tsd)
        tr = FusedTrainer(s, t)
        tr.single_pass = single
        label = torch.tensor([1], device=DEV)
        logits, losses = tr.forward_backward(x[None], label, perm=perm, ids_shuffle=shuf)
        res.append((logits.clone(), losses.clone(), tr.flat.grad.clone()))
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(res[0][1].cpu().numpy(), res[1][1].cpu().numpy(), atol=5e-5, rtol=0)
    g0, g1 = res[0][2].cpu().numpy(), res[1][2].cpu().numpy()
    np.testing.assert_allclose(g0, g1, atol=1e-3 * np.abs(g1).max(), rtol=1e-3)


def _draws_from_device(score_dev, rows_dev, R, n, k, n_sel):
    """The (perm, ids_shuffle) that make the oracle pick the row sets the device drew: rows_dev = [rows to merge (R) | rows that
    stay]; the masked rows are the rest.  The oracle selects on the SAME scores (score_dev), so its candidate list is the device's."""
    top = O.topk_indices(score_dev, k, True)
    part = np.zeros(n, dtype=bool)
    part[rows_dev] = True
    masked = np.nonzero(~part)[0]
    assert masked.shape[0] == n_sel
    pos = {int(v): i for i, v in enumerate(top)}
    assert all(int(m) in pos for m in masked), "a masked row is not among the top-k candidates"
    first = np.array([pos[int(m)] for m in masked], dtype=np.int64)
    rest = np.setdiff1d(np.arange(k), first)
    perm = np.concatenate([first, rest])
    kept = np.nonzero(part)[0]                                   # ascending, as the oracle's select emits them
    where = {int(v): i for i, v in enumerate(kept)}
    merge_rows, stay_rows = rows_dev[:R], rows_dev[R:]
    ids_shuffle = np.array([where[int(v)] for v in stay_rows] + [where[int(v)] for v in merge_rows], dtype=np.int64)
    return perm, ids_shuffle


def test_production_step_vs_oracle_c2():
    """The path bench.py times, at BASELINE's c2 size (N = 10 000, D = 1024): device-drawn hard-instance subsets
    (mhimx_select_rows, rows = [merge | stay]), single-pass projection, bag-ordered buffers, then a captured hipGraph replayed on
    a second bag.  The row sets and the teacher scores the device used are read back and handed to the oracle as its
    perm / ids_shuffle draws: logits 1e-4, every gradient 2e-3 of its scale, parameters after Adam + EMA.  Dropout off (the
    counter-hash masks have no CPU twin; tests above pin them)."""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.mhim import MHIM
    n, d = 10000, 1024
    cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
               merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)

    def mk(sd):
        m = MHIM(input_dim=d, n_classes=2, baseline="attn", **cfg)
        sd = dict(sd)
        sd["merge.global_q"] = sd["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        m = m.to(DEV).train()
        m.merge.dropout = 0.0
        return m

    s, t = mk(base), mk(tsd)
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997)
    ocfg = O.Cfg(**cfg)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    stu, tea, opt = O.as_torch(base), O.as_torch(tsd), {}
    bags = [torch.from_numpy(synth.bag(300 + i, n, d)) for i in range(2)]
    labels = [1, 0]

    def check(step, info, stu_new, tea_new, logits, losses):
        np.testing.assert_allclose(logits.cpu().numpy().ravel(), info["logits"].numpy().ravel(), atol=1e-4, rtol=0)
        assert abs(float(losses[0]) - info["loss"]) < 3e-4
        sd, td = s.state_dict(), t.state_dict()
        for name, ref in stu_new.items():          # Adam's first steps are sign-like: rounding-level gradient differences move the
            if name == "merge.global_q":           # few elements whose gradient is ~0 by up to 2 lr; the bulk must agree tightly
                continue
            err = (sd[name].detach().cpu().double() - ref.double()).abs()
            assert err.mean().item() <= 3e-6 and err.max().item() <= 4.1e-4 * (step + 1), (step, name, err.mean().item(), err.max().item())
            err = (td[name].detach().cpu().double() - tea_new[name].double()).abs()
            assert err.max().item() <= 2e-6, (step, "teacher", name, err.max().item())

    # ---- step 0, eager, gradients inspected before the update
    x0 = bags[0].to(DEV)
    logits, losses = tr.forward_backward(x0[None], torch.tensor([labels[0]], device=DEV))
    torch.cuda.synchronize()
    rows = tr.last["rows"].cpu().numpy()
    score = tr.last["score"].cpu().numpy()
    R = tr.last["R"]
    assert rows.shape[0] == n - n_sel and np.unique(rows).shape[0] == rows.shape[0]
    perm, shuf = _draws_from_device(score, rows, R, n, k, n_sel)
    stu1, tea1, opt, info = O.train_step(bags[0], labels[0], stu, tea, opt, ocfg, 1, perm=perm, ids_shuffle=shuf,
                                         score_override=torch.from_numpy(score))
    np.testing.assert_allclose(score, O.forward_teacher(bags[0], tea, ocfg)[1].numpy().ravel(), atol=1e-4, rtol=2e-3)
    gv = tr.flat.grad_views
    for name, ref in info["grads"].items():
        g, r = gv[name].cpu().numpy(), ref.numpy()
        np.testing.assert_allclose(g, r.reshape(g.shape), atol=2e-3 * (np.abs(r).max() + 1e-30), rtol=2e-3, err_msg=name)
    tr.update()
    torch.cuda.synchronize()
    check(0, info, stu1, tea1, logits, losses)

    # ---- step 1: the whole step as ONE captured hipGraph, replayed on the second bag
    s.load_state_dict({**{kk: vv for kk, vv in stu1.items()}, "merge.global_q": stu1["merge.global_q_mm"]})
    t.load_state_dict({**{kk: vv for kk, vv in tea1.items()}, "merge.global_q": tea1["merge.global_q_mm"]})
    x1 = bags[1].to(DEV)
    lab1 = torch.tensor([labels[1]], device=DEV)
    # Adam moments continue from step 0 on the device; the oracle carries `opt`.  (capture() warms up with real steps: snapshot
    # and restore the whole flat state around it so that exactly ONE step separates the states compared below.)
    snap = [tr.flat.student.clone(), tr.flat.teacher.clone(), tr.flat.m.clone(), tr.flat.v.clone(), tr.opt_step.clone(), tr.tick.clone(),
            tr.flat.step]
    g = tr.capture(x1[None], lab1, warmup=1)
    tr.flat.student.copy_(snap[0]); tr.flat.teacher.copy_(snap[1]); tr.flat.m.copy_(snap[2]); tr.flat.v.copy_(snap[3])
    tr.opt_step.copy_(snap[4]); tr.tick.copy_(snap[5]); tr.flat.step = snap[6]
    tr.flat.grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    rows = tr.last["rows"].cpu().numpy()
    score = tr.last["score"].cpu().numpy()
    perm, shuf = _draws_from_device(score, rows, tr.last["R"], n, k, n_sel)
    stu2, tea2, opt, info = O.train_step(bags[1], labels[1], stu1, tea1, opt, ocfg, 2, perm=perm, ids_shuffle=shuf,
                                         score_override=torch.from_numpy(score))
    check(1, info, stu2, tea2, tr.last["logits"], tr.last["losses"])


def test_production_step_with_dropout_vs_oracle_c2():
    """The timed configuration itself - dropout 0.25 on the teacher's AND the student's feature rows (the trainer keeps the teacher in
    train mode, base_engine.py:37-38; mhim.py:76) - at c2 size: the keep-masks the projection kernel drew from its counter hash are read
    back (a GELU output is exactly zero only where it was dropped) and handed to the oracle with the device's row sets and teacher scores:
    logits 1e-4, every gradient 2e-3 of its scale, parameters after Adam + EMA.  (Merge's own 0.1 dropouts stay off here: its kernels'
    masks are pinned by tests/test_ops_gpu.py.)"""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.mhim import MHIM
    n, d = 10000, 1024
    cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
               merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)

    def mk(sd):
        m = MHIM(input_dim=d, n_classes=2, baseline="attn", **cfg)
        sd = dict(sd)
        sd["merge.global_q"] = sd["merge.global_q_mm"]
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        m = m.to(DEV).train()
        m.merge.dropout = 0.0
        return m

    s, t = mk(base), mk(tsd)
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.9997)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    bag = torch.from_numpy(synth.bag(300, n, d))
    logits, losses = tr.forward_backward(bag.to(DEV)[None], torch.tensor([1], device=DEV))
    torch.cuda.synchronize()
    keep_t, keep_s = (tr.last["H_teacher"] != 0).cpu(), (tr.last["H_student"] != 0).cpu()
    for kp in (keep_t, keep_s):                                         # a 0.25 dropout, not a degenerate mask
        assert abs(float(kp.float().mean()) - 0.75) < 2e-3
    assert not torch.equal(keep_t, keep_s)
    rows, score = tr.last["rows"].cpu().numpy(), tr.last["score"].cpu().numpy()
    perm, shuf = _draws_from_device(score, rows, tr.last["R"], n, k, n_sel)
    stu1, tea1, _, info = O.train_step(bag, 1, O.as_torch(base), O.as_torch(tsd), {}, O.Cfg(**cfg), 1, perm=perm, ids_shuffle=shuf,
                                       score_override=torch.from_numpy(score), drop_mask_t=keep_t, drop_mask_s=keep_s)
    np.testing.assert_allclose(score, O.forward_teacher(bag, O.as_torch(tsd), O.Cfg(**cfg), keep_t)[1].numpy().ravel(), atol=1e-4, rtol=2e-3)
    np.testing.assert_allclose(logits.cpu().numpy().ravel(), info["logits"].numpy().ravel(), atol=1e-4, rtol=0)
    assert abs(float(losses[0]) - info["loss"]) < 3e-4
    gv = tr.flat.grad_views
    for name, ref in info["grads"].items():
        g, r = gv[name].cpu().numpy(), ref.numpy()
        np.testing.assert_allclose(g, r.reshape(g.shape), atol=2e-3 * (np.abs(r).max() + 1e-30), rtol=2e-3, err_msg=name)
    tr.update()
    torch.cuda.synchronize()
    sd, td = s.state_dict(), t.state_dict()
    for name, ref in stu1.items():
        if name == "merge.global_q":
            continue
        err = (sd[name].detach().cpu().double() - ref.double()).abs()
        assert err.mean().item() <= 3e-6 and err.max().item() <= 4.1e-4, (name, err.mean().item(), err.max().item())
        err = (td[name].detach().cpu().double() - tea1[name].double()).abs()
        assert err.max().item() <= 2e-6, ("teacher", name, err.max().item())
