"""Round-5 host-logic tests that need no GPU."""
import copy

import torch

from mhim_mil_amd.mhim import MHIM


def _model():
    return MHIM(input_dim=64, n_classes=2, baseline="attn", mask_ratio_h=0.03, mask_ratio_hr=0.5, merge_enable=True, merge_k=5,
                merge_mm=0.9999, merge_ratio=0.9, act="gelu", da_act="relu", dropout=0.0)


def test_deepcopy_drops_the_fused_optimiser_mark():
    """a teacher adopted by FusedAdamEMA hides its parameters (``_ema_owned``); a deepcopy of it (a best-teacher snapshot)
    must be an ordinary module again."""
    m = _model()
    n_params = len(list(m.parameters()))
    assert n_params > 0
    m._ema_owned = True
    assert len(list(m.parameters())) == 0
    c = copy.deepcopy(m)
    assert not getattr(c, "_ema_owned", False)
    assert len(list(c.parameters())) == n_params
    # the copy is a real copy: same values, different storage, the global_q alias kept inside the copy
    for (na, pa), (nb, pb) in zip(m.named_parameters(), c.named_parameters()):
        assert na == nb and torch.equal(pa, pb) and pa.data_ptr() != pb.data_ptr()
    sd = c.state_dict()
    assert set(sd) == set(m.state_dict())


def test_pinned_stream_is_thread_local():
    """the launch-stream pin of a trainer step is per thread (and per device), not a process global."""
    import threading
    from mhim_mil_amd import ops
    ops._PIN.stream, ops._PIN.device = object(), -12345          # a pin of THIS thread (device index that is never current)
    seen = {}

    def other():
        seen["stream"] = getattr(ops._PIN, "stream", None)

    th = threading.Thread(target=other)
    th.start()
    th.join()
    assert seen["stream"] is None
    ops._PIN.stream, ops._PIN.device = None, -1


def test_schedule_at_updates_follows_the_loop_update_rule():
    """base_engine.py:47-49: an update after every accumulation_steps-th bag of an epoch AND after its last bag; the momentum schedule is
    read at epoch * len(loader) + batch_idx of that bag (base_engine.py:161-162).  ADVICE r5: the [acc - 1::acc] slice drifts when
    len(loader) % acc != 0."""
    from mhim_mil_amd.optim import schedule_at_updates
    sche = list(range(30))                           # 3 epochs of 10 bags, value == global bag index
    assert schedule_at_updates(sche, 1, None) is sche
    assert schedule_at_updates(sche, 5, None) == sche[4::5]
    assert schedule_at_updates(sche, 5, 10) == sche[4::5]                    # acc divides len: the two agree
    got = schedule_at_updates(sche, 4, 10)                                   # updates at bags 3, 7, 9 of every epoch
    assert got == [3, 7, 9, 13, 17, 19, 23, 27, 29]
    assert schedule_at_updates(sche, 1, 10) == sche


def test_step_layout_takes_whole_slide_bags_and_refuses_beyond_the_limit():
    """mhimx_step_layout_of is host code: the executor's shape checks (csrc/step.hip:check_cfg) without a GPU.  Round 6: bags up to
    MHIMX_STEP_MAX_ROWS = 262 144 rows (the multi-workgroup select above 16 384), k_top <= 16 384, rows to merge <= 32 768."""
    import ctypes as C
    from mhim_mil_amd import _lib as L
    lib = L.lib()
    one = C.c_void_p(256)                                  # any non-null "device pointer": nothing is dereferenced on the host
    par = L.StepParams(**{f: 256 for f, _ in L.StepParams._fields_})
    grd = L.StepGrads(**{f: 256 for f, _ in L.StepGrads._fields_})
    cfg = L.StepCfg(D=1024, E=512, A=128, C=2, k=5, act=2, da_act=1, attn2score=1, student=par, teacher=par, grad=grd, tick=256)
    totals = []
    for n in (64, 10000, 16384, 16385, 60000, 262144):
        cnt, lay = L.StepCounts(), L.StepLayout()
        assert lib.mhimx_step_counts_of(n, 0.03, 0.5, 0.9, C.byref(cnt)) == 0
        assert lib.mhimx_step_layout_of(C.byref(cfg), n, C.byref(cnt), C.byref(lay)) == 0, (n, lib.mhimx_last_error())
        assert lay.total > 0 and lay.rows_all > 0 and lay.logits > 0
        totals.append(lay.total)
    assert totals == sorted(totals)
    cnt, lay = L.StepCounts(), L.StepLayout()
    assert lib.mhimx_step_counts_of(262145, 0.03, 0.5, 0.9, C.byref(cnt)) == 0
    assert lib.mhimx_step_layout_of(C.byref(cfg), 262145, C.byref(cnt), C.byref(lay)) < 0
    assert b"262144" in lib.mhimx_last_error()
    # a recipe that leaves more than 32 768 rows to merge (merge_ratio 0.5 on 200 000 rows) is refused as well
    assert lib.mhimx_step_counts_of(200000, 0.03, 0.5, 0.5, C.byref(cnt)) == 0 and cnt.R > 32768
    assert lib.mhimx_step_layout_of(C.byref(cfg), 200000, C.byref(cnt), C.byref(lay)) < 0
    _ = one
