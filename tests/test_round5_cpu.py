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
