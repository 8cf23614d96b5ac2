"""Round-6 GPU tests: the step as a DAG (mhimx_step_cfg.side_stream), ADVICE r5 fixes on the executor path."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(mask_ratio_h=0.03, mask_ratio_hr=0.5, merge_enable=True, merge_k=5, merge_mm=0.9999, merge_ratio=0.9, act="gelu", da_act="relu",
           attn2score=True, temp_t=0.1)


def _models(D=256, dropout=0.0, seed=3):
    from mhim_mil_amd.mhim import MHIM
    torch.manual_seed(seed)
    s = MHIM(input_dim=D, n_classes=2, baseline="attn", dropout=dropout, **CFG).cuda().train()
    t = copy.deepcopy(s)
    t.merge_test = False
    return s, t.train()


def _pair_of_trainers(D=512, dropout=0.25, **kw):
    from mhim_mil_amd.engine import FusedTrainer
    out = []
    for _ in range(2):
        torch.manual_seed(11)
        s, t = _models(D=D, dropout=dropout, seed=11)
        out.append(FusedTrainer(s, t, lr=1e-3, mm=0.999, aux_alpha=0.5, **kw))
    return out


def _state(tr):
    fl = tr.flat
    return [fl.student.clone(), fl.teacher.clone(), fl.m.clone(), fl.v.clone(), tr.opt_step.clone(), tr.tick.clone()]


def test_dag_step_equals_the_chain_bit_for_bit():
    """mhimx_step_cfg.side_stream: the executor forks the launches that share no data onto a second stream and joins them through events.
    Same kernels, arguments and seeds (the scorer-weight-gradient product keeps the slab count it has beside the Merge row tiles): the
    DAG step has the bits of the one-stream chain - over bags whose size changes every step, dropout on, nothing captured."""
    tr_d, tr_c = _pair_of_trainers()
    tr_d.step_dag, tr_c.step_dag = True, False
    g = torch.Generator(device="cuda").manual_seed(9)
    sizes = [2048, 1777, 3001, 1024, 2500, 10000]
    for step, n in enumerate(sizes):
        x = torch.randn(n, 512, device="cuda", generator=g).abs_()
        lab = torch.tensor([step % 2], device="cuda")
        ld, sd = tr_d.train_step(x, lab)
        lc, sc = tr_c.train_step(x, lab)
        assert tr_d._exec is not None and tr_d._exec["cfg"].side_stream and not tr_c._exec["cfg"].side_stream
        assert torch.equal(ld, lc) and torch.equal(sd, sc), (step, ld, lc)
        assert torch.equal(tr_d.last["rows"], tr_c.last["rows"]) and torch.equal(tr_d.last["tokens"], tr_c.last["tokens"])
        for a, b in zip(_state(tr_d), _state(tr_c)):
            assert torch.equal(a, b), step
    assert tr_d.flat.step == tr_c.flat.step == len(sizes)


def test_dag_step_forward_backward_only():
    """update = 0 through the DAG form: the complete gradient (both branches joined, every queued reduction flushed) equals the chain's."""
    tr_d, tr_c = _pair_of_trainers(clip_grad=0.5)
    tr_d.step_dag, tr_c.step_dag = True, False
    x = torch.rand(1500, 512, device="cuda")
    lab = torch.tensor([1], device="cuda")
    for tr in (tr_d, tr_c):
        tr.forward_backward(x, lab)
    assert torch.equal(tr_d.flat.grad, tr_c.flat.grad) and tr_d.flat.grad.abs().max() > 0
    for tr in (tr_d, tr_c):
        tr.update()
        tr.train_step(x, lab)
    for a, b in zip(_state(tr_d), _state(tr_c)):
        assert torch.equal(a, b)


def test_dag_step_captured_graph_has_branches_and_replays_equal():
    """A hipGraph of the DAG step (shape_cached: eager, captured on the second visit, replayed afterwards) equals the same sequence through
    the chain bit for bit: the fork / join events become graph dependencies, both branches end inside the capture."""
    tr_d, tr_c = _pair_of_trainers()
    tr_d.step_dag, tr_c.step_dag = True, False
    x = torch.rand(1800, 512, device="cuda")
    lab = torch.tensor([0], device="cuda")
    for it in range(5):
        od = tr_d.shape_cached("train_step", x, lab)
        oc = tr_c.shape_cached("train_step", x, lab)
        assert od is not None and oc is not None
        torch.cuda.synchronize()
        assert torch.equal(od[0], oc[0]) and torch.equal(od[1], oc[1]), it
    assert len(tr_d._shape_graphs["graphs"]) == 1 and not tr_d._shape_graphs["bad"], tr_d._shape_graphs.get("errors")
    for a, b in zip(_state(tr_d), _state(tr_c)):
        assert torch.equal(a, b)


def test_run_steps_takes_the_per_bag_path_when_the_executor_cannot_clip():
    """run_steps with --clip_grad (base_engine.py:115-119) used to run the executor's unclipped update.  It now goes bag by bag
    through train_step (which clips): identical to calling train_step in a loop; labels are validated; a per-bag iteration index is taken."""
    from mhim_mil_amd import _lib as L
    tr_m, tr_1 = _pair_of_trainers(clip_grad=0.05)
    g = torch.Generator(device="cuda").manual_seed(4)
    bags = [torch.randn(n, 512, device="cuda", generator=g).abs_() for n in (1200, 1536, 999)]
    labels = [torch.tensor([j % 2], device="cuda") for j in range(len(bags))]
    lm, _ = tr_m.run_steps(bags, labels, i=0)
    for j, (b, l) in enumerate(zip(bags, labels)):
        l1, _ = tr_1.train_step(b, l, i=j)
    assert torch.equal(lm, l1)
    for a, b in zip(_state(tr_m), _state(tr_1)):
        assert torch.equal(a, b)
    # and the clip did something: an unclipped trainer ends elsewhere
    tr_u, _ = _pair_of_trainers()
    for b, l in zip(bags, labels):
        tr_u.train_step(b, l)
    assert not torch.equal(tr_u.flat.student, tr_m.flat.student)
    tr_e, _ = _pair_of_trainers()
    with pytest.raises(L.MhimxError):
        tr_e.run_steps(bags, [torch.tensor([1])] * 3)                  # CPU labels
    with pytest.raises(L.MhimxError):
        tr_e.run_steps(bags, labels[:2])


@pytest.mark.parametrize("n,D", [(16385, 512), (20000, 512), (40000, 512), (120000, 256)])
def test_step_executor_takes_whole_slide_bags(n, D):
    """VERDICT r5 missing 1: bags of more than 16 384 rows (datasets/dataset_feat.py:93-111 yields whatever the slide has) ran the Python
    orchestration.  mhimx_step_run now issues the multi-workgroup select and the two keyed permutations MHIM.student_rows issues for such
    bags - same launches, same seeds: logits, row lists, parameters and optimiser state agree bit for bit with the Python path."""
    tr_c, tr_p = _pair_of_trainers(D=D)
    tr_p.use_executor = False
    g = torch.Generator(device="cuda").manual_seed(n)
    for step in range(2):
        x = torch.randn(n, D, device="cuda", generator=g).abs_()
        lab = torch.tensor([step % 2], device="cuda")
        assert tr_c._exec_ok(x)
        lc, sc = tr_c.train_step(x, lab)
        lp, sp = tr_p.train_step(x, lab)
        assert tr_c.last.get("ws") is not None, "the executor did not run"
        assert torch.equal(lc, lp) and torch.equal(sc, sp), (step, lc, lp)
        assert torch.equal(tr_c.last["rows"], tr_p.last["rows"]) and torch.equal(tr_c.last["score"], tr_p.last["score"])
        rows = tr_c.last["rows"]
        assert rows.unique().numel() == rows.numel() and int(rows.max()) < n            # a set of distinct bag rows
        for a, b in zip(_state(tr_c), _state(tr_p)):
            assert torch.equal(a, b), step


# ------------------------------------------------------------------------------------------------------------------------------
# a data-parallel rank's step through the executor: forward + backward with update = 0 and the global queries' EMA
# sent to the QueryChain's scratch (mhimx_step_cfg.q_out), then the all-reduce and the update as always
# ------------------------------------------------------------------------------------------------------------------------------
def _dp_exec_worker(rank, port, out, use_exec):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)                                   # (both ranks share the box's one GPU; gloo stages through the host)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from mhim_mil_amd.engine import FusedTrainer
    torch.manual_seed(11)
    s, t = _models(D=512, dropout=0.25, seed=11)
    tr = FusedTrainer(s, t, lr=1e-3, mm=0.999, aux_alpha=0.5)
    assert tr._chain is not None and tr.world == 2
    tr.overlap_comm = False                                    # (the mid-backward all-reduce hook lives in the Python orchestration)
    tr.use_executor = bool(use_exec)
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    logits = []
    for step, n in enumerate((2048, 1500 + 300 * rank, 17000)):
        x = torch.randn(n, 512, device="cuda", generator=g).abs_()
        lab = torch.tensor([(step + rank) % 2], device="cuda")
        lg, _ = tr.train_step(x, lab)
        assert (tr.last.get("ws") is not None) == bool(use_exec), "wrong step path"
        logits.append(lg.cpu().clone())
    torch.cuda.synchronize()
    torch.save({"logits": logits, "state": [v.cpu() for v in _state(tr)]}, os.path.join(out, f"dpx{int(use_exec)}{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_rank_step_through_the_executor(tmp_path):
    import os
    import torch.multiprocessing as mp
    res = {}
    for use_exec in (1, 0):
        port = 42300 + (os.getpid() % 500) + 7 * use_exec
        mp.spawn(_dp_exec_worker, args=(port, str(tmp_path), use_exec), nprocs=2, join=True)
        res[use_exec] = [torch.load(os.path.join(tmp_path, f"dpx{use_exec}{r}.pt")) for r in range(2)]
    for use_exec in (1, 0):                                    # replicas stay bit-identical (parameters, optimiser state, counters)
        for a, b in zip(res[use_exec][0]["state"], res[use_exec][1]["state"]):
            assert torch.equal(a, b)
    for r in range(2):      # and the executor's ranks follow the Python orchestration's: the first step's logits bit for bit, then to rounding
        # (the executor's scorer backward writes its share of the dPRE image itself - another order of the same weight-gradient sum)
        for j, (a, b) in enumerate(zip(res[1][r]["logits"], res[0][r]["logits"])):
            assert torch.equal(a, b) if j == 0 else torch.allclose(a, b, atol=2e-4), (r, j, a, b)
        for name, a, b in zip(("student", "teacher", "m", "v"), res[1][r]["state"], res[0][r]["state"]):
            d = (a.float() - b.float()).abs()
            assert float(d.mean()) <= 1.2e-5 and float(d.max()) <= 6.6e-3, (name, float(d.mean()), float(d.max()))
        for a, b in zip(res[1][r]["state"][4:], res[0][r]["state"][4:]):
            assert torch.equal(a, b)
