"""MHIM(TransMIL) train step on an instance-sharded bag (mhim_mil_amd/sharded_transmil.py, SURVEY.md §8(e) third line) against the
single-process FusedTrainer on the whole bag: world 1 in-process; 2 and 4 ranks as processes sharing the one GPU (gloo moves the buffers;
RCCL on two devices when the box has them)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mhim_mil_amd import synth
from oracle import mhim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
V2 = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
          merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
N, D = 1900, 256                       # (D % 256 == 0: the single-pass projection + matrix-core-image weight gradient take these shapes)
COUNTS = {1: [1900], 2: [1000, 900], 4: [500, 450, 550, 400]}


def build(sd, **kw):
    from mhim_mil_amd.mhim import MHIM
    m = MHIM(baseline="selfattn", n_classes=2, **kw)
    sd = dict(sd)
    sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV)
    m.merge.dropout = 0.0
    m.online_encoder.layer1.attn.dropout = 0.0
    m.online_encoder.layer2.attn.dropout = 0.0
    return m.train()


def _models(attn2score=True):
    base = synth.mhim_state(31, input_dim=D, merge_k=5, baseline="selfattn")
    kw = {**V2, "attn2score": attn2score}
    return build(base, input_dim=D, **kw), build(synth.spread_teacher(base), input_dim=D, **kw)


def _draws():
    k, n_sel, _ = O.mask_count(N, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    return [(torch.from_numpy(synth.permutation(30 + s, k)).to(DEV), torch.from_numpy(synth.permutation(40 + s, N - n_sel)).to(DEV))
            for s in range(2)]


def _reference_run(attn2score=True):
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models(attn2score)
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999)
    outs = []
    for step, (perm, shuf) in enumerate(_draws()):
        x = torch.from_numpy(synth.bag(900 + step, N, D)).to(DEV)
        logits, losses = tr.train_step(x, torch.tensor([step % 2], device=DEV), perm=perm, ids_shuffle=shuf)
        outs.append((logits.cpu(), losses.cpu()))
    return outs, {k: v.detach().cpu() for k, v in s.state_dict().items()}, {k: v.detach().cpu() for k, v in t.state_dict().items()}


def _run_sharded(rank, world, attn2score=True):
    from mhim_mil_amd.sharded import ShardedBagTrainer
    s, t = _models(attn2score)
    counts = COUNTS[world]
    tr = ShardedBagTrainer(s, t, counts=counts, aux_alpha=0.5, mm=0.999)
    lo = sum(counts[:rank])
    res = {"logits": [], "losses": [], "rows": [], "score": []}
    for step, (perm, shuf) in enumerate(_draws()):
        x = torch.from_numpy(synth.bag(900 + step, N, D))[lo:lo + counts[rank]].to(DEV)
        logits, losses = tr.train_step(x, torch.tensor([step % 2], device=DEV), perm=perm, ids_shuffle=shuf)
        res["logits"].append(logits.cpu())
        res["losses"].append(losses.cpu())
        res["rows"].append(tr.last["rows"].cpu())
        res["score"].append(tr.last["score"].cpu())
    res["stu"] = {k: v.detach().cpu() for k, v in s.state_dict().items()}
    res["tea"] = {k: v.detach().cpu() for k, v in t.state_dict().items()}
    return res


def _check(res, ref):
    outs, s_ref, t_ref = ref
    for step in range(2):
        np.testing.assert_allclose(res["logits"][step].numpy(), outs[step][0].numpy(), atol=1e-4, rtol=0)
        np.testing.assert_allclose(res["losses"][step].numpy(), outs[step][1].numpy(), atol=2e-4, rtol=0)
    # Adam's first steps are sign-like: elements whose gradient is rounding noise may move the other way (see test_sharded_gpu.py)
    for r, got in ((s_ref, res["stu"]), (t_ref, res["tea"])):
        for k, v in r.items():
            err = (got[k].double() - v.double()).abs()
            assert err.mean().item() <= 4e-6 and err.max().item() <= 4.1e-4, (k, err.mean().item(), err.max().item())


@pytest.mark.parametrize("attn2score", [True, False])
def test_world1_equals_fused_trainer(attn2score):
    _check(_run_sharded(0, 1, attn2score), _reference_run(attn2score))


def _worker(rank, world, port, out, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run_sharded(rank, world)
    torch.save(res, os.path.join(out, f"g{rank}.pt"))
    dist.destroy_process_group()


def _spawn_and_check(tmp_path, world, backend):
    ref = _reference_run()
    port = 36100 + (os.getpid() % 1500) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), backend), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(world)]
    for r in res[1:]:
        for step in range(2):
            assert torch.equal(r["rows"][step], res[0]["rows"][step])             # the same row lists on every rank
            assert torch.equal(r["logits"][step], res[0]["logits"][step])
        for k, v in res[0]["stu"].items():                                        # replicas stay bit-identical
            assert torch.equal(r["stu"][k], v), k
        for k, v in res[0]["tea"].items():
            assert torch.equal(r["tea"][k], v), k
    _check(res[0], ref)


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_equal_single_process(tmp_path, world):
    _spawn_and_check(tmp_path, world, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_two_devices_rccl_equal_single_process(tmp_path):
    _spawn_and_check(tmp_path, 2, "nccl")


def test_exchange_plan_round_trip():
    """_AssembleTokens on one rank: rows land at pad + 1 + j, the tail behind them, and the backward returns every gradient row."""
    from mhim_mil_amd.sharded import _Comm
    from mhim_mil_amd.sharded_transmil import ExchangePlan, _AssembleTokens, seq_layout
    n_tok, k, E = 300, 5, 512
    pad, T, Tr = seq_layout(n_tok + k, 1)
    ids = torch.from_numpy(synth.permutation(3, 1000)[:n_tok].astype(np.int64)).to(DEV)
    plan = ExchangePlan(ids, [0, 1000], pad, Tr, _Comm())
    rows = torch.randn(n_tok, E, device=DEV, requires_grad=True)
    tail = torch.randn(k, E, device=DEV, requires_grad=True)
    blk = _AssembleTokens.apply(rows, tail, plan, pad + 1 + n_tok)
    assert blk.shape == (T, E) and torch.equal(blk[pad + 1:pad + 1 + n_tok], rows.detach()) and torch.equal(blk[pad + 1 + n_tok:], tail.detach())
    assert float(blk.detach()[:pad + 1].abs().max()) == 0.0
    g = torch.randn(T, E, device=DEV)
    blk.backward(g)
    assert torch.equal(rows.grad, g[pad + 1:pad + 1 + n_tok]) and torch.equal(tail.grad, g[pad + 1 + n_tok:])
