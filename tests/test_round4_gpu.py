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
