"""Data-parallel update logic on CPU: world_size-2 gloo processes (the GPU path uses the same function over RCCL).

Checks: the all-reduced flat gradient scaled by 1/world equals the gradient a single process
accumulates over the same bags with accumulation_steps = world, and the non-trainable tail (merge.global_q_mm, which
every rank EMA-updates in its own forward) comes back as the rank mean.
"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mhim_mil_amd import synth
from oracle import mhim_oracle as O

N_TRAIN, N_ALL, WORLD = 1000, 1040, 2


def _rank_buffers(rank):
    g = torch.from_numpy(synth.normal(100 + rank, (N_ALL,), std=1e-2).astype(np.float32))
    g[N_TRAIN:] = 0
    p = torch.from_numpy(synth.normal(7, (N_ALL,), std=0.1).astype(np.float32))
    p[N_TRAIN:] += 0.01 * (rank + 1)             # each rank's own in-forward EMA moved the tail differently
    return g, p


def _worker(rank, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from mhim_mil_amd.engine import sync_flat_gradient
    g, p = _rank_buffers(rank)
    scale = sync_flat_gradient(g, p, N_TRAIN, WORLD)
    torch.save({"g": g, "p": p, "scale": scale}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(WORLD)]
    g0, p0 = _rank_buffers(0)
    g1, p1 = _rank_buffers(1)
    for r in res:
        assert r["scale"] == 0.5
        np.testing.assert_allclose(r["g"][:N_TRAIN].numpy(), (g0 + g1)[:N_TRAIN].numpy(), rtol=1e-6)
        assert float(r["g"][N_TRAIN:].abs().max()) == 0.0
        np.testing.assert_allclose(r["p"][N_TRAIN:].numpy(), ((p0 + p1) / 2)[N_TRAIN:].numpy(), rtol=1e-6)
        np.testing.assert_allclose(r["p"][:N_TRAIN].numpy(), p0[:N_TRAIN].numpy())
    # identical on both ranks => replicas stay in lock-step
    assert torch.equal(res[0]["g"], res[1]["g"]) and torch.equal(res[0]["p"], res[1]["p"])
    # == single-process gradient accumulation over the same two bags (loss/accum, base_engine.py:102)
    acc = g0 / WORLD + g1 / WORLD
    np.testing.assert_allclose((res[0]["g"] * res[0]["scale"])[:N_TRAIN].numpy(), acc[:N_TRAIN].numpy(), rtol=1e-6)
    # and the Adam step on it equals the oracle's Adam on the accumulated gradient
    pn, _, _ = O.adam_step(p0[:N_TRAIN], acc[:N_TRAIN], torch.zeros(N_TRAIN), torch.zeros(N_TRAIN), 1)
    pn2, _, _ = O.adam_step(p0[:N_TRAIN], (res[1]["g"] * 0.5)[:N_TRAIN], torch.zeros(N_TRAIN), torch.zeros(N_TRAIN), 1)
    np.testing.assert_allclose(pn.numpy(), pn2.numpy(), rtol=1e-6)


def _chain_worker(rank, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from mhim_mil_amd.engine import QueryChain, sync_flat_gradient
    g, p = _rank_buffers(rank)
    p[N_TRAIN:] = _rank_buffers(0)[1][N_TRAIN:]                 # the queries are q0 on every rank until the update
    ch = QueryChain(N_TRAIN, 40, 0.99, WORLD, rank)
    ch.tokens = torch.from_numpy(synth.normal(300 + rank, (5, 8), std=1.0).astype(np.float32))
    sync_flat_gradient(g, p, N_TRAIN, WORLD, chain=ch)
    torch.save({"g": g, "p": p}, os.path.join(out, f"c{rank}.pt"))
    dist.destroy_process_group()


def test_query_chain_two_ranks(tmp_path):
    """QueryChain: the EMA of Merge's global queries over the ranks of one update == the bag-after-bag chain of a single process on the
    same tokens, q <- mm q + (1 - mm) z_r for r = 0, 1 (merge.py:142-143) - not the rank mean of one-step results."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_chain_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(os.path.join(tmp_path, f"c{r}.pt")) for r in range(WORLD)]
    q = _rank_buffers(0)[1][N_TRAIN:].double()
    for r in range(WORLD):
        z = torch.from_numpy(synth.normal(300 + r, (5, 8), std=1.0)).double().reshape(-1)
        q = 0.99 * q + 0.01 * z
    for r in res:
        np.testing.assert_allclose(r["p"][N_TRAIN:].double().numpy(), q.numpy(), rtol=0, atol=2e-7)
        assert float(r["g"][N_TRAIN:].abs().max()) == 0.0
    assert torch.equal(res[0]["p"], res[1]["p"])


# ---------------------------------------------------------------------------------------------------------------------
# Instance-sharded bag (BASELINE config c5): the exchange + partition logic of mhim_mil_amd/sharded.py on 2 gloo ranks,
# with the oracle's math standing in for the shard-local kernels.  (The kernels themselves need a GPU: tests/test_sharded_gpu.py.)
# ---------------------------------------------------------------------------------------------------------------------
SH_N, SH_D, SH_COUNTS = 900, 64, [500, 400]


def _lse_merge_np(parts):
    """What mhimx_lse_merge computes: parts [W, 2+E] = (max_w, L_w, z_w) -> (max, L, z)."""
    live = parts[:, 1] > 0
    M = parts[live, 0].max()
    w = parts[:, 1] * np.where(live, np.exp(parts[:, 0] - M), 0.0)
    return M, w.sum(), (parts[:, 2:] * w[:, None]).sum(0) / w.sum()


def _shard_worker(rank, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from mhim_mil_amd.sharded import _Comm, partition_rows
    cm = _Comm()
    assert (cm.world, cm.rank, cm.stage) == (WORLD, rank, True)
    p = O.as_torch(synth.mhim_state(5, input_dim=SH_D, merge_k=3))
    lo = sum(SH_COUNTS[:rank])
    x = torch.from_numpy(synth.bag(77, SH_N, SH_D))[lo:lo + SH_COUNTS[rank]]
    # shard-local teacher: scorer logits, local softmax statistics and pooled feature
    h = O.feature(x, p, "gelu")
    s = O.scorer_logits(h, p["online_encoder.attention.attention.0.weight"], p["online_encoder.attention.attention.2.weight"], "relu").view(-1)
    mx = s.max()
    e = torch.exp(s - mx)
    part = torch.cat([mx.view(1), e.sum().view(1), (e[:, None] * h).sum(0) / e.sum()])
    parts = cm.all_gather(part)                                            # [W, 2+E]
    M, Lsum, z = _lse_merge_np(parts.double().numpy())
    attn_local = torch.exp(s - M) / Lsum                                   # needs the GLOBAL denominators
    attn = cm.all_gather_rows(attn_local.float(), SH_COUNTS)               # unequal shard sizes
    # replicated row list [stay | merge] and this shard's slice of it
    rows = torch.from_numpy(synth.permutation(9, SH_N)[:800].astype(np.int64))
    Lk = 720
    rows_local, n_stay, merge_pos = partition_rows(rows, Lk, lo, SH_COUNTS[rank])
    Hm = torch.zeros(800 - Lk, 4)
    Hm[merge_pos] = (rows_local[n_stay:] + lo).float()[:, None].expand(-1, 4)      # "features" = the global row id
    cm.all_reduce_sum(Hm)
    torch.save({"z": torch.from_numpy(z), "attn": attn, "rows_local": rows_local + lo, "n_stay": n_stay, "Hm": Hm},
               os.path.join(out, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_bag_exchanges_two_ranks(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_shard_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(os.path.join(tmp_path, f"s{r}.pt")) for r in range(WORLD)]
    p = O.as_torch(synth.mhim_state(5, input_dim=SH_D, merge_k=3))
    x = torch.from_numpy(synth.bag(77, SH_N, SH_D))
    h = O.feature(x, p, "gelu")
    z_ref, a_ref, _ = O.dattention(h, p, "relu")                           # the unsharded softmax pool
    rows = torch.from_numpy(synth.permutation(9, SH_N)[:800].astype(np.int64))
    for r in res:
        np.testing.assert_allclose(r["z"].numpy(), z_ref.numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(r["attn"].numpy(), a_ref.view(-1).numpy(), rtol=2e-5, atol=1e-9)
        # the all-reduced merge block holds every merge row exactly once, in merge-list order
        np.testing.assert_array_equal(r["Hm"][:, 0].long().numpy(), rows[720:].numpy())
    # the shards' row lists partition the replicated list: stay rows first, order preserved
    stay = torch.cat([r["rows_local"][:r["n_stay"]] for r in res])
    merge = torch.cat([r["rows_local"][r["n_stay"]:] for r in res])
    assert sorted(stay.tolist()) == sorted(rows[:720].tolist()) and sorted(merge.tolist()) == sorted(rows[720:].tolist())
    lo1 = SH_COUNTS[0]
    assert all(v < lo1 for v in res[0]["rows_local"].tolist()) and all(v >= lo1 for v in res[1]["rows_local"].tolist())


def _fixed_worker(rank, port, out):
    """The fixed-shape formulation of the sharded student (sharded._fixed_gen), restated with torch-CPU math: every rank pools ALL its
    rows with excluded rows at score -inf (what mhimx_pool_io.excl does), collects the merge rows by ownership (mhimx_shard_gather) and
    exchanges through _Comm's in-place collectives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from mhim_mil_amd.sharded import _Comm
    cm = _Comm()
    p = O.as_torch(synth.mhim_state(5, input_dim=SH_D, merge_k=3))
    n, lo = SH_COUNTS[rank], sum(SH_COUNTS[:rank])
    x = torch.from_numpy(synth.bag(77, SH_N, SH_D))[lo:lo + n]
    rows_all = torch.from_numpy(synth.permutation(9, SH_N)[:800].astype(np.int64))          # [merge R | stay Lk]
    R, Lk, k_tok = 80, 720, 3
    h = O.feature(x, p, "gelu")
    tokens = torch.from_numpy(synth.normal(123, (k_tok, h.shape[1]), std=0.3).astype(np.float32))   # "merged tokens" (replicated)
    hbuf = torch.cat([h, tokens])
    # mhimx_shard_flags
    excl = torch.ones(n + k_tok, dtype=torch.bool)
    stay = rows_all[R:R + Lk]
    own = (stay >= lo) & (stay < lo + n)
    excl[stay[own] - lo] = False
    excl[n:] = rank != 0
    # pool with flags: excluded rows get score -inf
    s = O.scorer_logits(hbuf, p["online_encoder.attention.attention.0.weight"], p["online_encoder.attention.attention.2.weight"], "relu").view(-1)
    s = torch.where(excl, torch.full_like(s, float("-inf")), s)
    mx = s.max()
    e = torch.exp(s - mx)
    part = torch.cat([mx.view(1), e.sum().view(1), (e[:, None] * hbuf).sum(0) / e.sum()])
    parts = torch.empty((WORLD, part.numel()))
    cm.all_gather_into(parts, part)
    M, Lsum, z = _lse_merge_np(parts.double().numpy())
    # mhimx_shard_gather + all-reduce
    mr = rows_all[:R]
    ownm = (mr >= lo) & (mr < lo + n)
    Hm = torch.zeros((R, h.shape[1]))
    Hm[ownm] = h[mr[ownm] - lo]
    cm.all_reduce_sum(Hm)
    torch.save({"z": torch.from_numpy(z), "Hm": Hm}, os.path.join(out, f"f{rank}.pt"))
    dist.destroy_process_group()


def test_fixed_shape_sharded_student_two_ranks(tmp_path):
    port = 32500 + (os.getpid() % 2000)
    mp.spawn(_fixed_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(os.path.join(tmp_path, f"f{r}.pt")) for r in range(WORLD)]
    p = O.as_torch(synth.mhim_state(5, input_dim=SH_D, merge_k=3))
    x = torch.from_numpy(synth.bag(77, SH_N, SH_D))
    h = O.feature(x, p, "gelu")
    rows_all = torch.from_numpy(synth.permutation(9, SH_N)[:800].astype(np.int64))
    tokens = torch.from_numpy(synth.normal(123, (3, h.shape[1]), std=0.3).astype(np.float32))
    z_ref, _, _ = O.dattention(torch.cat([h[rows_all[80:]], tokens]), p, "relu")             # the unsharded pool over [stay | tokens]
    for r in res:
        np.testing.assert_allclose(r["z"].numpy(), z_ref.numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_array_equal(r["Hm"].numpy(), h[rows_all[:80]].numpy())                # every merge row exactly once, in list order


def test_cosine_scheduler_matches_reference_fixture():
    """engine.cosine_scheduler == utils.cosine_scheduler (utils.py:199-210) on the fixture made from the reference."""
    from tests import golden_util as G
    from mhim_mil_amd.engine import cosine_scheduler
    meta, a = G.load("g12_cosine_scheduler")
    for i, (base, final, ep, nit, warm, start) in enumerate(meta["cases"]):
        got = cosine_scheduler(base, final, epochs=ep, niter_per_ep=nit, warmup_epochs=warm, start_warmup_value=start)
        np.testing.assert_allclose(got, a[f"c{i}"], rtol=0, atol=1e-15)


# ---------------------------------------------------------------------------------------- sharded TransMIL: the re-balancing exchange
def _exchange_case():
    """A bag of 1000 rows over two shards [0, 600) | [600, 1000); 300 kept rows in shuffled token order; 5 tail tokens."""
    ids = torch.from_numpy(synth.permutation(3, 1000)[:300].astype(np.int64))
    rows = torch.from_numpy(synth.normal(5, (1000, 8), std=1.0).astype(np.float32))       # "feature rows" of the whole bag
    tail = torch.from_numpy(synth.normal(6, (5, 8), std=1.0).astype(np.float32))
    g = torch.from_numpy(synth.normal(7, (512, 8), std=1.0).astype(np.float32))           # d block of the whole sequence (T = 512)
    return ids, rows, tail, g, [0, 600, 1000]


def _exchange_worker(rank, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from mhim_mil_amd.sharded import _Comm
    from mhim_mil_amd.sharded_transmil import ExchangePlan, _AssembleTokens, seq_layout
    ids, rows, tail, g, bounds = _exchange_case()
    pad, T, Tr = seq_layout(300 + 5, WORLD)
    plan = ExchangePlan(ids, bounds, pad, Tr, _Comm())
    mine = rows[ids[plan.local]].clone().requires_grad_()            # my kept rows in ascending token order
    tl = tail.clone().requires_grad_()
    blk = _AssembleTokens.apply(mine, tl, plan, pad + 1 + 300)
    blk.backward(g[rank * Tr:(rank + 1) * Tr])
    torch.save({"blk": blk.detach(), "drows": mine.grad, "dtail": tl.grad, "local": plan.local, "pad": pad, "T": T}, os.path.join(out, f"x{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_transmil_token_exchange_two_ranks(tmp_path):
    """sharded_transmil._AssembleTokens over gloo: the ranks' blocks are the sequence [zeros(pad) | cls slot | kept rows in token order |
    tail]; the backward returns each row's gradient to its owner and the tail's gradient (summed over its owners) to everybody."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(os.path.join(tmp_path, f"x{r}.pt")) for r in range(WORLD)]
    ids, rows, tail, g, bounds = _exchange_case()
    pad, T = res[0]["pad"], res[0]["T"]
    seq = torch.cat([r["blk"] for r in res])
    assert seq.shape[0] == T == 512 and pad == 512 - 306
    assert float(seq[:pad + 1].abs().max()) == 0.0
    assert torch.equal(seq[pad + 1:pad + 301], rows[ids]) and torch.equal(seq[pad + 301:], tail)
    for r in range(WORLD):
        loc = res[r]["local"]
        assert bool(((ids[loc] >= bounds[r]) & (ids[loc] < bounds[r + 1])).all())
        assert torch.equal(res[r]["drows"], g[pad + 1 + loc])
        assert torch.equal(res[r]["dtail"], g[pad + 301:])
    assert sum(r["local"].numel() for r in res) == 300
