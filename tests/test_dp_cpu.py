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
