"""Sequence-parallel Nystrom TransLayer (mhim_mil_amd/nystrom_sharded.py): two and four ranks sharing the box's one GPU (gloo, host staging)
against the single-rank layer on the whole sequence: outputs, input gradients and the SUM of the ranks' parameter gradients."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, E = 2048, 512


def _layer_and_input():
    from mhim_mil_amd import nystrom as NY
    torch.manual_seed(11)
    layer = NY.TransLayer(E)
    with torch.no_grad():                                  # non-trivial LayerNorm / bias / conv parameters
        layer.norm.weight.add_(0.1 * torch.randn(E))
        layer.norm.bias.add_(0.1 * torch.randn(E))
        layer.attn.to_out[0].bias.add_(0.05 * torch.randn(E))
    g = torch.Generator().manual_seed(12)
    x = torch.randn(T, E, generator=g) * 0.7
    dy = torch.randn(T, E, generator=g) * 0.1
    return layer, x, dy


def _reference():
    layer, x, dy = _layer_and_input()
    layer = layer.to(DEV).eval()
    xd = x.to(DEV).requires_grad_()
    y = layer(xd, False, False, 0, None, False)
    y.backward(dy.to(DEV))
    return y.detach().cpu(), xd.grad.cpu(), {k: p.grad.detach().cpu() for k, p in layer.named_parameters()}


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mhim_mil_amd.nystrom_sharded import sharded_trans_layer
    layer, x, dy = _layer_and_input()
    layer = layer.to(DEV).eval()
    Tr = T // world
    xl = x[rank * Tr:(rank + 1) * Tr].to(DEV).requires_grad_()
    y = sharded_trans_layer(layer, xl)
    y.backward(dy[rank * Tr:(rank + 1) * Tr].to(DEV))
    torch.save({"y": y.detach().cpu(), "dx": xl.grad.cpu(), "g": {k: p.grad.detach().cpu() for k, p in layer.named_parameters()}},
               os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_layer_equals_the_whole_sequence(tmp_path, world):
    y_ref, dx_ref, g_ref = _reference()
    port = 34100 + (os.getpid() % 1500) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    y = torch.cat([r["y"] for r in res])
    dx = torch.cat([r["dx"] for r in res])
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(y, y_ref) < 2e-5
    assert rel(dx, dx_ref) < 2e-4
    for k, ref in g_ref.items():
        tot = sum(r["g"][k] for r in res)                  # parameter gradients are local partial sums: the flat all-reduce adds them
        assert rel(tot, ref) < 5e-4, (k, rel(tot, ref))


def test_shard_alignment_is_checked():
    from mhim_mil_amd import _lib as L
    from mhim_mil_amd import nystrom as NY
    from mhim_mil_amd.nystrom_sharded import sharded_trans_layer
    layer = NY.TransLayer(E).to(DEV).eval()
    with pytest.raises(L.MhimxError):
        sharded_trans_layer(layer, torch.randn(200, E, device=DEV))            # (a world of one: 200 tokens are not 256 landmarks' worth)


# ---------------------------------------------------------------------------------------------------- encoder level
N_TOK = 1850                                               # + cls = 1851 tokens -> 197 front pad rows -> T = 2048


def _encoder_and_tokens(n_tok=None):
    n_tok = N_TOK if n_tok is None else n_tok
    from mhim_mil_amd import nystrom as NY
    torch.manual_seed(21)
    enc = NY.SAttention(E)
    with torch.no_grad():
        enc.cls_token.mul_(0.3)
        enc.norm.weight.add_(0.1 * torch.randn(E))
        enc.norm.bias.add_(0.1 * torch.randn(E))
    g = torch.Generator().manual_seed(22)
    h = torch.randn(n_tok, E, generator=g) * 0.7
    dz = torch.randn(E, generator=g)
    return enc, h, dz


def _enc_worker(rank, world, port, out, n_tok=None):
    n_tok = N_TOK if n_tok is None else n_tok
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mhim_mil_amd.nystrom_sharded import sharded_sattention
    enc, h, dz = _encoder_and_tokens(n_tok)
    enc = enc.to(DEV).eval()
    n = n_tok + 1
    pad = (256 - n % 256) % 256
    seq = torch.cat([torch.zeros(pad + 1, E), h])          # [zeros(pad) | cls slot | tokens]
    Tr = seq.shape[0] // world
    hl = seq[rank * Tr:(rank + 1) * Tr].to(DEV).requires_grad_()
    z = sharded_sattention(enc, hl, pad, n)
    z.backward(dz.to(DEV))
    torch.save({"z": z.detach().cpu(), "dh": hl.grad.cpu(), "g": {k: (p.grad.detach().cpu() if p.grad is not None else None)
                                                                    for k, p in enc.named_parameters()}}, os.path.join(out, f"e{rank}.pt"))
    dist.destroy_process_group()


# (1850 tokens: a 44 x 44 grid with 86 wrap cells, every rank holds tokens; 300 tokens over 4 ranks: T = 512, the first rank is all front
#  padding, the second holds the cls row and 44 tokens - the PPEG's band exchange with ranks that own nothing; 3000: several grid rows
#  per halo, blocks that do not start on a grid row)
@pytest.mark.parametrize("world,n_tok", [(2, N_TOK), (4, N_TOK), (4, 300), (2, 3000)])
def test_sharded_encoder_equals_the_single_rank_encoder(tmp_path, world, n_tok):
    enc, h, dz = _encoder_and_tokens(n_tok)
    enc = enc.to(DEV).eval()
    hd = h.to(DEV).requires_grad_()
    z_ref = enc(hd)
    z_ref.backward(dz.to(DEV))
    g_ref = {k: p.grad.detach().cpu() for k, p in enc.named_parameters()}
    port = 35100 + (os.getpid() % 1500) + world
    mp.spawn(_enc_worker, args=(world, port, str(tmp_path), n_tok), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"e{r}.pt")) for r in range(world)]
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    for r in res:                                           # the cls feature is replicated
        assert rel(r["z"], z_ref.detach().cpu()) < 5e-5
    pad = (256 - (n_tok + 1) % 256) % 256
    dh = torch.cat([r["dh"] for r in res])[pad + 1:]        # the token rows of the padded sequence
    assert rel(dh, hd.grad.cpu()) < 5e-4
    for k, ref in g_ref.items():
        parts = [r["g"][k] for r in res if r["g"][k] is not None]
        assert parts, k
        assert rel(sum(parts), ref) < 2e-3, (k, rel(sum(parts), ref))
