"""Sequence-parallel Nystrom TransLayer: ONE bag's token sequence sharded over the ranks (third line).

Rank r holds the contiguous token block [r T/W, (r+1) T/W) of a sequence of T tokens (T % 256 == 0, W | 256, T/W a multiple of 64:
every rank owns 256/W WHOLE landmark groups - the caller re-balances ragged shards first).  What couples the shards in
``y = x + to_out(Nystrom(to_qkv(LayerNorm(x))))`` (baseline.py:213-218, nystrom_attention.py:65-152), and how it is exchanged:

  forward   landmark means q~, k~        own groups, all-gather of [256/W, 1024]
            attn2, its pseudo-inverse    landmark-only: replicated on every rank (bit-identical replicas, no traffic)
            a3v = softmax_T(q~ k^T) v    the streamed kernel's per-shard result (a3v_r, lse_r) merged over ranks in a fixed order:
                                         lse = log2 sum_r 2^lse_r, a3v = sum_r 2^(lse_r - lse) a3v_r  (all-gather of 8 x 256 x 65 floats)
            out = softmax(q k~^T) w2     per token: local
            res_conv(v), 33 taps         16 halo rows of v from each neighbour
  backward  dw2 (sum over tokens)        all-reduce [8, 256, 64]; dz, da3v and the pseudo-inverse backward replicated
            dk, dv, dq                   local (the kernels take the GLOBAL a3v / lse3: the restriction of the global formula to the shard)
            dq~, dk~                     token-side terms are partial sums: ONE all-reduce of [256, 1024]; the attn2 terms are added after it
            conv backward                16 halo rows of dout; the conv weight gradient is a local partial sum
Parameter gradients leave as LOCAL partial sums (the data-parallel flat-gradient all-reduce adds them up, as for instance-sharded ABMIL).

``sharded_sattention`` is the encoder level (cls token, front padding, the two layers, LayerNorm of the cls row) and the PPEG between the
layers on a band of its token grid per rank (``ShardedPPEGFn``: the rank's own grid rows, three halo rows either side and the cells that
wrap, fetched by ONE all-to-all; rounds 3: an all-gathered replica of the sequence).  The sharded MHIM(TransMIL) train STEP on top of it -
the re-balancing all-to-all from the instance shards into these token blocks, the teacher's cls attention and pseudo score, the trainer
wiring - is sharded_transmil.py.  All exchanges go through ``sharded._Comm`` (RCCL; gloo with host staging in the one-GPU tests).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import nystrom as NY
from . import ops
from .sharded import _Comm

HEADS, DH, INNER, M, KS, HALO = NY.HEADS, NY.DH, NY.INNER, NY.LANDMARKS, NY.CONV_K, NY.CONV_K // 2
_PPEG_HALO = os.environ.get("MHIMX_PPEG_REPLICA", "0") != "1"          # (1: the round-3 form, the PPEG on an all-gathered replica)
_PPEG_BAND_W1 = os.environ.get("MHIMX_PPEG_BAND_W1", "0") == "1"        # (1: the band kernels at world 1 too - tests of the band path on one rank)


def _edge_halos(comm, block, ld=None):
    """The neighbours' boundary rows of a row-sharded [Tr, C] block (``block`` may be a strided view: ``ld`` floats between rows):
    (prev [HALO, C] = the last rows of rank - 1, next [HALO, C] = the first rows of rank + 1), None at the sequence ends.  World 1: no traffic."""
    if comm.world == 1:
        return None, None
    Tr = block.shape[0]
    edges = torch.stack([block[:HALO], block[Tr - HALO:]]).contiguous()              # [2, 16, C]: my first / last rows
    allg = comm.all_gather(edges)                                                      # [W, 2, 16, C]
    prev = allg[comm.rank - 1, 1].contiguous() if comm.rank > 0 else None
    nxt = allg[comm.rank + 1, 0].contiguous() if comm.rank < comm.world - 1 else None
    return prev, nxt


def _resconv_sharded(x, wc, out, accumulate, flip, prev, nxt):
    """out (+)= res_conv(x) for this rank's rows of a row-sharded sequence (x, out: [Tr, 512], row-strided views welcome; mhimx_resconv pads
    with zeros outside its rows): ONE launch on the block where it lies - no halo-extended copy of the block (round 4 built a [Tr + 32, C]
    copy per call, ~75 us at c3 size) - plus, where a neighbour exists, what ITS 16 boundary rows add to my first / last 16 outputs: the same
    stencil on a 48-row buffer [halo | zeros] (or [zeros | halo]) whose middle 16 outputs are exactly those terms."""
    lib = L.lib()
    Tr = x.shape[0]
    assert x.stride(1) == 1 and out.stride(1) == 1
    L.check(lib.mhimx_resconv(NY._st(), NY._ptr(x), x.stride(0), NY._ptr(wc), KS, DH, Tr, INNER, NY._ptr(out), out.stride(0), int(accumulate),
                              int(flip)), "resconv")
    for halo, first in ((prev, True), (nxt, False)):
        if halo is None:
            continue
        small = torch.zeros((3 * HALO, INNER), device=halo.device)
        (small[:HALO] if first else small[2 * HALO:]).copy_(halo)
        part = torch.empty((3 * HALO, INNER), device=halo.device)
        L.check(lib.mhimx_resconv(NY._st(), NY._ptr(small), INNER, NY._ptr(wc), KS, DH, 3 * HALO, INNER, NY._ptr(part), INNER, 0, int(flip)), "resconv")
        (out[:HALO] if first else out[Tr - HALO:]).add_(part[HALO:2 * HALO])          # the outputs at the 16 rows next to the halo


def _resconv_dw_sharded(dout, v, prev_v, next_v):
    """d(conv weight) [8, 33] of this rank's OUTPUT rows (a local partial sum): the block's own (dout, v) pairs in one launch on the rows where
    they lie, plus the pairs of my first / last 16 outputs with the neighbour's halo rows of v (48-row buffers, as _resconv_sharded)."""
    lib = L.lib()
    dev, Tr = dout.device, dout.shape[0]
    dwc = torch.empty((HEADS, KS), device=dev)
    ws = torch.empty(lib.mhimx_resconv_dw_ws_floats(Tr, INNER, DH, KS), device=dev)
    L.check(lib.mhimx_resconv_dw(NY._st(), NY._ptr(dout), dout.stride(0), NY._ptr(v), v.stride(0), KS, DH, Tr, INNER, NY._ptr(dwc), NY._ptr(ws)),
            "resconv_dw")
    for halo, first in ((prev_v, True), (next_v, False)):
        if halo is None:
            continue
        vs, gs = torch.zeros((3 * HALO, INNER), device=dev), torch.zeros((3 * HALO, INNER), device=dev)
        (vs[:HALO] if first else vs[2 * HALO:]).copy_(halo)
        gs[HALO:2 * HALO].copy_(dout[:HALO] if first else dout[Tr - HALO:])
        part = torch.empty((HEADS, KS), device=dev)
        ws2 = torch.empty(lib.mhimx_resconv_dw_ws_floats(3 * HALO, INNER, DH, KS), device=dev)
        L.check(lib.mhimx_resconv_dw(NY._st(), NY._ptr(gs), INNER, NY._ptr(vs), INNER, KS, DH, 3 * HALO, INNER, NY._ptr(part), NY._ptr(ws2)), "resconv_dw")
        dwc += part
    return dwc


class ShardedTransLayerFn(torch.autograd.Function):
    """y_r = x_r + dropout(to_out(Nystrom(to_qkv(LayerNorm(x)))))_r for this rank's token block.  drop_p / seed: to_out's dropout
    (nystrom_attention.py:60-63) from the counter stream keyed by (seed, LOCAL element index) - give every rank its own seed.  need_attn:
    also returns the cls token's attention row over THIS block's tokens [8, Tr] (nystrom:143-150) and the block's v rows [Tr, 512]."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_qkv, w_out, b_out, conv_w, scale, comm, pad=0, drop_p=0.0, seed=0, need_attn=False):
        lib = L.lib()
        x = x.contiguous()
        Tr, E = x.shape
        W, dev = comm.world, x.device
        T = Tr * W
        npad = max(0, min(int(pad) - comm.rank * Tr, Tr))            # leading rows of THIS block that are the front zero padding
        if T % M or M % W or Tr % 64 or Tr % (T // M):
            raise L.MhimxError("sharded TransLayer: T % 256 == 0, world | 256 and aligned shards (T / world a multiple of 64 and of T / 256)")
        l, gl = T // M, M // W
        xn = torch.empty_like(x)
        mean, rstd = torch.empty(Tr, device=dev), torch.empty(Tr, device=dev)
        L.check(lib.mhimx_layernorm_fwd(NY._st(), NY._ptr(x), Tr, E, NY._ptr(ln_w), NY._ptr(ln_b), NY._ptr(xn), NY._ptr(mean), NY._ptr(rstd)),
                "layernorm_fwd")
        if npad:
            xn[:npad].zero_()                        # nystrom_attention.py:70-73 pads AFTER the LayerNorm: pad tokens are zero rows of q, k, v
        big = Tr >= 2048 and NY._PREC == "bf16x3"                                       # (the projection / weight-gradient kernels, as TransLayerFn)
        if big:
            qkv = torch.empty((Tr, 3 * INNER), device=dev)
            ops.bag_project(xn, [ops.ProjHead(ops.pair_planes(w_qkv), None, out=qkv)], act=0)
        else:
            qkv = ops.gemm_nt(xn, w_qkv, prec=NY._PREC)                                 # [Tr, 1536]
        ld = qkv.shape[1]
        lm_loc = torch.empty((gl, 2 * INNER), device=dev)
        L.check(lib.mhimx_landmark_mean(NY._st(), NY._ptr(qkv), ld, Tr, l, 2 * INNER, NY._ptr(lm_loc)), "landmark_mean")
        lm = comm.all_gather(lm_loc).reshape(M, 2 * INNER).contiguous()                 # rank order = group order
        a2, z, z0, stats, chain = NY._landmark_pinv_forward(lm, scale)                  # replicated
        no = ops.NysOperands(qkv, lm, scale)
        a3v_r, lse_r = ops.nys_a3v_fwd(no)
        if W == 1:                                                                      # (the merge below with one term: the same bits)
            a3v, lse3 = a3v_r, lse_r
        else:
            A, Ls = comm.all_gather(a3v_r), comm.all_gather(lse_r)                      # [W,8,256,64], [W,8,256]
            mx = Ls.max(0).values
            lse3 = (mx + torch.log2(torch.exp2(Ls - mx).sum(0))).contiguous()           # fixed rank order: every rank gets the same bits
            a3v = (torch.exp2(Ls - lse3).unsqueeze(-1) * A).sum(0).contiguous()
        w2 = torch.empty((HEADS, M, DH), device=dev)
        NY._heads_mm("nn", NY.batched(z), NY.batched(a3v), NY.batched(w2), HEADS)
        out, lse1 = ops.nys_out_fwd(no, w2)
        wc = conv_w.reshape(HEADS, -1).contiguous()
        v_prev, v_next = _edge_halos(comm, qkv[:, 2 * INNER:])                         # the neighbours' 16 boundary rows of v (kept for d conv weight)
        _resconv_sharded(qkv[:, 2 * INNER:], wc, out, 1, 0, v_prev, v_next)              # out += res_conv(v) (nystrom:135-136)
        if big:                                                                         # y = x + dropout(to_out(.)): one launch
            y = ops.bag_project(out, [ops.ProjHead(ops.pair_planes(w_out), b_out, drop_p=float(drop_p), drop_seed=int(seed), resid=x)], act=0)[0].out
        else:
            y = ops.gemm_nt(out, w_out, bias=b_out, drop_p=float(drop_p), drop_seed=int(seed), prec=NY._PREC)
            L.check(lib.mhimx_axpby(NY._st(), NY._ptr(x), NY._ptr(y), y.numel(), 1.0, 1.0), "axpby")        # y += x
        ctx.saved = (x, xn, mean, rstd, ln_w, w_qkv, w_out, out, qkv, lm, a2, z, z0, stats, chain, a3v, w2, wc, lse1, lse3, no.ws, v_prev, v_next)
        ctx.cfg = (l, gl, scale, comm, conv_w.shape, npad, float(drop_p), int(seed), big)
        if not need_attn:
            return y
        # the cls token's attention row (nystrom:143-150) = (attn1[cls] pinv) attn3: the owner of the cls row forms u = softmax(q_cls k~^T) z
        # [8, 256], everybody gets it (a sum against zeros), and attn3's columns of the local tokens need the global lse3 only
        g0 = comm.rank * Tr
        u = torch.zeros((HEADS, M), device=dev)
        if g0 <= pad < g0 + Tr:
            a1c = torch.empty((HEADS, 1, M), device=dev)
            NY._heads_mm("nt", NY.Op(qkv, (pad - g0) * ld, DH, ld, 1, DH), NY.Op(lm, INNER, DH, 2 * INNER, M, DH), NY.batched(a1c), HEADS)
            L.check(lib.mhimx_softmax_rows(NY._st(), NY._ptr(a1c), NY._ptr(a1c), HEADS, M, float(scale)), "softmax_rows")
            u3 = torch.empty((HEADS, 1, M), device=dev)
            NY._heads_mm("nn", NY.batched(a1c), NY.batched(z), NY.batched(u3), HEADS)
            u.copy_(u3.view(HEADS, M))
        comm.all_reduce_sum(u)
        attn = ops.nys_cls_attn(ops.NysOperands(qkv, lm, scale), lse3, u.contiguous())  # [8, Tr] (a workspace of its own: the backward keeps no.ws)
        v = qkv[:, 2 * INNER:]
        ctx.mark_non_differentiable(attn, v)
        return y, attn, v

    @staticmethod
    def backward(ctx, dy, *_):
        lib = L.lib()
        x, xn, mean, rstd, ln_w, w_qkv, w_out, out, qkv, lm, a2, z, z0, stats, chain, a3v, w2, wc, lse1, lse3, nws, v_prev, v_next = ctx.saved
        ctx.saved = None
        l, gl, scale, comm, wshape, npad, drop_p, seed, big = ctx.cfg
        dy = dy.contiguous()
        Tr, E = x.shape
        dev, ld = x.device, qkv.shape[1]
        g = dy
        if drop_p > 0:                          # the forward's mask again (the stream of the kernel that drew it)
            g = torch.empty_like(dy)
            fn = lib.mhimx_dropout_apply_proj if big else lib.mhimx_dropout_apply
            L.check(fn(NY._st(), NY._ptr(dy), NY._ptr(g), Tr, E, drop_p, seed & 0xFFFFFFFFFFFFFFFF, None), "dropout_apply")
        dout = torch.empty((Tr, INNER), device=dev)
        wg = big and ops.bag_wgrad_ok(x, INNER, Tr)
        if big:
            ops.bag_project(g, [ops.ProjHead(ops.pair_planes_t(w_out), None, out=dout)], act=0)
        else:
            NY._gemm("nn", g, 0, E, w_out, 0, INNER, dout, 0, INNER, Tr, INNER, E)
        if wg:
            dw_out, db_out = ops.bag_wgrad(g, None, out, None, Tr)
        else:
            dw_out = ops.gemm_tn(g, out, splits=8 if Tr >= 4096 else 1, prec=NY._PREC)
            db_out = ops.colsum(g)
        dqkv = torch.empty_like(qkv)
        # residual convolution: dv = flip-conv(dout) with dout's halo rows; its weight gradient from the local outputs against v with halos
        g_prev, g_next = _edge_halos(comm, dout)
        _resconv_sharded(dout, wc, dqkv[:, 2 * INNER:], 0, 1, g_prev, g_next)            # dv, written where it lies in dqkv
        dwc = _resconv_dw_sharded(dout, qkv[:, 2 * INNER:], v_prev, v_next)              # (my outputs against v with the neighbours' halo rows)
        # out = attn1 w2 : dq local, the token-side term of dk~ and dw2 are partial sums over this shard
        no = ops.NysOperands(qkv, lm, scale, ws=nws)
        dlm = torch.empty_like(lm)
        dw2 = ops.nys_out_bwd(no, w2, dout, lse1, dqkv, dlm)
        comm.all_reduce_sum(dw2)
        dz = torch.empty_like(z)
        NY._heads_mm("nt", NY.batched(dw2), NY.batched(a3v), NY.batched(dz), HEADS)
        da3v = torch.empty_like(a3v)
        NY._heads_mm("tn", NY.batched(z), NY.batched(dw2), NY.batched(da3v), HEADS)
        ops.nys_a3v_bwd(no, a3v, da3v, lse3, dqkv, dlm, accumulate_dv=True)              # dk, dv += (local); the token-side term of dq~
        comm.all_reduce_sum(dlm)
        NY._landmark_pinv_backward(lm, scale, a2, z0, stats, chain, dz, dlm)             # + the attn2 terms (replicated: added after the sum)
        own = dlm[comm.rank * gl:(comm.rank + 1) * gl].contiguous()
        L.check(lib.mhimx_landmark_mean_bwd(NY._st(), NY._ptr(own), Tr, l, 2 * INNER, NY._ptr(dqkv), ld, 1), "landmark_mean_bwd")
        dxn = torch.empty_like(x)
        if big:
            ops.bag_project(dqkv, [ops.ProjHead(ops.pair_planes_t(w_qkv), None, out=dxn)], act=0)
        else:
            NY._gemm("nn", dqkv, 0, ld, w_qkv, 0, E, dxn, 0, E, Tr, E, ld)
        if npad:
            dxn[:npad].zero_()                       # the pad rows of xn are constants
        if wg:
            dw_qkv, _ = ops.bag_wgrad(dqkv, None, xn, None, Tr, want_bias=False)
        else:
            dw_qkv = ops.gemm_tn(dqkv, xn, splits=8 if Tr >= 4096 else 1, prec=NY._PREC)
        dx, dlw, dlb = torch.empty_like(x), torch.empty_like(ln_w), torch.empty_like(ln_w)
        wsl = torch.empty(2 * 512 * E, device=dev)
        L.check(lib.mhimx_layernorm_bwd_res(NY._st(), NY._ptr(dxn), NY._ptr(x), Tr, E, NY._ptr(ln_w), NY._ptr(mean), NY._ptr(rstd), NY._ptr(dy),
                                            NY._ptr(dx), NY._ptr(dlw), NY._ptr(dlb), 0, NY._ptr(wsl)), "layernorm_bwd_res")
        return dx, dlw, dlb, dw_qkv, dw_out, db_out, dwc.reshape(wshape), None, None, None, None, None, None


def sharded_trans_layer(layer: "NY.TransLayer", x_local, comm=None, pad=0, drop_p=0.0, seed=0, need_attn=False):
    """``layer`` (nystrom.TransLayer: the reference's parameter names) applied to this rank's token block of a sharded sequence whose
    first ``pad`` rows (global indices) are the front zero padding of nystrom_attention.py:70-73."""
    comm = comm if comm is not None else _Comm()
    a = layer.attn
    return ShardedTransLayerFn.apply(x_local, layer.norm.weight, layer.norm.bias, a.to_qkv.weight, a.to_out[0].weight, a.to_out[0].bias,
                                     a.res_conv.weight, a.scale, comm, pad, drop_p, seed, need_attn)


class _GatherRows(torch.autograd.Function):
    """Every rank's row block -> the whole [T, C] sequence on every rank; backward: the ranks' gradients of the whole sequence are SUMMED
    (each rank differentiates only what it used of the replica) and the own block is returned."""

    @staticmethod
    def forward(ctx, x, comm):
        ctx.comm, ctx.Tr = comm, x.shape[0]
        return comm.all_gather(x.contiguous()).reshape(comm.world * x.shape[0], x.shape[1])

    @staticmethod
    def backward(ctx, dfull):
        comm, Tr = ctx.comm, ctx.Tr
        dfull = comm.all_reduce_sum(dfull.contiguous().clone())
        return dfull[comm.rank * Tr:(comm.rank + 1) * Tr].contiguous(), None


class _PutRow(torch.autograd.Function):
    """x[i] = row, IN PLACE (x: this rank's freshly assembled token block, whose slot i is the cls token's - baseline.py:253-255; round 4
    rebuilt the block with torch.cat: one more pass over 100 MB).  Backward: d row = dy[i]; the slot's own gradient is zero."""

    @staticmethod
    def forward(ctx, x, row, i):
        ctx.mark_dirty(x)
        x[i].copy_(row.reshape(-1))
        ctx.i, ctx.rshape = i, row.shape
        return x

    @staticmethod
    def backward(ctx, dy):
        drow = dy[ctx.i].clone().reshape(ctx.rshape)
        dy[ctx.i].zero_()                                   # (dy is the layer's fresh dx: nobody else reads it)
        return dy, drow, None


class _OwnerRow(torch.autograd.Function):
    """Row ``idx`` (global) of the sharded sequence on EVERY rank (an all-reduce of the owner's row against zeros); the gradient goes back to
    the owner only (the consumers of the row are replicated: every rank holds the same gradient, it counts once)."""

    @staticmethod
    def forward(ctx, x, idx, comm):
        Tr = x.shape[0]
        own = comm.rank * Tr <= idx < (comm.rank + 1) * Tr
        ctx.cfg = (own, idx - comm.rank * Tr, x.shape)
        row = x[idx - comm.rank * Tr:idx - comm.rank * Tr + 1].clone() if own else torch.zeros((1, x.shape[1]), device=x.device)
        return comm.all_reduce_sum(row)

    @staticmethod
    def backward(ctx, drow):
        own, i, shape = ctx.cfg
        dx = torch.zeros(shape, device=drow.device)
        if own:
            dx[i:i + 1].copy_(drow)
        return dx, None, None


# ---------------------------------------------------------------------------------------------------- PPEG with halos
_PPEG_PLANS = {}


class _BandPlan:
    """The band of the PPEG's token grid one rank works on, and the exchange that fills it - from the layout alone (pad, tokens, ranks).
    The sequence is [zeros(pad) | cls | n_tok tokens] in blocks of Tr rows; token j is grid cell j of a side x side grid whose cells
    [n_tok, wrapN) repeat the first tokens (emb_position.py:100-103).  Rank q owns the cells of its tokens [t0, t1) (the owner of the last
    token also answers for the wrap cells' gradient) and needs the grid rows of those cells plus three either side."""

    def __init__(self, pad, n_tok, Tr, comm, dev):
        import numpy as np
        W, me = comm.world, comm.rank
        wrap = C.c_int64(0)
        H = int(L.lib().mhimx_ppeg_side(n_tok, C.byref(wrap)))
        wrapN, skip = int(wrap.value), pad + 1
        self.H, self.wrapN, self.n_tok, self.skip, self.Tr = H, wrapN, n_tok, skip, Tr
        rng = [(min(max(q * Tr - skip, 0), n_tok), min(max((q + 1) * Tr - skip, 0), n_tok)) for q in range(W)]
        last = max(q for q in range(W) if rng[q][1] > rng[q][0])

        def band(q):
            t0, t1 = rng[q]
            if t1 <= t0:
                return None
            t1e = wrapN if q == last else t1
            rA, rB = max(0, t0 // H - 3), min(H, (t1e - 1) // H + 4)
            return t0, t1, t1e, rA * H, (rB - rA) * H

        need = []
        for q in range(W):
            bq = band(q)
            if bq is None:
                need.append(np.zeros(0, dtype=np.int64))
                continue
            cells = bq[3] + np.arange(bq[4], dtype=np.int64)
            need.append(np.where(cells < n_tok, skip + cells, np.where(cells < wrapN, skip + cells - n_tok, -1)))
        self.band = band(me)
        mine = need[me]
        owner = np.where(mine >= 0, mine // Tr, W)

        def runs(pos, src):                                   # maximal runs where band position and source row advance together
            if pos.size == 0:
                return []
            cut = np.flatnonzero((np.diff(pos) != 1) | (np.diff(src) != 1)) + 1
            st = np.concatenate([[0], cut])
            en = np.concatenate([cut, [pos.size]])
            return [(int(pos[a]), int(src[a]), int(b - a)) for a, b in zip(st, en)]

        # the cells this rank holds ITSELF (all of them at world 1; its own tokens and most of the halo otherwise) are block rows in the same
        # order: slice copies instead of a gather into a send buffer and a scatter out of the receive buffer (round 4: 53 + 60 us per fetch at
        # c3 size); cells without a token: zeroed as runs instead of a zero fill of the whole band
        selfp = np.flatnonzero(owner == me)
        self.self_runs = runs(selfp, mine[selfp] - me * Tr)
        nonep = np.flatnonzero(owner == W)
        self.zero_runs = [(a, n_) for a, _, n_ in runs(nonep, nonep)]
        if len(self.self_runs) > 16 or len(self.zero_runs) > 16:                       # (a layout that fragments: the general path for everything)
            self.self_runs, self.zero_runs = None, None
        by_runs = self.self_runs is not None
        send_idx, self.send_counts = [], []
        for q in range(W):
            nl = need[q]
            own = (nl >= 0) & (nl // Tr == me)
            if by_runs and q == me:
                own = np.zeros_like(own)
            send_idx.append(nl[own] - me * Tr)
            self.send_counts.append(int(own.sum()))
        self.send_idx = torch.as_tensor(np.concatenate(send_idx), dtype=torch.int64).to(dev)
        order = np.argsort(owner, kind="stable")
        order = order[(owner[order] < W) & ((owner[order] != me) | (not by_runs))]
        self.place = torch.as_tensor(order, dtype=torch.int64).to(dev)
        self.recv_counts = [int((owner == q).sum()) if not (by_runs and q == me) else 0 for q in range(W)]
        self.last = last
        self.comm = comm

    def fetch(self, block):
        """This rank's band [ncell, C] of cells from the ranks' sequence blocks (zero where the grid has no token)."""
        ncell = self.band[4] if self.band is not None else 0
        if self.self_runs is None:
            got = self.comm.all_to_all_rows(block.index_select(0, self.send_idx), self.send_counts, self.recv_counts)
            out = torch.zeros((ncell, block.shape[1]), device=block.device)
            if got.shape[0]:
                out.index_copy_(0, self.place, got)
            return out
        out = torch.empty((ncell, block.shape[1]), device=block.device)
        for a, n_ in self.zero_runs:
            out[a:a + n_].zero_()
        for a, src, n_ in self.self_runs:
            out[a:a + n_].copy_(block[src:src + n_])
        if self.comm.world > 1:                                                        # (every rank calls it: the counts may be zero here only)
            got = self.comm.all_to_all_rows(block.index_select(0, self.send_idx), self.send_counts, self.recv_counts)
            if got.shape[0]:
                out.index_copy_(0, self.place, got)
        return out


def _band_plan(pad, n_tok, Tr, comm, dev):
    key = (pad, n_tok, Tr, comm.world, comm.rank, str(dev))
    pl = _PPEG_PLANS.get(key)
    if pl is None:
        if len(_PPEG_PLANS) > 64:
            _PPEG_PLANS.clear()
        pl = _PPEG_PLANS[key] = _BandPlan(pad, n_tok, Tr, comm, dev)
    return pl


class ShardedPPEGFn(torch.autograd.Function):
    """baseline.py:265-266 `cat([cls, ppeg(tokens)])` on a sequence sharded in blocks: the 7 x 7 stencil (emb_position.py:92-120) of this
    rank's tokens from a band of the grid - its own grid rows, three halo rows either side and the wrap cells, fetched with ONE all-to-all
    (forward: token rows; backward: their gradients) instead of an all-gathered replica of the sequence.  The gradient of the wrap cells
    (computed where the grid ends) returns to the first tokens through one small all-reduce ((side^2 - n_tok) x C floats).  The rows of the
    block that are not tokens (front padding, the cls row) pass through.  Parameter gradients: local partial sums."""

    @staticmethod
    def forward(ctx, x, w7, w5, w3, b7, b5, b3, comm, pad, n):
        lib = L.lib()
        x = x.contiguous()
        Tr, Cc = x.shape
        pl = _band_plan(int(pad), int(n) - 1, Tr, comm, x.device)
        wc, bc = torch.empty((Cc, 49), device=x.device), torch.empty(Cc, device=x.device)
        L.check(lib.mhimx_ppeg_combine(NY._st(), NY._ptr(w7.contiguous()), NY._ptr(w5.contiguous()), NY._ptr(w3.contiguous()), NY._ptr(b7),
                                       NY._ptr(b5), NY._ptr(b3), Cc, NY._ptr(wc), NY._ptr(bc)), "ppeg_combine")
        xb = pl.fetch(x)
        if pl.band is not None:
            t0, t1, t1e, cell0, ncell = pl.band
            bd = L.PpegBand(H=pl.H, cell0=cell0, ncell=ncell, out0=t0, out1=t1)
            r0 = pl.skip + t0 - comm.rank * Tr                                           # block row of token t0
            y = torch.empty_like(x)
            y[:r0].copy_(x[:r0])                                                         # the rows that are not tokens pass through
            y[r0 + (t1 - t0):].copy_(x[r0 + (t1 - t0):])
            L.check(lib.mhimx_ppeg_band_fwd(NY._st(), NY._ptr(xb), C.byref(bd), Cc, NY._ptr(wc), NY._ptr(bc), NY._ptr(y, r0 * Cc)), "ppeg_band_fwd")
        else:
            y = x.clone()
        ctx.saved = (xb, wc)
        ctx.cfg = (pl, comm, Tr, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        xb, wc = ctx.saved
        ctx.saved = None
        pl, comm, Tr, Cc = ctx.cfg
        dy = dy.contiguous()
        dev = dy.device
        dyb = pl.fetch(dy)
        if pl.band is not None:                                                          # pass-through rows (the token rows are written below)
            r0_, nt_ = pl.skip + pl.band[0] - comm.rank * Tr, pl.band[1] - pl.band[0]
            dx = torch.empty_like(dy)
            dx[:r0_].copy_(dy[:r0_])
            dx[r0_ + nt_:].copy_(dy[r0_ + nt_:])
        else:
            dx = dy.clone()
        add = pl.wrapN - pl.n_tok
        dxw = torch.zeros((add, Cc), device=dev) if add else None
        dwc, dbc = torch.zeros((Cc, 49), device=dev), torch.zeros(Cc, device=dev)
        if pl.band is not None:
            t0, t1, t1e, cell0, ncell = pl.band
            if pl.n_tok - cell0 < ncell:
                dyb[max(pl.n_tok - cell0, 0):].zero_()                                   # the fetch filled the wrap cells with dy of the first tokens: not outputs
            bd = L.PpegBand(H=pl.H, cell0=cell0, ncell=ncell, out0=t0, out1=t1e)
            dxo = torch.empty((t1e - t0, Cc), device=dev)
            ws = torch.empty(lib.mhimx_ppeg_band_bwd_ws_floats(t1e - t0, Cc), device=dev)
            L.check(lib.mhimx_ppeg_band_bwd(NY._st(), NY._ptr(dyb), NY._ptr(xb), C.byref(bd), t1 - t0, Cc, NY._ptr(wc), NY._ptr(dxo), NY._ptr(dwc),
                                            NY._ptr(dbc), NY._ptr(ws)), "ppeg_band_bwd")
            r0 = pl.skip + t0 - comm.rank * Tr
            dx[r0:r0 + (t1 - t0)].copy_(dxo[:t1 - t0])
            if t1e > t1 and add:
                dxw.copy_(dxo[t1 - t0:])
        if add:
            comm.all_reduce_sum(dxw)                                                     # (one contributor: the rank where the grid ends)
            if pl.band is not None and pl.band[0] < add:                                # my tokens among the first `add`
                t0, t1 = pl.band[0], min(pl.band[1], add)
                r0 = pl.skip + t0 - comm.rank * Tr
                seg = dx[r0:r0 + (t1 - t0)]
                L.check(lib.mhimx_axpby(NY._st(), NY._ptr(dxw, t0 * Cc), NY._ptr(seg), seg.numel(), 1.0, 1.0), "axpby")
        g = dwc.view(Cc, 1, 7, 7)
        return (dx, g.contiguous(), g[:, :, 1:6, 1:6].contiguous(), g[:, :, 2:5, 2:5].contiguous(), dbc, dbc.clone(), dbc.clone(), None, None, None)


def sharded_ppeg(pe: "NY._PPEG", x_local, comm, pad, n):
    return ShardedPPEGFn.apply(x_local, pe.proj.weight, pe.proj1.weight, pe.proj2.weight, pe.proj.bias, pe.proj1.bias, pe.proj2.bias, comm, pad, n)


def sharded_sattention(enc: "NY.SAttention", h_local, pad, n, comm=None, return_attn=False, seeds=(0, 0), training=False):
    """mhim_modules/baseline.SAttention (cls token, TransLayer, PPEG, TransLayer, LayerNorm of the cls row: baseline.py:222-288) on a
    sequence sharded over the ranks.  h_local: this rank's block of the PADDED token sequence [zeros(pad) | cls slot | n - 1 token rows]
    (T = pad + n, T % 256 == 0; the cls slot's content is ignored).  Returns the cls feature [512] on every rank.
    First cut of the encoder level: the two layers are sequence-parallel (ShardedTransLayerFn); the PPEG between them runs REPLICATED on
    an all-gathered copy of the sequence (one all-gather forward, one all-reduce of its gradient backward), each rank keeping its rows."""
    comm = comm if comm is not None else _Comm()
    Tr = h_local.shape[0]
    g0 = comm.rank * Tr
    x = h_local
    if g0 <= pad < g0 + Tr:                                                  # the cls token's row lives here (baseline.py:253-255)
        i = pad - g0
        if (h_local.requires_grad and h_local.is_leaf) or not h_local.is_contiguous():
            x = torch.cat([h_local[:i], enc.cls_token.view(1, -1), h_local[i + 1:]], 0)
        else:
            x = _PutRow.apply(h_local, enc.cls_token, i)
    p1 = enc.layer1.attn.dropout if training else 0.0
    p2 = enc.layer2.attn.dropout if training else 0.0
    attn, v = [], None
    if return_attn:
        x, a, v = sharded_trans_layer(enc.layer1, x, comm, pad, p1, seeds[0], True)
        attn.append(a)
    else:
        x = sharded_trans_layer(enc.layer1, x, comm, pad, p1, seeds[0])
    if comm.world == 1 and not _PPEG_BAND_W1:                                # one rank: the band is the whole grid - the replica path's PPEG
        x = enc.pos_embedding(x, skip=pad + 1)                               # on the block itself (rows [0, pad] pass through), no band copy
    elif n - 1 >= 49 and _PPEG_HALO:                                         # baseline.py:265-266: cat([cls, ppeg(tokens)])
        x = sharded_ppeg(enc.pos_embedding, x, comm, pad, n)                 # its own grid rows + halos: one all-to-all (ShardedPPEGFn)
    else:                                                                    # (grids below 7 x 7: the zero-padded rule, on a replica)
        full = _GatherRows.apply(x, comm)
        full = enc.pos_embedding(full, skip=pad + 1)                         # rows [0, pad] pass through; the stencil sees the n - 1 tokens
        x = full[g0:g0 + Tr]
    if return_attn:
        x, a, _ = sharded_trans_layer(enc.layer2, x, comm, pad, p2, seeds[1], True)
        attn.append(a)
    else:
        x = sharded_trans_layer(enc.layer2, x, comm, pad, p2, seeds[1])
    row = _OwnerRow.apply(x, pad, comm)
    # only the cls row is used (baseline.py:276-278).  The final LayerNorm runs replicated: its parameter gradient is counted ONCE, on the
    # rank that owns the cls row (the others see constants), so that the flat-gradient SUM over the ranks is the gradient
    own = g0 <= pad < g0 + Tr
    w, b = (enc.norm.weight, enc.norm.bias) if own else (enc.norm.weight.detach(), enc.norm.bias.detach())
    z = NY.LayerNorm.apply(row, w, b)[0]
    return (z, attn, v) if return_attn else z                      # attn: per layer [8, Tr] over THIS block's rows (global row pad + 1 + j = token j)
