"""Thin tensor-level wrappers over the C-ABI (one Python function per entry point of include/mhimx.h).

PyTorch is only plumbing here: it owns the device memory and the stream.  Every function enqueues on
torch's current HIP stream and returns immediately (no host sync).  Tensors must be fp32 (indices int64),
contiguous and on the GPU; nothing falls back to torch math.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


# bench.py installs a callable (tag, M, N, K) -> (start_event, end_event) | None to bracket one kernel with HIP
# events on the launch stream (torch's current stream IS the stream the kernels are enqueued on).
KERNEL_EVENT_HOOK = None


import threading

_PIN = threading.local()        # .stream (c_void_p) / .device (index): the pinned launch stream of THIS thread's trainer step


def _pinned():
    st = getattr(_PIN, "stream", None)
    if st is not None and getattr(_PIN, "device", -1) == torch.cuda.current_device():
        return st
    return None


def _stream():
    st = _pinned()
    if st is not None:
        return st
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class pinned_stream:
    """``with ops.pinned_stream():`` - every launch inside goes to the stream that is current at entry, looked up ONCE (a step's ~20 launches
    each asked torch for it: ~4 us of host time apiece, and the eager step is host-bound).  Only around code that does not switch streams.
    The pin belongs to the calling THREAD and to the device that is current at entry: another thread, or a trainer on another device, never
    sees it (and launches on its own current stream)."""

    def __enter__(self):
        self._old = (getattr(_PIN, "stream", None), getattr(_PIN, "device", -1))
        _PIN.device = torch.cuda.current_device()
        _PIN.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return self

    def __exit__(self, *a):
        _PIN.stream, _PIN.device = self._old


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None or _pinned() is not None:           # (inside a trainer's pinned step every tensor is the trainer's own: the bag, the label and
                                                      #  injected draws were checked at entry, FusedTrainer._forward_backward_nat)
        return
    if not t.is_cuda:
        raise L.MhimxError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise L.MhimxError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.MhimxError(f"{name}: expected a contiguous tensor")


def prec_code(prec) -> int:
    return L.PREC[prec] if isinstance(prec, str) else int(prec)


# ------------------------------------------------------------------------------------------------ GEMMs
def gemm_nt(a, b, out=None, rows=None, bias=None, act=0, pre=None, rowv=None, colv=None, drop_p=0.0, drop_seed=0,
            drop_mask=None, accumulate=False, prec="f16s", M=None, drop_tick=None, paired=False, dact=None):
    """out[m,n] = epi(sum_k a[rows[m] or m, k] * b[n,k]) — see mhimx_gemm_nt."""
    for t, nm in ((a, "a"), (b, "b"), (bias, "bias"), (pre, "pre"), (rowv, "rowv"), (colv, "colv"), (out, "out")):
        _chk(t, name=nm)
    _chk(rows, torch.int64, "rows")
    _chk(drop_mask, torch.uint8, "drop_mask")
    K = a.shape[-1]
    N = b.shape[0]
    if M is None:
        M = rows.shape[0] if rows is not None else a.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    g = L.GemmNT(A=_p(a), lda=a.stride(0) if a.dim() == 2 else K, rows=_p(rows), B=_p(b), ldb=b.stride(0), C=_p(out),
                 ldc=out.stride(0), M=M, N=N, K=K, bias=_p(bias), rowv=_p(rowv), colv=_p(colv), pre=_p(pre),
                 ldpre=pre.stride(0) if pre is not None else 0, act=int(act), drop_p=float(drop_p),
                 drop_seed=int(drop_seed) & 0xFFFFFFFFFFFFFFFF, drop_mask=_p(drop_mask), accumulate=int(bool(accumulate)),
                 prec=prec_code(prec), drop_tick=_p(drop_tick), paired=int(bool(paired)), dact=_p(dact),
                 lddact=dact.stride(0) if dact is not None else 0)
    evs = KERNEL_EVENT_HOOK("gemm_nt", M, N, K) if KERNEL_EVENT_HOOK is not None else None
    if evs:
        evs[0].record()
    L.check(L.lib().mhimx_gemm_nt(_stream(), C.byref(g)), "mhimx_gemm_nt")
    if evs:
        evs[1].record()
    return out


class ProjHead:
    """One model of a bag projection (ops.bag_project): paired-plane weight image, bias, dropout stream, outputs."""

    def __init__(self, wp, bias=None, drop_p=0.0, drop_seed=0, drop_mask=None, out=None, want_dact=False, dact=None, resid=None):
        self.wp, self.bias, self.drop_p, self.drop_seed, self.drop_mask = wp, bias, float(drop_p), int(drop_seed), drop_mask
        self.out, self.want_dact, self.dact, self.resid = out, want_dact, dact, resid


class ProjScoreBuf:
    """The teacher's scorer inside the projection (mhimx_proj_score): the [Wa ; Wp] image, the second scorer layer, and the buffers the
    launch fills - s [N], cproj [N, C], the G pool partials - plus the pair-exchange scratch and its counters (allocated once per N: the
    counters are zeroed once and only count up).  ``finalize`` merges the partials (mhimx_pool_finalize)."""

    _scratch = {}

    def __init__(self, N, wa16, wc, act, C_classes, device):
        lib = L.lib()
        self.N, self.C_classes, self.act = int(N), int(C_classes), int(act)
        self.wa16, self.wc = wa16, wc
        G = self.G = lib.mhimx_proj_score_parts(self.N)
        key = (self.N, str(device))
        sc = ProjScoreBuf._scratch.get(key)
        if sc is None:
            sc = ProjScoreBuf._scratch[key] = (torch.empty(lib.mhimx_proj_score_xch_floats(self.N), device=device),
                                               torch.zeros(G, dtype=torch.int32, device=device))
        self.xch, self.gate = sc
        E = 512
        self.s = torch.empty(self.N, device=device)
        self.cproj = torch.empty((self.N, max(self.C_classes, 1)), device=device) if self.C_classes else None
        self.pm, self.pl, self.pz = torch.empty(G, device=device), torch.empty(G, device=device), torch.empty((G, E), device=device)
        self.stats, self.z, self.pscore = torch.empty(2, device=device), torch.empty(E, device=device), None
        self.c = L.ProjScore(wa16=_p(wa16), wc=_p(wc), act=self.act, C=self.C_classes, s=_p(self.s), cproj=_p(self.cproj), pm=_p(self.pm),
                             pl=_p(self.pl), pz=_p(self.pz), xch=_p(self.xch), gate=_p(self.gate))

    def finalize(self, bp=None):
        """-> self (stats, z and, with the predictor bias ``bp``, pscore [N] filled)."""
        if bp is not None:
            self.pscore = torch.empty(self.N, device=self.s.device)
        L.check(L.lib().mhimx_pool_finalize(_stream(), _p(self.pm), _p(self.pl), _p(self.pz), self.G, 512, _p(self.stats), _p(self.z), _p(self.s),
                                            _p(self.cproj), _p(bp), self.C_classes, self.N, _p(self.pscore)), "mhimx_pool_finalize")
        return self


def _project_args(x, heads, act, drop_tick, extra_rows):
    _chk(x, name="x")
    N, D = x.shape
    E = heads[0].wp.shape[0]
    a = L.BagProject(X=_p(x), ldx=x.stride(0), N=N, D=D, E=E, act=int(act), n_heads=len(heads), drop_tick=_p(drop_tick))
    for i, h in enumerate(heads):
        _chk(h.wp, name="wp"); _chk(h.bias, name="bias"); _chk(h.drop_mask, torch.uint8, "drop_mask"); _chk(h.resid, name="resid")
        if h.out is None:
            h.out = torch.empty((N + extra_rows, E), device=x.device)
        _chk(h.out, name="out")
        if h.want_dact and h.dact is None:
            h.dact = torch.empty((N, E), device=x.device, dtype=torch.float16)
        a.head[i] = L.ProjHead(wp=_p(h.wp), bias=_p(h.bias), H=_p(h.out), ldh=h.out.stride(0), dact=_p(h.dact),
                               drop_p=h.drop_p, drop_seed=h.drop_seed & 0xFFFFFFFFFFFFFFFF, drop_mask=_p(h.drop_mask),
                               resid=_p(h.resid), ldr=h.resid.stride(0) if h.resid is not None else 0)
    return a


def bag_project_multi(xs, heads_per_bag, act=0, drop_tick=None, extra_rows=0):
    """bag_project for the bags of an accumulation window in ONE launch (mhimx_bag_project_multi): xs[b] with heads_per_bag[b] (the same
    models' ProjHeads: shared weight images / bias / drop_p, per-bag outputs and seeds)."""
    arr = (L.BagProject * len(xs))(*[_project_args(x, hs, act, drop_tick, extra_rows) for x, hs in zip(xs, heads_per_bag)])
    L.check(L.lib().mhimx_bag_project_multi(_stream(), arr, len(xs)), "mhimx_bag_project_multi")
    return heads_per_bag


def bag_project(x, heads, act=0, drop_tick=None, extra_rows=0, score0=None, keep_rows0=False):
    """Every model's feature rows H_g = dropout_g(act(x W_g^T + b_g)) in ONE pass over the fp32 bag x [N,D] (mhimx_bag_project).
    heads: 1 or 2 ProjHead (teacher, student).  Fills head.out [N + extra_rows, E] fp32 (the extra rows are left for the caller:
    merged tokens) and, with want_dact, head.dact [N, E] fp16 (d out / d pre).  Returns the heads.
    ``score0`` (ProjScoreBuf): model 0's scorer and pool partials are computed in the launch's epilogue and its feature rows are NOT
    written (heads[0].out stays None) unless ``keep_rows0`` (tests)."""
    _chk(x, name="x")
    N, D = x.shape
    E = heads[0].wp.shape[0]
    a = L.BagProject(X=_p(x), ldx=x.stride(0), N=N, D=D, E=E, act=int(act), n_heads=len(heads), drop_tick=_p(drop_tick),
                     score0=None if score0 is None else C.cast(C.pointer(score0.c), C.c_void_p))
    for i, h in enumerate(heads):
        _chk(h.wp, name="wp"); _chk(h.bias, name="bias"); _chk(h.drop_mask, torch.uint8, "drop_mask"); _chk(h.resid, name="resid")
        if h.out is None and not (i == 0 and score0 is not None and not keep_rows0):
            h.out = torch.empty((N + extra_rows, E), device=x.device)
        _chk(h.out, name="out")
        if h.want_dact and h.dact is None:
            h.dact = torch.empty((N, E), device=x.device, dtype=torch.float16)
        a.head[i] = L.ProjHead(wp=_p(h.wp), bias=_p(h.bias), H=_p(h.out), ldh=E if h.out is None else h.out.stride(0), dact=_p(h.dact),
                               drop_p=h.drop_p, drop_seed=h.drop_seed & 0xFFFFFFFFFFFFFFFF, drop_mask=_p(h.drop_mask),
                               resid=_p(h.resid), ldr=h.resid.stride(0) if h.resid is not None else 0)
    evs = KERNEL_EVENT_HOOK("bag_project", N, E * len(heads), D) if KERNEL_EVENT_HOOK is not None else None
    if evs:
        evs[0].record()
    L.check(L.lib().mhimx_bag_project(_stream(), C.byref(a)), "mhimx_bag_project")
    if evs:
        evs[1].record()
    return heads


PREP_TRANSPOSE, PREP_PAIR, PREP_COPY, PREP_TICK, PREP_FRAG, PREP_FRAG_T, PREP_MERGE, PREP_PAIR_T, PREP_FRAG16, PREP_XIMG = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9


def _prep_array(jobs):
    """The C job array of a (kind, in_tensor | None, out_tensor) list (mhimx_prep_job)."""
    arr = (L.PrepJob * len(jobs))()
    for i, (kind, src, dst) in enumerate(jobs):
        if kind == PREP_TICK:
            arr[i] = L.PrepJob(kind, None, _p(dst), 1, 1)
        elif kind == PREP_MERGE:                  # (MergeW, (workspace uint8 tensor, rows to merge)): the struct is read at enqueue time
            ws, n_rows = dst
            arr[i] = L.PrepJob(kind, C.c_void_p(C.addressof(src.c)), _p(ws), int(n_rows), ws.numel())
        else:
            _chk(src, name="prep in"); _chk(dst, name="prep out")
            R, Cc = (src.shape[0], src.numel() // src.shape[0]) if src.dim() >= 2 else (1, src.numel())
            arr[i] = L.PrepJob(kind, _p(src), _p(dst), R, Cc)
    return arr


def prep_batch(jobs):
    """jobs: list of (kind, in_tensor | None, out_tensor) -> ONE launch.  kind: PREP_TRANSPOSE (out = in^T), PREP_PAIR
    (paired bf16 planes), PREP_COPY, PREP_TICK (out: int64/uint64 [1] counter += 1), PREP_FRAG (matrix-core fragment image
    of a [32a, 16b] weight, see mhimx.h)."""
    arr = _prep_array(jobs)
    L.check(L.lib().mhimx_prep_batch(_stream(), arr, len(jobs)), "mhimx_prep_batch")


# Weight images of ONE train step made ahead by the trainer's preparation launch (FusedTrainer: prep_batch jobs PREP_PAIR /
# PREP_PAIR_T), keyed by (storage address, transposed).  pair_planes / pair_planes_t hand them out instead of launching; the trainer
# clears the table when the step's backward is done - the weights change then.
_STEP_IMAGES = {}


def step_images(table=None):
    """Install (or, with None, drop) the table {(data_ptr, transposed): image} of this step's prepared weight images."""
    _STEP_IMAGES.clear()
    if table:
        _STEP_IMAGES.update(table)


def pair_planes(x):
    """x [M,K] fp32 -> its paired-plane image (same shape; 8 bf16 hi | 8 bf16 lo per 8 consecutive k) for gemm_nt(paired=True)."""
    _chk(x, name="x")
    img = _STEP_IMAGES.get((x.data_ptr(), False))
    if img is not None and img.shape == x.shape:
        return img
    M, K = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().mhimx_pair_planes(_stream(), _p(x), x.stride(0), M, K, _p(out)), "mhimx_pair_planes")
    return out


def pair_planes_t(x):
    """The paired-plane image of x^T ([K, M]) - the weight operand of dX = dY W on the projection kernel."""
    _chk(x, name="x")
    img = _STEP_IMAGES.get((x.data_ptr(), True))
    if img is not None and img.shape == (x.shape[1], x.shape[0]):
        return img
    return pair_planes(transpose(x))


class ReduceList:
    """Deferred final reductions of one step (mhimx_reduce_list): producers given ``defer=lst`` queue the last stage of their
    gradient reductions instead of launching it; ``reduce_flush(lst)`` runs all of them in one launch.  The object keeps
    the partial buffers alive until the flush (a captured graph's allocator would otherwise hand them to a later tensor)."""

    def __init__(self):
        self.c = L.ReduceListC()
        self.c.n = 0
        self.c.side.pending = 0
        self.c.parked.pending = 0
        self.c.pre.pending = 0
        self.keep = []

    def ptr(self):
        return C.addressof(self.c)


def _dp(defer):
    return None if defer is None else defer.ptr()


def reduce_flush(lst: "ReduceList"):
    L.check(L.lib().mhimx_reduce_flush(_stream(), lst.ptr()), "mhimx_reduce_flush")
    lst.keep.clear()


def reduce_slabs_job(lst: "ReduceList", slabs, out, accumulate=True):
    """Queue ``out (+)= sum_z slabs[z]`` (slabs [G, n] contiguous, out [n]) on a reduce list: a kind-1 job of mhimx_reduce_flush
    (the public mhimx_reduce_list struct: any caller may append jobs)."""
    _chk(slabs, name="slabs"); _chk(out, name="out")
    G, n = slabs.shape
    if lst.c.n >= L.REDUCE_MAX:
        raise L.MhimxError("reduce list full")
    lst.c.j[lst.c.n] = L.ReduceJob(kind=1, accumulate=int(bool(accumulate)), parts=_p(slabs), out=_p(out), G=G, W=0, ld=0, K1=1, K2=n, ldo=n)
    lst.c.n += 1
    lst.keep.append(slabs)


def gemm_tn(a, b, out=None, rows=None, splits=1, accumulate=False, prec="bf16x3", M=None, defer=None):
    """out[i,j] = sum_m a[m,i] * b[rows[m] or m, j] — see mhimx_gemm_tn."""
    _chk(a, name="a"); _chk(b, name="b"); _chk(out, name="out"); _chk(rows, torch.int64, "rows")
    if M is None:
        M = a.shape[0]
    K1, K2 = a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((K1, K2), device=a.device, dtype=torch.float32)
    nslab = max(int(splits), 1)
    if splits > 1:                       # room for the library to split the reduction further (fills the chip)
        nslab = max(nslab, min(64, -(-512 // max(1, (K1 // 128) * (K2 // 128)))), -(-M // 4096))
    ws = torch.empty((nslab, K1, K2), device=a.device, dtype=torch.float32) if nslab > 1 else None
    g = L.GemmTN(A=_p(a), lda=a.stride(0), B=_p(b), ldb=b.stride(0), rows=_p(rows), C=_p(out), ldc=out.stride(0), M=M,
                 K1=K1, K2=K2, splits=int(splits), ws=_p(ws), accumulate=int(bool(accumulate)), prec=prec_code(prec),
                 ws_floats=0 if ws is None else ws.numel(), defer=_dp(defer))
    L.check(L.lib().mhimx_gemm_tn(_stream(), C.byref(g)), "mhimx_gemm_tn")
    if defer is not None:
        defer.keep.append(ws)
    return out


def transpose(x, out=None):
    _chk(x, name="x")
    R, Cc = x.shape
    if out is None:
        out = torch.empty((Cc, R), device=x.device, dtype=torch.float32)
    L.check(L.lib().mhimx_transpose(_stream(), _p(x), _p(out), R, Cc), "mhimx_transpose")
    return out


# ------------------------------------------------------------------------------------------------ pool
class ScorerW:
    """Scorer weights in the C layout (keeps the tensors alive)."""

    def __init__(self, wa, wc, act, ba=None, wb=None, bb=None, bc=None, prec="bf16x3", wa_frag=None, gate_drop_p=0.0, gate_drop_seed=0):
        self.t = (wa, wc, ba, wb, bb, bc, wa_frag)
        for t in self.t:
            _chk(t, name="scorer weight")
        self.A, self.E = wa.shape
        self.gated = wb is not None
        self.c = L.Scorer(E=self.E, A=self.A, act=int(act), gated=int(self.gated), prec=prec_code(prec), wa=_p(wa),
                          ba=_p(ba), wb=_p(wb), bb=_p(bb), wc=_p(wc), bc=_p(bc), wa_frag=_p(wa_frag),
                          gate_drop_p=float(gate_drop_p), gate_drop_seed=int(gate_drop_seed) & 0xFFFFFFFFFFFFFFFF)


class PoolState:
    """Buffers of one pool forward (kept for the backward)."""

    def __init__(self, T1, T2, C_classes=0, wp=None, device=None, bp=None, rows1=None, excl=None):
        dev = T1.device
        M1 = T1.shape[0] if rows1 is None else rows1.shape[0]
        self.rows1 = rows1
        self.excl = excl
        M2 = 0 if T2 is None else T2.shape[0]
        self.T1, self.T2, self.M1, self.M2 = T1, T2, M1, M2
        M = M1 + M2
        self.s = torch.empty(M, device=dev)
        self.stats = torch.empty(2, device=dev)
        self.z = torch.empty(T1.shape[1], device=dev)
        self.cproj = torch.empty((M, C_classes), device=dev) if wp is not None else None
        self.wp = wp
        self.bp = bp
        self.pscore = torch.empty(M1, device=dev) if (wp is not None and bp is not None) else None
        self.ws = None
        self.ride = None                     # (C job array, count): preparation jobs riding in the forward's scorer launch

    def io(self, sc: ScorerW):
        M = self.M1 + self.M2
        nbytes = L.lib().mhimx_abmil_pool_ws_bytes(M, sc.E, sc.A, int(sc.gated))
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = torch.empty(nbytes, device=self.T1.device, dtype=torch.uint8)
        return L.PoolIO(T1=_p(self.T1), M1=self.M1, T2=_p(self.T2), M2=self.M2, s=_p(self.s), stats=_p(self.stats),
                        z=_p(self.z), u_pre=None, wp=_p(self.wp), C=0 if self.cproj is None else self.cproj.shape[1],
                        cproj=_p(self.cproj), ws=_p(self.ws), ws_bytes=self.ws.numel(), bp=_p(self.bp), pscore=_p(self.pscore),
                        rows1=_p(self.rows1), excl=_p(self.excl),
                        ride_jobs=None if self.ride is None else C.cast(self.ride[0], C.c_void_p), n_ride_jobs=0 if self.ride is None else self.ride[1])


def abmil_pool_fwd_split(sc: ScorerW, T1, rows1, tail_tokens, ride_merge=None):
    """The pool forward in two calls around the launches that produce the last ``tail_tokens`` tokens of the list (mhimx_pool_io.phase):
    this is phase 1 - the one-pass scorer launch over the first len(rows1) - tail_tokens tokens, with, for ``ride_merge`` = (MergeW, X, ws),
    the row tiles of that Merge forward riding at its front.  Returns (PoolState, rode): when ``rode``, call merge_fwd(rows_done=True);
    then abmil_pool_fwd_finish(sc, state)."""
    _chk(T1, name="T1"); _chk(rows1, torch.int64, "rows1")
    st = PoolState(T1, None, 0, None, rows1=rows1)
    st.tail_tokens = int(tail_tokens)
    io = st.io(sc)
    io.phase, io.tail_tokens = 1, st.tail_tokens
    if ride_merge is not None:
        mw, X, ws = ride_merge
        io.ride_merge, io.ride_X = C.cast(C.pointer(mw.c), C.c_void_p), _p(X)
        io.ride_R, io.ride_ws, io.ride_ws_bytes = mw.x_rows.shape[0], _p(ws), ws.numel()
    L.check(L.lib().mhimx_abmil_pool_fwd(_stream(), C.byref(sc.c), C.byref(io)), "mhimx_abmil_pool_fwd (phase 1)")
    return st, bool(io.rode_merge)


def abmil_pool_fwd_finish(sc: ScorerW, st: PoolState, wa_t=None, tail_row0=-1):
    """Phase 2 of abmil_pool_fwd_split: the finalize launch scores the tail tokens and merges them with the partials -> st.stats, st.z.
    ``wa_t``: the transposed scorer weight [E, A] if at hand; ``tail_row0`` >= 0: the tail tokens are the rows tail_row0.. of T1."""
    io = st.io(sc)
    io.phase, io.tail_tokens = 2, st.tail_tokens
    io.tail_wa_t, io.tail_row0 = _p(wa_t), int(tail_row0)
    L.check(L.lib().mhimx_abmil_pool_fwd(_stream(), C.byref(sc.c), C.byref(io)), "mhimx_abmil_pool_fwd (phase 2)")
    return st


def abmil_pool_fwd(sc: ScorerW, T1, T2=None, wp=None, bp=None, rows1=None, excl=None, ride_jobs=None, no_backward=False):
    """Scorer + softmax pool over tokens [T1; T2] -> PoolState (s, stats, z, cproj; with ``bp`` also ``pscore`` [M1], the
    pseudo score of the T1 instances, written by the pool's finalize launch).  ``rows1`` (int64): the tokens are T1[rows1].
    ``excl`` (uint8, by source row): rows that do not take part (score -inf; see mhimx_pool_io.excl).  ``ride_jobs``: prep_batch
    jobs that run as extra workgroups of this forward's scorer launch (mhimx_pool_io.ride_jobs).  ``no_backward``: a teacher's forward -
    the one-pass scorer does not store its pre-activations (mhimx_pool_io.no_backward); abmil_pool_bwd must not follow."""
    _chk(T1, name="T1"); _chk(T2, name="T2"); _chk(wp, name="wp"); _chk(bp, name="bp"); _chk(rows1, torch.int64, "rows1")
    _chk(excl, torch.uint8, "excl")
    st = PoolState(T1, T2, 0 if wp is None else wp.shape[0], wp, bp=bp, rows1=rows1, excl=excl)
    if ride_jobs:
        st.ride = (_prep_array(ride_jobs), len(ride_jobs))
    io = st.io(sc)
    io.no_backward = int(bool(no_backward))
    L.check(L.lib().mhimx_abmil_pool_fwd(_stream(), C.byref(sc.c), C.byref(io)), "mhimx_abmil_pool_fwd")
    st.ride = None                           # (the jobs ran with the forward; the backward's io carries none)
    return st


def abmil_pool_bwd(sc: ScorerW, st: PoolState, g_z, wa_t, wb_t=None, need_bias=False, splits=8, grads=None,
                   accumulate=False, defer=None, wa_t_frag=None, img=None, img_dact=None, img_part=None, img_rows=0):
    """Backward of the pool.  Returns dict(dT1, dT2, d_wa, d_wc, [d_wb, d_ba, d_bb, d_bc]).
    ``img`` (round 6, one-pass backward): the first ``img_rows`` tokens' gradient leaves as their part of the projection's dPRE image
    (dT * img_dact[row], rows_dpre_image's format; tile t of the launch = k-step t) instead of as rows of dT1; ``img_part`` [tiles, E]
    takes the per-tile column sums (mhimx_pool_grad.img)."""
    dev = g_z.device
    E, A = sc.E, sc.A
    out = grads or {}
    out.setdefault("dT1", torch.empty((st.M1, E), device=dev))
    if st.M2:
        out.setdefault("dT2", torch.empty((st.M2, E), device=dev))
    out.setdefault("d_wa", torch.empty((A, E), device=dev))
    out.setdefault("d_wc", torch.empty((1, A), device=dev))
    if sc.gated:
        out.setdefault("d_wb", torch.empty((A, E), device=dev))
    if need_bias:
        out.setdefault("d_ba", torch.empty(A, device=dev))
        out.setdefault("d_bc", torch.empty(1, device=dev))
        if sc.gated:
            out.setdefault("d_bb", torch.empty(A, device=dev))
    io = st.io(sc)
    g = L.PoolGrad(g_z=_p(g_z), dT1=_p(out["dT1"]), dT2=_p(out.get("dT2")), d_wa=_p(out["d_wa"]), d_ba=_p(out.get("d_ba")),
                   d_wb=_p(out.get("d_wb")), d_bb=_p(out.get("d_bb")), d_wc=_p(out["d_wc"]), d_bc=_p(out.get("d_bc")),
                   wa_t=_p(wa_t), wb_t=_p(wb_t), accumulate=int(bool(accumulate)), splits=int(splits), defer=_dp(defer),
                   wa_t_frag=_p(wa_t_frag), img=_p(img), img_dact=_p(img_dact), img_part=_p(img_part), img_rows=int(img_rows))
    if defer is not None:
        defer.keep.append(st)
    L.check(L.lib().mhimx_abmil_pool_bwd(_stream(), C.byref(sc.c), C.byref(io), C.byref(g)), "mhimx_abmil_pool_bwd")
    return out


def softmax_from_stats(s, stats):
    out = torch.empty_like(s)
    L.check(L.lib().mhimx_softmax_from_stats(_stream(), _p(s), _p(stats), _p(out), s.numel()), "mhimx_softmax_from_stats")
    return out


def pseudo_score(s, stats, cproj, bp, want_attn=False):
    M = cproj.shape[0] if s is None else s.numel()
    score = torch.empty(M, device=cproj.device)
    attn = torch.empty_like(s) if want_attn else None
    L.check(L.lib().mhimx_pseudo_score(_stream(), _p(s), _p(stats), _p(cproj), _p(bp), _p(score), _p(attn), M,
                                       cproj.shape[1]), "mhimx_pseudo_score")
    return (score, attn) if want_attn else score


def lse_merge(parts):
    """parts [W, 2+E] (max, L, z) of W shard-local pools -> (stats [2], z [E]) of the whole bag."""
    _chk(parts, name="parts")
    W, E = parts.shape[0], parts.shape[1] - 2
    stats, z = torch.empty(2, device=parts.device), torch.empty(E, device=parts.device)
    L.check(L.lib().mhimx_lse_merge(_stream(), _p(parts), W, E, _p(stats), _p(z)), "mhimx_lse_merge")
    return stats, z


# ------------------------------------------------------------------------------------------------ select
def select_mask(score, k, n_sel, largest=True, perm=None, other=None, want_topk=False):
    """Device-side select_mask_fn (2-D scores).  Returns (mask_ids int64 [N], len_keep_dev int64 [1], topk|None)."""
    _chk(score, name="score"); _chk(perm, torch.int64, "perm"); _chk(other, torch.int64, "other")
    N = score.numel()
    dev = score.device
    mask_ids = torch.empty(N, device=dev, dtype=torch.int64)
    len_keep = torch.empty(1, device=dev, dtype=torch.int64)
    topk = torch.empty(k, device=dev, dtype=torch.int64) if want_topk else None
    ws = torch.empty(L.lib().mhimx_select_ws_bytes(N), device=dev, dtype=torch.uint8)
    L.check(L.lib().mhimx_select_mask(_stream(), _p(score), N, int(k), int(n_sel), int(bool(largest)), _p(perm), _p(other),
                                      0 if other is None else other.numel(), _p(mask_ids), _p(len_keep), _p(topk), _p(ws),
                                      ws.numel()), "mhimx_select_mask")
    return mask_ids, len_keep, topk


def select_rows(score, k, n_sel, merge_R, rand_seed, tick=None, largest=True, want_mask_ids=False, merge_first=False, out=None):
    """Fused HAM mask + Merge split with device-side random subsets -> rows int64 [N - n_sel] (stay | merge), or
    (merge | stay) with merge_first."""
    _chk(score, name="score")
    N = score.numel()
    dev = score.device
    rows = out if out is not None else torch.empty(N - n_sel, device=dev, dtype=torch.int64)
    _chk(rows, torch.int64, "rows out")
    mask_ids = torch.empty(N, device=dev, dtype=torch.int64) if want_mask_ids else None
    ws = torch.empty(L.lib().mhimx_select_ws_bytes(N), device=dev, dtype=torch.uint8)
    L.check(L.lib().mhimx_select_rows(_stream(), _p(score), N, int(k), int(n_sel), int(bool(largest)),
                                      int(rand_seed) & 0xFFFFFFFFFFFFFFFF, _p(tick), int(merge_R), _p(rows), _p(mask_ids), _p(ws),
                                      ws.numel(), int(bool(merge_first))), "mhimx_select_rows")
    return (rows, mask_ids) if want_mask_ids else rows


def select_rows_img(score, k, n_sel, merge_R, rand_seed, img_merge_off, tick=None, largest=True, merge_first=True):
    """select_rows + the kept rows in the dPRE image's order (mhimx_select_rows_img): returns (rows [N - n_sel], rows_img
    [ceil((img_merge_off + merge_R) / 32) * 32] = [stay | 0.. | merge from img_merge_off | 0..])."""
    _chk(score, name="score")
    N = score.numel()
    dev = score.device
    rows = torch.empty(N - n_sel, device=dev, dtype=torch.int64)
    rows_img = torch.empty(-(-(int(img_merge_off) + int(merge_R)) // 32) * 32, device=dev, dtype=torch.int64)
    ws = torch.empty(L.lib().mhimx_select_ws_bytes(N), device=dev, dtype=torch.uint8)
    L.check(L.lib().mhimx_select_rows_img(_stream(), _p(score), N, int(k), int(n_sel), int(bool(largest)), int(rand_seed) & 0xFFFFFFFFFFFFFFFF,
                                          _p(tick), int(merge_R), _p(rows), _p(rows_img), int(img_merge_off), _p(ws), ws.numel(),
                                          int(bool(merge_first))), "mhimx_select_rows_img")
    return rows, rows_img


def vote_scores(attn, k, largest=True):
    _chk(attn, name="attn")
    H, N = attn.shape
    vote = torch.empty(N, device=attn.device)
    L.check(L.lib().mhimx_vote_scores(_stream(), _p(attn), H, N, int(k), int(bool(largest)), _p(vote), None, 0),
            "mhimx_vote_scores")
    return vote


def random_perm(n, seed, tick=None, src=None, out=None, device=None):
    """A keyed pseudo-random permutation of 0..n-1 (or src permuted by it) as ONE element-wise launch (mhimx_random_perm): what the
    reference draws with torch.randperm."""
    _chk(src, torch.int64, "src"); _chk(out, torch.int64, "out")
    dev = device if device is not None else (src.device if src is not None else (out.device if out is not None else torch.device("cuda")))
    if out is None:
        out = torch.empty(int(n), dtype=torch.int64, device=dev)
    L.check(L.lib().mhimx_random_perm(_stream(), int(n), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(tick), _p(src), _p(out)), "mhimx_random_perm")
    return out


def compose_ids(a, b):
    _chk(a, torch.int64, "a"); _chk(b, torch.int64, "b")
    out = torch.empty_like(b)
    L.check(L.lib().mhimx_compose_ids(_stream(), _p(a), _p(b), _p(out), b.numel()), "mhimx_compose_ids")
    return out


# ------------------------------------------------------------------------------------------------ merge
class MergeW:
    def __init__(self, q_param, ln_w, ln_b, wkv, wq, wo, bo, mm, heads=8, dim_head=64, drop_p=0.0, drop_seed=0,
                 prec="bf16x3", transposes=None, drop_tick=None, wkv_frag=None, x_rows=None, prepared=False, own=None, rep=1.0):
        """own = (lo, n): a shard of an instance-sharded bag - x_rows holds bag row ids, X / dX only the rows [lo, lo + n) (mhimx_merge.own_*);
        rep: weight of the gradient terms every shard computes alike (1 on one shard, 0 on the others)."""
        self.t = [q_param, ln_w, ln_b, wkv, wq, wo, bo, wkv_frag]
        _chk(x_rows, torch.int64, "x_rows")
        self.x_rows = x_rows
        for t in self.t:
            _chk(t, name="merge weight")
        self.k, self.E = q_param.shape[-2], q_param.shape[-1]
        self.heads, self.dim_head = heads, dim_head
        self.tr = transposes or (None, None, None)
        self.c = L.Merge(E=self.E, k=self.k, heads=heads, dim_head=dim_head, q_param=_p(q_param), ln_w=_p(ln_w),
                         ln_b=_p(ln_b), wkv=_p(wkv), wq=_p(wq), wo=_p(wo), bo=_p(bo), wkv_t=_p(self.tr[0]),
                         wq_t=_p(self.tr[1]), wo_t=_p(self.tr[2]), mm=float(mm), drop_p=float(drop_p),
                         drop_seed=int(drop_seed) & 0xFFFFFFFFFFFFFFFF, prec=prec_code(prec), drop_tick=_p(drop_tick),
                         wkv_frag=_p(wkv_frag), x_rows=_p(x_rows), prepared=int(bool(prepared)),
                         own_lo=0 if own is None else int(own[0]), own_n=0 if own is None else int(own[1]), rep=float(rep))
        self.own = own

    def ws_for(self, R, device):
        n = L.lib().mhimx_merge_ws_bytes(R, self.E, self.k, self.heads, self.dim_head)
        return torch.empty(n, device=device, dtype=torch.uint8)


def merge_fwd(mw: MergeW, X, z_out=None, update_q=True, ws=None, q_out=None, rows_done=False):
    """q_out: where the EMA-updated queries go (may be the query parameter itself: the update is element-wise and the
    forward has consumed LayerNorm(q) by then)."""
    _chk(X, name="X")
    R = X.shape[0] if mw.x_rows is None else mw.x_rows.shape[0]
    dev = X.device
    z = z_out if z_out is not None else torch.empty((mw.k, mw.E), device=dev)
    q_new = (q_out if q_out is not None else torch.empty((mw.k, mw.E), device=dev)) if update_q else None
    ws = ws if ws is not None else mw.ws_for(R, dev)
    mw.c.rows_done = int(bool(rows_done))            # (the rows pass rode in the caller's previous launch: abmil_pool_fwd_split)
    L.check(L.lib().mhimx_merge_fwd(_stream(), C.byref(mw.c), _p(X), R, _p(z), _p(q_new), int(bool(update_q)), _p(ws),
                                    ws.numel()), "mhimx_merge_fwd")
    mw.c.rows_done = 0
    return z, q_new, ws


def merge_fwd_part(mw: MergeW, X, ws=None, part=None):
    """One shard's half of the forward of an instance-sharded bag (mhimx_merge_fwd_part): -> (part [mhimx_merge_part_floats()], ws)."""
    _chk(X, name="X")
    R = mw.x_rows.shape[0]
    dev = X.device
    part = part if part is not None else torch.empty(L.lib().mhimx_merge_part_floats(), device=dev)
    ws = ws if ws is not None else mw.ws_for(R, dev)
    L.check(L.lib().mhimx_merge_fwd_part(_stream(), C.byref(mw.c), _p(X), R, _p(part), _p(ws), ws.numel()), "mhimx_merge_fwd_part")
    return part, ws


def merge_fwd_finish(mw: MergeW, parts, ws, z_out=None, update_q=True, q_out=None):
    """The other half (mhimx_merge_fwd_finish): parts [W, part floats] of all shards -> (tokens z, updated queries)."""
    _chk(parts, name="parts")
    R = mw.x_rows.shape[0]
    dev = parts.device
    z = z_out if z_out is not None else torch.empty((mw.k, mw.E), device=dev)
    q_new = (q_out if q_out is not None else torch.empty((mw.k, mw.E), device=dev)) if update_q else None
    L.check(L.lib().mhimx_merge_fwd_finish(_stream(), C.byref(mw.c), _p(parts), parts.shape[0], R, _p(z), _p(q_new), int(bool(update_q)),
                                           _p(ws), ws.numel()), "mhimx_merge_fwd_finish")
    return z, q_new


def merge_bwd_park(mw: MergeW, X, dz, ws, grads, accumulate=False, defer=None):
    """BEFORE the pool backward that produces ``dz``: park the Merge backward's first stage on the step's list so that it rides in the pool
    backward's rows launch (mhimx_merge_bwd_park; a no-op for shapes / callers it does not apply to).  Same ``mw`` / ``grads`` / ``ws`` as the
    merge_bwd call that follows."""
    if defer is None or mw.x_rows is None:
        return
    R = mw.x_rows.shape[0]
    g = L.MergeGrad(d_ln_w=_p(grads["d_ln_w"]), d_ln_b=_p(grads["d_ln_b"]), d_wkv=_p(grads["d_wkv"]), d_wq=_p(grads["d_wq"]),
                    d_wo=_p(grads["d_wo"]), d_bo=_p(grads["d_bo"]), accumulate=int(bool(accumulate)), splits=8, defer=_dp(defer))
    L.check(L.lib().mhimx_merge_bwd_park(C.byref(mw.c), _p(X), R, _p(dz), _p(grads["dX"]), C.byref(g), _p(ws), ws.numel()), "mhimx_merge_bwd_park")


def merge_bwd(mw: MergeW, X, dz, ws, splits=8, grads=None, accumulate=False, defer=None):
    dev = X.device
    R, E = X.shape
    if mw.x_rows is not None:
        R = mw.x_rows.shape[0]
        if not grads or "dX" not in grads:
            raise L.MhimxError("merge_bwd with gathered rows scatters into a caller-provided dX buffer")
    I = mw.heads * mw.dim_head
    out = grads or {}
    out.setdefault("dX", torch.empty((R, E), device=dev))
    out.setdefault("d_ln_w", torch.empty(E, device=dev))
    out.setdefault("d_ln_b", torch.empty(E, device=dev))
    out.setdefault("d_wkv", torch.empty((2 * I, E), device=dev))
    out.setdefault("d_wq", torch.empty((I, E), device=dev))
    out.setdefault("d_wo", torch.empty((E, I), device=dev))
    out.setdefault("d_bo", torch.empty(E, device=dev))
    g = L.MergeGrad(d_ln_w=_p(out["d_ln_w"]), d_ln_b=_p(out["d_ln_b"]), d_wkv=_p(out["d_wkv"]), d_wq=_p(out["d_wq"]),
                    d_wo=_p(out["d_wo"]), d_bo=_p(out["d_bo"]), accumulate=int(bool(accumulate)), splits=int(splits),
                    defer=_dp(defer))
    if defer is not None:
        defer.keep.append(ws)
    L.check(L.lib().mhimx_merge_bwd(_stream(), C.byref(mw.c), _p(X), R, _p(dz), _p(out["dX"]), C.byref(g), _p(ws),
                                    ws.numel()), "mhimx_merge_bwd")
    return out


# ------------------------------------------------------------------------------------------------ misc
def act_bwd(dH, H, pre, act, drop_p=0.0, drop_seed=0, drop_mask=None, rows=None, colsum_out=None, want_colsum=False,
            accumulate=False, drop_tick=None):
    """In-place backward through activation+dropout; optionally the column sums (bias gradient) in the same pass."""
    M, E = dH.shape
    if want_colsum and colsum_out is None:
        colsum_out = torch.empty(E, device=dH.device)
    ws = torch.empty(1024 * E, device=dH.device) if colsum_out is not None else None
    L.check(L.lib().mhimx_act_bwd(_stream(), _p(dH), _p(H), _p(pre), M, E, int(act), float(drop_p),
                                  int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _p(drop_mask), _p(rows), _p(colsum_out),
                                  int(bool(accumulate)), _p(ws), 0 if ws is None else ws.numel() * 4, _p(drop_tick)), "mhimx_act_bwd")
    return (dH, colsum_out) if colsum_out is not None else dH


def mul_colsum(dH, dact, colsum_out=None, want_colsum=True, accumulate=False, defer=None):
    """dH *= dact in place; column sums of the result (the feature-bias gradient) in the same pass."""
    _chk(dH, name="dH"); _chk(dact, name="dact")
    M, E = dH.shape
    if want_colsum and colsum_out is None:
        colsum_out = torch.empty(E, device=dH.device)
    ws = torch.empty(1024 * E, device=dH.device) if colsum_out is not None else None
    L.check(L.lib().mhimx_mul_colsum(_stream(), _p(dH), _p(dact), M, E, _p(colsum_out), int(bool(accumulate)), _p(ws),
                                     0 if ws is None else ws.numel() * 4, _dp(defer)), "mhimx_mul_colsum")
    if defer is not None:
        defer.keep.append(ws)
    return dH, colsum_out


def rows_dpre(dH, dact16, rows, n_rows, colsum_out=None, accumulate=False, defer=None):
    """dpre [L,E] = dH[rows] * dact16[rows] (compact) and colsum_out[e] (+)= sum_p dpre[p,e]  (mhimx_rows_dpre)."""
    _chk(dH, name="dH"); _chk(dact16, torch.float16, "dact16"); _chk(rows, torch.int64, "rows")
    E = dH.shape[1]
    dpre = torch.empty((n_rows, E), device=dH.device)
    if colsum_out is None:
        colsum_out = torch.empty(E, device=dH.device)
    ws = torch.empty(1024 * E, device=dH.device)
    L.check(L.lib().mhimx_rows_dpre(_stream(), _p(dH), _p(dact16), _p(rows), int(n_rows), E, _p(dpre), _p(colsum_out), int(bool(accumulate)),
                                      _p(ws), ws.numel() * 4, _dp(defer)), "mhimx_rows_dpre")
    if defer is not None:
        defer.keep.append(ws)
    return dpre, colsum_out


def bag_wgrad_ok(x, E, n_rows):
    """Shapes the dedicated weight-gradient pair (mhimx_rows_dpre_image + mhimx_bag_wgrad) takes."""
    return (n_rows >= 1 and E % 128 == 0 and x.shape[1] % 256 == 0 and x.stride(0) % 4 == 0
            and x.shape[0] * x.stride(0) * 4 < (1 << 32))


def bag_ximage_floats(x):
    """Floats of the bag's weight-gradient operand image (prep job PREP_XIMG of x [N, D])."""
    return -(-x.shape[0] // 32) * 32 * x.shape[1]


def bag_wgrad_ximg_ok(x, E):
    """Shapes the all-image weight-gradient product takes (bag_wgrad(ximg=)): contiguous rows, D % 256 == 0, E % 128 == 0."""
    return x.dim() == 2 and x.is_contiguous() and x.shape[1] % 256 == 0 and E % 128 == 0


def bag_wgrad(dH, dact16, x, rows, n_rows, out_w=None, out_b=None, accumulate=False, defer=None, rows_dh="same", want_bias=True,
              dh_compact=False, ride_tail=False, ximg=None, keep=None):
    """The projection's weight and bias gradient from the bag-ordered buffers:
    dPre[p] = dH[rows[p]] * dact16[rows[p]],  out_b (+)= sum_p dPre[p],  out_w [E,D] (+)= dPre^T x[rows]   (p < n_rows)
    — mhimx_rows_dpre_image (dPre as a bf16 hi/lo matrix-core image) + mhimx_bag_wgrad.
    dact16 None: dPre = dH (any Linear's weight gradient dy^T x); rows_dh: the row list of dH when it differs from x's (None: dH is
    compact); want_bias False: no column sums; dh_compact: dH is compact while dact16 is gathered by rows (mhimx_rows_dpre_image_c).
    ride_tail (with defer): the reductions already queued on the list and the last stage of a parked Merge tail run as trailing workgroups
    of the product launch (mhimx_bag_wgrad_args.ride_tail) - the list then holds the product's own slab sum only."""
    _chk(dH, name="dH"); _chk(dact16, torch.float16, "dact16"); _chk(rows, torch.int64, "rows")
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise L.MhimxError("x: expected a GPU fp32 matrix with contiguous rows")
    if ximg is not None:
        # both operands as images (mhimx_bag_wgrad_args.ximg): the k dimension runs over the bag's rows in memory order, `keep` (uint8 [N])
        # says which of them took part (rows / n_rows are not used)
        _chk(ximg, name="ximg"); _chk(keep, torch.uint8, "keep")
        N, E, D = x.shape[0], dH.shape[1], x.shape[1]
        if not bag_wgrad_ximg_ok(x, E) or ximg.numel() < bag_ximage_floats(x) or (keep is not None and keep.numel() < N):
            raise L.MhimxError("bag_wgrad(ximg=): contiguous bag, D % 256 == 0, E % 128 == 0, image / keep flags of the bag's size")
        dev, lib = dH.device, L.lib()
        if out_w is None:
            out_w = torch.empty((E, D), device=dev)
        if out_b is None and want_bias:
            out_b = torch.empty(E, device=dev)
        img = torch.empty(lib.mhimx_wgrad_image_bytes(N, E) // 4, device=dev)
        ws_b = torch.empty(-(-N // 32) * E, device=dev) if want_bias else None
        L.check(lib.mhimx_rows_dpre_image_k(_stream(), _p(dH), _p(dact16), _p(keep), N, E, _p(img), _p(out_b) if want_bias else None,
                                            int(bool(accumulate)), _p(ws_b), ws_b.numel() * 4 if want_bias else 0, _dp(defer)), "mhimx_rows_dpre_image_k")
        ws = torch.empty(lib.mhimx_wgrad_ws_floats(N, E, D), device=dev)
        g = L.BagWgrad(img=_p(img), X=_p(x), ldx=x.stride(0), n_bag_rows=N, rows=None, L=N, E=E, D=D, C=_p(out_w), ldc=out_w.stride(0),
                       accumulate=int(bool(accumulate)), ws=_p(ws), ws_floats=ws.numel(), defer=_dp(defer),
                       ride_tail=int(bool(ride_tail and defer is not None)), ximg=_p(ximg))
        L.check(lib.mhimx_bag_wgrad(_stream(), C.byref(g)), "mhimx_bag_wgrad")
        if defer is not None:
            defer.keep.extend((ws_b, ws, img))
        return out_w, out_b
    rows_h = rows if isinstance(rows_dh, str) else rows_dh
    _chk(rows_h, torch.int64, "rows_dh")
    E, D = dH.shape[1], x.shape[1]
    n_rows = int(n_rows)
    dev = dH.device
    lib = L.lib()
    if out_w is None:
        out_w = torch.empty((E, D), device=dev)
    if out_b is None and want_bias:
        out_b = torch.empty(E, device=dev)
    img = torch.empty(lib.mhimx_wgrad_image_bytes(n_rows, E) // 4, device=dev)
    ws_b = torch.empty(-(-n_rows // 32) * E, device=dev) if want_bias else None
    fn = lib.mhimx_rows_dpre_image_c if dh_compact else lib.mhimx_rows_dpre_image
    L.check(fn(_stream(), _p(dH), _p(dact16), _p(rows_h), n_rows, E, _p(img), _p(out_b) if want_bias else None,
                                      int(bool(accumulate)), _p(ws_b), ws_b.numel() * 4 if want_bias else 0, _dp(defer)), "mhimx_rows_dpre_image")
    ws = torch.empty(lib.mhimx_wgrad_ws_floats(n_rows, E, D), device=dev)
    g = L.BagWgrad(img=_p(img), X=_p(x), ldx=x.stride(0), n_bag_rows=x.shape[0], rows=_p(rows), L=n_rows, E=E, D=D, C=_p(out_w),
                   ldc=out_w.stride(0), accumulate=int(bool(accumulate)), ws=_p(ws), ws_floats=ws.numel(), defer=_dp(defer),
                   ride_tail=int(bool(ride_tail and defer is not None)))
    L.check(lib.mhimx_bag_wgrad(_stream(), C.byref(g)), "mhimx_bag_wgrad")
    if defer is not None:
        defer.keep.extend((ws_b, ws, img))
    return out_w, out_b


class WgradImage:
    """One bag's half of the projection's weight gradient, parked until the window's ONE product launch (bag_wgrad_multi): the dPre image
    (mhimx_rows_dpre_image: already made, bias gradient included), the bag, its row ids."""

    def __init__(self, img, x, rows, n_rows, E, keep):
        self.img, self.x, self.rows, self.n_rows, self.E, self.keep = img, x, rows, int(n_rows), int(E), keep


def bag_wgrad_image(dH, dact16, x, rows, n_rows, out_b=None, accumulate=False, defer=None):
    """The first half of bag_wgrad alone: dPre = dH[rows] * dact16[rows] as the matrix-core image + the bias gradient out_b (+)= sum dPre.
    Returns a WgradImage for bag_wgrad_multi."""
    _chk(dH, name="dH"); _chk(dact16, torch.float16, "dact16"); _chk(rows, torch.int64, "rows")
    E = dH.shape[1]
    n_rows = int(n_rows)
    dev = dH.device
    lib = L.lib()
    img = torch.empty(lib.mhimx_wgrad_image_bytes(n_rows, E) // 4, device=dev)
    ws_b = torch.empty(-(-n_rows // 32) * E, device=dev)
    if out_b is None:
        out_b = torch.empty(E, device=dev)
    L.check(lib.mhimx_rows_dpre_image(_stream(), _p(dH), _p(dact16), _p(rows), n_rows, E, _p(img), _p(out_b), int(bool(accumulate)), _p(ws_b),
                                      ws_b.numel() * 4, _dp(defer)), "mhimx_rows_dpre_image")
    if defer is not None:
        defer.keep.append(ws_b)
    return WgradImage(img, x, rows, n_rows, E, (ws_b, dH, dact16))


def bag_wgrad_multi(images, out_w, accumulate=False, defer=None):
    """out_w [E, D] (+)= sum over the window's bags of dPre_b^T x_b[rows_b]: ONE launch (mhimx_bag_wgrad_multi); images: WgradImage list."""
    _chk(out_w, name="out_w")
    a0 = images[0]
    D = a0.x.shape[1]
    lib = L.lib()
    nb = len(images)
    ws = torch.empty(lib.mhimx_wgrad_multi_ws_floats(a0.n_rows, a0.E, D, nb) if nb > 1 else lib.mhimx_wgrad_ws_floats(a0.n_rows, a0.E, D),
                     device=out_w.device)
    arr = (L.BagWgrad * nb)()
    for b, im in enumerate(images):
        arr[b] = L.BagWgrad(img=_p(im.img), X=_p(im.x), ldx=im.x.stride(0), n_bag_rows=im.x.shape[0], rows=_p(im.rows), L=im.n_rows, E=im.E, D=D,
                            C=_p(out_w), ldc=out_w.stride(0), accumulate=int(bool(accumulate)), ws=_p(ws), ws_floats=ws.numel(), defer=_dp(defer))
    L.check(lib.mhimx_bag_wgrad_multi(_stream(), arr, nb), "mhimx_bag_wgrad_multi")
    if defer is not None:
        defer.keep.extend([ws] + [im.img for im in images])
    return out_w


def shard_flags(rows_all, R, Lk, lo, n, k_tokens, tokens_live, out=None):
    """uint8 [n + k_tokens]: 0 for the shard's stay rows (rows_all[R:R+Lk] inside [lo, lo+n)) and, if tokens_live, the token rows; else 1."""
    _chk(rows_all, torch.int64, "rows_all")
    if out is None:
        out = torch.empty(n + k_tokens, dtype=torch.uint8, device=rows_all.device)
    L.check(L.lib().mhimx_shard_flags(_stream(), _p(rows_all), int(R), int(Lk), int(lo), int(n), int(k_tokens), int(bool(tokens_live)), _p(out)),
            "mhimx_shard_flags")
    return out


def shard_gather(H, rows, lo, n, out=None):
    """out[j] = H[rows[j] - lo] where the shard [lo, lo+n) owns rows[j], zero elsewhere."""
    _chk(H, name="H"); _chk(rows, torch.int64, "rows")
    R, E = rows.numel(), H.shape[1]
    if out is None:
        out = torch.empty((R, E), device=H.device)
    L.check(L.lib().mhimx_shard_gather(_stream(), _p(H), E, _p(rows), R, int(lo), int(n), _p(out)), "mhimx_shard_gather")
    return out


def shard_scatter(dX, rows, lo, n, dH):
    """dH[rows[j] - lo] = dX[j] for the rows the shard [lo, lo+n) owns."""
    _chk(dX, name="dX"); _chk(rows, torch.int64, "rows"); _chk(dH, name="dH")
    L.check(L.lib().mhimx_shard_scatter(_stream(), _p(dX), dX.shape[1], _p(rows), rows.numel(), int(lo), int(n), _p(dH)), "mhimx_shard_scatter")
    return dH


def colsum(X, out=None, accumulate=False):
    M, E = X.shape
    if out is None:
        out = torch.empty(E, device=X.device)
    ws = torch.empty(512 * E, device=X.device)                 # (>= 128 x E required; more lets the kernel use more, shorter row chunks)
    L.check(L.lib().mhimx_colsum(_stream(), _p(X), M, E, _p(out), int(bool(accumulate)), _p(ws), ws.numel() * 4), "mhimx_colsum")
    return out


def head_fwd_bwd(z, t, wp, bp, label, temp_t=1.0, main_alpha=1.0, aux_alpha=0.0, inv_accum=1.0, d_wp=None, d_bp=None,
                 accumulate=False, g_logits_in=None, g_cl_in=None, g_z=None):
    """logits, losses[3] = {main*ce + aux*cl, ce, cl}, g_z, d_wp, d_bp."""
    dev = z.device
    Cc, E = wp.shape
    logits = torch.empty(Cc, device=dev)
    losses = torch.empty(3, device=dev)
    g_z = g_z if g_z is not None else torch.empty(E, device=dev)
    d_wp = d_wp if d_wp is not None else torch.empty_like(wp)
    d_bp = d_bp if d_bp is not None else torch.empty(Cc, device=dev)
    L.check(L.lib().mhimx_head_fwd_bwd(_stream(), _p(z), _p(t), _p(wp), _p(bp), _p(label), E, Cc, float(temp_t),
                                       float(main_alpha), float(aux_alpha), float(inv_accum), _p(logits), _p(losses),
                                       _p(g_z), _p(d_wp), _p(d_bp), int(bool(accumulate)), _p(g_logits_in), _p(g_cl_in)),
            "mhimx_head_fwd_bwd")
    return logits, losses, g_z, d_wp, d_bp


def adam_ema(p, g, m, v, teacher, n_train, step, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-5,
             grad_scale=1.0, ema_mm=0.9997, zero_grad=True, step_dev=None, mm_table=None):
    n_all = p.numel()
    L.check(L.lib().mhimx_adam_ema(_stream(), _p(p), _p(g), _p(m), _p(v), _p(teacher), int(n_train), int(n_all), int(step),
                                   float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                   float(grad_scale), float(ema_mm), int(bool(zero_grad)), _p(step_dev), _p(mm_table),
                                   0 if mm_table is None else mm_table.numel()), "mhimx_adam_ema")


def optim_step(p, g, m, v, teacher, n_train, step, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-5, grad_scale=1.0,
               ema_mm=0.9997, zero_grad=True, step_dev=None, mm_table=None, lr_table=None, g_extra=None, clip_norm=None, ws=None, fold=None):
    """mhimx_optim_step: fused Adam + EMA with the optional device-side pieces - ``lr_table`` (per-update schedule), ``g_extra``
    ([S, pitch] gradient slabs added to g first), ``clip_norm`` (clip_grad_norm_; ``ws``: >= 1024 floats), ``fold`` (a ReduceList: its
    split-K slab sums into g are performed by the update kernel itself, the rest is flushed first)."""
    _chk(g_extra, name="g_extra"); _chk(lr_table, name="lr_table"); _chk(mm_table, name="mm_table")
    if clip_norm and ws is None:
        ws = torch.empty(1024, device=p.device)
    a = L.OptimArgs(p=_p(p), g=_p(g), m=_p(m), v=_p(v), teacher=_p(teacher), n_train=int(n_train), n_all=p.numel(), step=int(step),
                    step_dev=_p(step_dev), lr=float(lr), lr_table=_p(lr_table), lr_len=0 if lr_table is None else lr_table.numel(),
                    beta1=float(beta1), beta2=float(beta2), eps=float(eps), weight_decay=float(weight_decay), grad_scale=float(grad_scale),
                    ema_mm=float(ema_mm), mm_table=_p(mm_table), mm_len=0 if mm_table is None else mm_table.numel(),
                    zero_grad=int(bool(zero_grad)), g_extra=_p(g_extra), n_extra=0 if g_extra is None else g_extra.shape[0],
                    extra_pitch=0 if g_extra is None else g_extra.stride(0), clip_norm=float(clip_norm or 0.0), ws=_p(ws),
                    ws_floats=0 if ws is None else ws.numel(), fold=_dp(fold))
    L.check(L.lib().mhimx_optim_step(_stream(), C.byref(a)), "mhimx_optim_step")
    if fold is not None:
        fold.keep.clear()                      # (the slabs were read by the launch just enqueued; stream order protects them)
    return ws


def stream_copy(src, dst):
    """dst = src with the library's float4 stream-copy kernel (the HBM copy-rate microbenchmark of bench.py)."""
    _chk(src, name="src"); _chk(dst, name="dst")
    L.check(L.lib().mhimx_stream_copy(_stream(), _p(src), _p(dst), src.numel()), "mhimx_stream_copy")
    return dst


def tick(counter):
    """counter (uint64/int64 [1], device) += 1 on the current stream."""
    L.check(L.lib().mhimx_tick(_stream(), _p(counter)), "mhimx_tick")


# ------------------------------------------------------------------------------------------------ validation metrics
METRIC_KEYS = ("Acc", "AUC", "Precision", "Recall", "F1", "CK", "Acc_micro")


def cls_metrics(logits, labels, n_classes, bin_metric=False, sample_idx=None):
    """Device metrics of engines/metrics.py (mhimx_cls_metrics).  logits [n, C] fp32, labels [n] int64 -> out [B, 7] in the
    order METRIC_KEYS; ``sample_idx`` [B, n] int64 evaluates B bootstrap resamples in the same launches (B = 1 without)."""
    _chk(logits, name="logits"); _chk(labels, torch.int64, "labels"); _chk(sample_idx, torch.int64, "sample_idx")
    n, Cc = logits.shape
    if Cc != n_classes:
        raise L.MhimxError(f"cls_metrics: logits have {Cc} columns, n_classes = {n_classes}")
    B = 1 if sample_idx is None else sample_idx.shape[0]
    out = torch.empty((B, 7), device=logits.device)
    ws = torch.empty(L.lib().mhimx_cls_metrics_ws_bytes(n, Cc, B), device=logits.device, dtype=torch.uint8)
    L.check(L.lib().mhimx_cls_metrics(_stream(), _p(logits), logits.stride(0), _p(labels), n, Cc, int(bool(bin_metric)), _p(sample_idx), B,
                                      _p(out), _p(ws), ws.numel()), "mhimx_cls_metrics")
    return out


# ------------------------------------------------------------------------------------------- streamed Nystrom attention
class NysOperands:
    """mhimx_nys: the packed to_qkv output qkv [T, 1536] (q | k | v, heads = 64-column groups), the landmark means lm [256, 1024]
    (q~ | k~) and a workspace; keeps the tensors alive as long as the struct."""

    def __init__(self, qkv, lm, scale, ws=None):
        _chk(qkv, name="qkv"); _chk(lm, name="lm")
        T, ld = qkv.shape
        if ld != 1536 or tuple(lm.shape) != (256, 1024):
            raise L.MhimxError("nys: qkv must be [T, 1536] and lm [256, 1024] (8 heads x 64, 256 landmarks)")
        n = L.lib().mhimx_nys_ws_floats(T)
        self.ws = ws if ws is not None and ws.numel() >= n else torch.empty(n, device=qkv.device)
        self.qkv, self.lm, self.T = qkv, lm, T
        b = qkv.data_ptr()
        self.c = L.Nys(q=b, k=b + 512 * 4, v=b + 1024 * 4, ld=ld, T=T, ql=lm.data_ptr(), kl=lm.data_ptr() + 512 * 4, ldl=1024,
                       scale=float(scale), ws=self.ws.data_ptr(), ws_floats=self.ws.numel())

    def ref(self):
        return C.byref(self.c)


def nys_a3v_fwd(o: NysOperands):
    """a3v [8,256,64] = softmax_n(scale q~ k^T) v and the base-2 log-sum-exp of its rows [8,256]."""
    a3v = torch.empty((8, 256, 64), device=o.qkv.device)
    lse3 = torch.empty((8, 256), device=o.qkv.device)
    L.check(L.lib().mhimx_nys_a3v_fwd(_stream(), o.ref(), _p(a3v), _p(lse3)), "mhimx_nys_a3v_fwd")
    return a3v, lse3


def nys_out_fwd(o: NysOperands, w2, out=None, accumulate=False):
    """out [T, 512] (= or +=) softmax_m(scale q k~^T) w2 (heads side by side), lse1 [8, T]."""
    _chk(w2, name="w2")
    if accumulate and out is None:
        raise L.MhimxError("nys_out_fwd: accumulate needs the buffer to add to")
    out = torch.empty((o.T, 512), device=w2.device) if out is None else out
    lse1 = torch.empty((8, o.T), device=w2.device)
    L.check(L.lib().mhimx_nys_out_fwd(_stream(), o.ref(), _p(w2), _p(out), out.stride(0), _p(lse1), int(bool(accumulate))), "mhimx_nys_out_fwd")
    return out, lse1


def nys_out_bwd(o: NysOperands, w2, dout, lse1, dqkv, dlm):
    """Writes dq into dqkv[:, :512] and the S1 term of dk~ into dlm[:, 512:]; returns dw2 [8,256,64]."""
    _chk(dout, name="dout")
    dw2 = torch.empty_like(w2)
    delta = torch.empty_like(lse1)
    L.check(L.lib().mhimx_nys_out_bwd(_stream(), o.ref(), _p(w2), _p(dout), dout.stride(0), _p(lse1), _p(delta), _p(dqkv), dqkv.stride(0),
                                       C.c_void_p(dlm.data_ptr() + 512 * 4), dlm.stride(0), _p(dw2)), "mhimx_nys_out_bwd")
    return dw2


def nys_a3v_bwd(o: NysOperands, a3v, da3v, lse3, dqkv, dlm, accumulate_dv):
    """Writes dk into dqkv[:, 512:1024], dv (= or +=) into dqkv[:, 1024:], the S3 term of dq~ into dlm[:, :512]."""
    _chk(da3v, name="da3v")
    b = dqkv.data_ptr()
    L.check(L.lib().mhimx_nys_a3v_bwd(_stream(), o.ref(), _p(a3v), _p(da3v), _p(lse3), C.c_void_p(b + 512 * 4), C.c_void_p(b + 1024 * 4),
                                       dqkv.stride(0), int(bool(accumulate_dv)), _p(dlm), dlm.stride(0)), "mhimx_nys_a3v_bwd")


def nys_cls_attn(o: NysOperands, lse3, u):
    """r [8, T] = u attn3 (u [8, 256]): the cls token's attention row without attn3."""
    _chk(u, name="u")
    r = torch.empty((8, o.T), device=u.device)
    L.check(L.lib().mhimx_nys_cls_attn(_stream(), o.ref(), _p(lse3), _p(u), _p(r)), "mhimx_nys_cls_attn")
    return r
