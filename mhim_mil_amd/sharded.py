"""One giant bag, instance-sharded across GPUs (BASELINE config c5: N=200 000, D=1536 on 8 x MI355X).

Rank r holds a contiguous block of the bag's rows.  Rows are independent until a reduction over instances, so the
MHIM(ABMIL) train step needs exactly these exchanges (RCCL on GPUs; every one is small next to the N x D stream):

  teacher   all-gather of the shard-local softmax-pool partial (max, L, z[E])            (E+2 floats per rank)
            all-gather of the per-instance scores                                         (N floats in total)
  select    none — every rank runs the same top-k / random subsample on the full score vector with a shared-seed
            generator, so the index sets are identical on all ranks and identical to the single-GPU result
  student   all-gather of Merge's per-slot partials (max, sum, dropped sum, pooled row: 48 x 515 floats = 99 KB per rank; every rank
            runs Merge's rows pass over ITS rows of the merge list only - round 4; rounds 1-3 all-reduced the [R, E] block of rows to
            merge, 39.7 MB at c5, and ran Merge over all R rows on every rank)
            all-gather of the pool partial                                                (E+2 floats per rank)
  backward  all-reduce (= broadcast from rank 0) of d(merged tokens) [k, E]
            all-reduce(SUM) of the flat gradient buffer (replicated terms are kept on rank 0 only)

Merge's query side (O(k E^2)), the head and the optimiser run replicated on identical inputs: ``merge.global_q_mm`` and the Adam state
stay bit-identical on every rank with no extra traffic.  Merge's parameter gradients are partial sums over the ranks (the terms every rank
computes alike are weighted 1 on rank 0 and 0 elsewhere) and ride in the flat-gradient all-reduce.

Everything numeric is a kernel of libmhimx.so; torch.distributed only moves buffers.  With a ``gloo`` group (the CPU-side
test harness on a 1-GPU box) buffers are staged through host memory; with ``nccl`` (= RCCL) they stay in HBM.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops
from .engine import FlatState
from .mhim import MHIM, BagPlan, _FEATURE_ACTS

_RANK_SALT = 0x9E3779B97F4A7C15


class _Comm:
    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.stage = self.on and dist.get_backend(group) == "gloo"

    def all_reduce_sum(self, t):
        if self.world == 1:
            return t
        if self.stage:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def all_gather(self, t):
        """[W, *t.shape]; every rank contributes the same shape."""
        if self.world == 1:
            return t.unsqueeze(0)
        src = t.cpu() if self.stage else t.contiguous()
        out = torch.empty(self.world * src.numel(), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src.reshape(-1), group=self.group)
        return out.view((self.world,) + tuple(src.shape)).to(t.device)

    def all_gather_into(self, out, t):
        """out [W, *t.shape] (preallocated, static under graph replay) <- every rank's t."""
        if self.world == 1:
            out[0].copy_(t)
            return out
        if self.stage:
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h.view(-1), t.cpu().reshape(-1), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)
        return out

    def all_to_all_rows(self, send, send_counts, recv_counts):
        """send [sum(send_counts), C], grouped by destination rank -> [sum(recv_counts), C], grouped by source rank."""
        if self.world == 1:
            return send
        src = send.cpu() if self.stage else send.contiguous()
        out = torch.empty((int(sum(recv_counts)), send.shape[1]), dtype=send.dtype, device=src.device)
        dist.all_to_all_single(out, src, output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts],
                               group=self.group)
        return out.to(send.device)

    def all_gather_rows(self, t, counts):
        """Concatenate per-rank vectors of (known) different lengths."""
        if self.world == 1:
            return t
        m = max(counts)
        pad = torch.zeros(m, dtype=t.dtype, device=t.device)
        pad[:t.numel()] = t
        g = self.all_gather(pad)
        return torch.cat([g[r, :counts[r]] for r in range(self.world)])


def partition_rows(rows, Lk, lo, n):
    """Split the replicated row list ``rows`` = [rows that stay (Lk) | rows to merge] for the shard holding bag rows
    [lo, lo+n).  Returns (rows_local [n_loc] shard-local ids, stay rows first; n_stay; merge_pos [n_loc - n_stay] =
    positions of this shard's merge rows inside the merge list).  One host sync (the local counts are data dependent)."""
    inshard = (rows >= lo) & (rows < lo + n)
    pos = torch.nonzero(inshard).view(-1)
    n_stay = int((pos < Lk).sum())
    return (rows[pos] - lo).contiguous(), n_stay, (pos[n_stay:] - Lk).contiguous()


class ShardedBagTrainer:
    """Train step of MHIM(ABMIL) - or MHIM(TransMIL): sharded_transmil.py - on ONE bag whose rows are split across the ranks of ``group``."""

    def __init__(self, student: MHIM, teacher: MHIM, counts=None, group=None, seed=0, lr=2e-4, weight_decay=1e-5,
                 betas=(0.9, 0.999), eps=1e-8, mm=0.9997, main_alpha=1.0, aux_alpha=0.5):
        if student.baseline not in ("attn", "selfattn") or teacher.baseline != student.baseline:
            raise NotImplementedError("instance sharding is built for the ABMIL and the TransMIL (Nystrom) encoders, teacher and student alike "
                                      "(SURVEY.md §8(e)); DSMIL bags are replicas only")
        self.s, self.t = student, teacher
        # this trainer's kernels take their dropout stream position from the by-value seeds alone: a device tick left behind by
        # a FusedTrainer that held these models earlier would enter the forward's seed but not the backward's
        student._tick = None
        teacher._tick = None
        self.comm = _Comm(group)
        self.flat = FlatState(student, teacher)
        self.lr, self.wd, self.betas, self.eps, self.mm = lr, weight_decay, betas, eps, mm
        self.main_alpha, self.aux_alpha = main_alpha, aux_alpha
        dev = self.flat.student.device
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))              # SAME seed on every rank: identical randperm draws
        self.seed, self._n = int(seed), 0
        self.counts = counts                         # rows per rank (list); None = equal shards
        self.opt_step = torch.zeros(1, dtype=torch.int64, device=dev)
        self.tick = torch.zeros(1, dtype=torch.int64, device=dev)           # device step counter mixed into the fixed-shape step's dropout seeds
        self.step_count = 0
        self.shard_merge = True                      # Merge's rows sharded like the pool (world > 1; False: the replicated round-1..3 form)
        self.last = {}

    def _seeds(self):
        self._n += 1
        shared = (self.seed * 0xD1B54A32D192ED03 + self._n * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
        local = (shared + (self.comm.rank + 1) * _RANK_SALT) & 0xFFFFFFFFFFFFFFFF
        return shared, local

    def _pool_merge(self, st, E, dev):
        """all-gather the shard's (max, L, z) and merge -> the bag's (stats, z)."""
        part = torch.zeros(E + 2, device=dev)
        if st is not None:
            part[:2].copy_(st.stats)
            part[2:].copy_(st.z)
        else:
            part[0] = float("-inf")
        return ops.lse_merge(self.comm.all_gather(part).contiguous())

    # -------------------------------------------------------------------------------------------------
    def fixed_shape_ok(self, x):
        """The sync-free step needs the one-pass scorer (its ``excl`` flags) and the single-pass projection; train_step also asks for
        the v2 recipe (HAM mask only): a v1 ratio makes the number of masked rows data dependent (one host read-back in get_mask)."""
        s = self.s
        if s.baseline != "attn":
            return False
        att = s.online_encoder.attention
        return (not s.online_encoder.gated and s.prec != "f32" and s.mlp_dim == 512 and att.attention[0].weight.shape[0] == 128
                and x.shape[1] % 32 == 0 and s._feature_prec(1 << 20) == "bf16x3" and s.merge_enable)

    def _bag_rows(self, x_local):
        return sum(self.counts) if self.counts is not None else x_local.shape[-2] * self.comm.world

    def train_step(self, x_local, label, perm=None, ids_shuffle=None, i=None):
        """x_local [n_r, D]: this rank's rows (rank order = row order of the bag).  Returns (logits [C], losses [3])."""
        if self.s.baseline == "selfattn":              # TransMIL: re-balancing exchange + sequence-parallel encoder (sharded_transmil.py)
            from .sharded_transmil import transmil_step
            return transmil_step(self, x_local, label, perm, ids_shuffle, i)
        if self.fixed_shape_ok(x_local) and self.s.v2_counts(self._bag_rows(x_local), i) is not None:
            return self._step_fixed(x_local, label, perm, ids_shuffle, i)
        return self._step_generic(x_local, label, perm, ids_shuffle, i)

    def _step_fixed(self, x_local, label, perm, ids_shuffle, i):
        for exchange in self._fixed_gen(x_local, label, perm, ids_shuffle, i):
            exchange()
        return self.last["logits"], self.last["losses"]

    def _fixed_gen(self, x_local, label, perm, ids_shuffle, i):
        """The step with every launch shape fixed by (n, R, Lk, k): NO host read-back, no torch index op on the row lists.  A
        generator: it yields a thunk at every exchange (the caller runs it: eagerly, or between two captured graph segments).

        Each rank keeps its rows in bag order.  ONE projection launch makes the teacher's and the student's feature rows of all n local
        rows (mhimx_bag_project; the reference's student projects every row before it masks, mhim.py:335-336).  The student's pool runs
        over all n rows (+ the k merged tokens on rank 0) with a per-row exclusion flag (mhimx_pool_io.excl: masked rows, rows that were
        merged away, tokens on the other ranks get score -inf: weight 0, gradient 0), the rows to merge are collected into the replicated
        [R, E] block by ownership (mhimx_shard_gather, zero elsewhere, then all-reduce: exact), and the weight-gradient pair runs over
        all n rows (excluded rows carry a zero gradient).  Costs ~3 % more scorer rows than a compacted list, removes both host syncs."""
        s, t, fl, cm = self.s, self.t, self.flat, self.comm
        gv = fl.grad_views
        x = s._check_x(x_local)
        n, dev, E = x.shape[0], x.device, s.mlp_dim
        counts = self.counts if self.counts is not None else [n] * cm.world
        N, lo = sum(counts), sum(counts[:cm.rank])
        assert counts[cm.rank] == n, "counts[rank] must equal the local row count"
        shared_seed, local_seed = self._seeds()
        k = s.merge.k
        defer = ops.ReduceList()
        tick = self.tick                                           # device step counter: the dropout streams advance under graph replay
        s._tick = t._tick = tick
        try:
            # ---- parameter-only preparation of both models + the step counters: one launch
            jt, prep_t = t.prep_jobs(backward=False)
            js, prep_s = s.prep_jobs(backward=True, lean_merge=False)
            ops.prep_batch([(ops.PREP_TICK, None, tick), (ops.PREP_TICK, None, self.opt_step)] + jt + js)

            # ---- one projection launch: teacher rows, student rows (+ room for the k tokens), fp16 d out / d pre
            act = L.act_code(s.act, _FEATURE_ACTS)
            Hbuf = torch.empty((n + k, E), device=dev)
            p_t = t.dropout_p if t.training else 0.0
            heads = [ops.ProjHead(prep_t["w1p"], t.feature[0].bias.data, drop_p=p_t, drop_seed=local_seed ^ 0x5bd1e995),
                     ops.ProjHead(prep_s["w1p"], s.feature[0].bias.data, drop_p=s.dropout_p, drop_seed=local_seed, out=Hbuf, want_dact=True)]
            ops.bag_project(x, heads, act=act, drop_tick=tick)
            DACT = heads[1].dact

            # ---- teacher: partial pool -> the bag's (stats, z); scores need the bag's denominators
            wp = t.predictor.weight.data if t.attn2score else None
            st_t = ops.abmil_pool_fwd(t._scorer(prep_t.get("wa_frag")), heads[0].out, None, wp=wp, no_backward=True)
            part = torch.empty(E + 2, device=dev)
            parts = torch.empty((cm.world, E + 2), device=dev)
            part[:2].copy_(st_t.stats)
            part[2:].copy_(st_t.z)
            yield lambda: cm.all_gather_into(parts, part)
            gstats, t_feat = ops.lse_merge(parts)
            if t.attn2score:
                sc_loc = ops.pseudo_score(st_t.s, gstats, st_t.cproj, t.predictor.bias.data)
            else:
                sc_loc = ops.softmax_from_stats(st_t.s, gstats)
            if cm.world == 1:
                score = sc_loc
            else:                                                  # ragged shards: gather at the widest shard's length, then one fixed-shape copy per rank
                m = max(counts)
                pad = torch.zeros(m, device=dev)
                pad[:n].copy_(sc_loc)
                gathered = torch.empty((cm.world, m), device=dev)
                yield lambda: cm.all_gather_into(gathered, pad)
                score = torch.empty(N, device=dev)
                o = 0
                for r, c in enumerate(counts):
                    score[o:o + c].copy_(gathered[r, :c])
                    o += c

            # ---- select: replicated, identical on every rank -> [rows to merge (R) | rows that stay (Lk)]
            # (draws: injected, or made on the device from the SHARED seed and the device tick - inside the select kernel up to 16 384
            # instances, as keyed permutations (mhimx_random_perm) beyond: every rank computes the same lists, no generator, no sort)
            rows, len_keep, Lk, R = s.student_rows(N, i, score.view(1, -1), perm=perm, ids_shuffle=ids_shuffle, merge_first=True,
                                                   generator=None, seed=shared_seed)
            plan = BagPlan(rows=None, L=n, Lk=Lk, R=R, drop_seed=local_seed, mca_seed=shared_seed, training=True)
            excl = ops.shard_flags(rows, R, Lk, lo, n, k, cm.rank == 0)

            # ---- Merge.  Sharded like the pool (round 4): every rank runs the rows pass over ITS rows of the merge list only (the list holds
            # bag row ids; Hbuf the shard's rows), merges its tile partials and the ranks all-gather ONE 99 KB block each - per score slot
            # (max, sum, dropped sum, pooled row) - which every rank merges in rank order (mhimx_merge_fwd_part / _finish).  The replicated
            # form all-reduced the [R, E] block of rows to merge (39.7 MB at c5) and ran Merge over all R rows on every rank.
            shard_merge = (self.shard_merge and cm.world > 1 and s._op_prec != "f32" and E == 512 and s.merge.k * 8 <= 48 and R <= 32768)
            Hm = None
            if shard_merge:
                own, rep = (lo, n), (1.0 if cm.rank == 0 else 0.0)
                mw = s._merge_w(plan, x_rows=rows[:R], own=own, rep=rep)
                mpart, mws = ops.merge_fwd_part(mw, Hbuf)
                mparts = torch.empty((cm.world, mpart.numel()), device=dev)
                yield lambda: cm.all_gather_into(mparts, mpart)
                z_tok, q_new = ops.merge_fwd_finish(mw, mparts, mws, z_out=Hbuf[n:], update_q=True)
            else:                                                  # replicated: the [R, E] block (each row comes from exactly one rank)
                Hm = ops.shard_gather(Hbuf, rows[:R], lo, n)
                if cm.world > 1:
                    yield lambda: cm.all_reduce_sum(Hm)
                z_tok, q_new, mws = ops.merge_fwd(s._merge_w(plan, wkv_frag=prep_s.get("wkv_frag")), Hm, z_out=Hbuf[n:], update_q=True)
            q_old = prep_s["q_old"]
            s.merge.global_q_mm.data.copy_(q_new.view_as(s.merge.global_q_mm))

            # ---- student pool over all local rows + tokens, excluded rows flagged
            sc = s._scorer(prep_s.get("wa_frag"))
            st = ops.abmil_pool_fwd(sc, Hbuf, None, excl=excl)
            part2 = torch.empty(E + 2, device=dev)
            parts2 = torch.empty((cm.world, E + 2), device=dev)
            part2[:2].copy_(st.stats)
            part2[2:].copy_(st.z)
            yield lambda: cm.all_gather_into(parts2, part2)
            gstats, z = ops.lse_merge(parts2)

            # ---- head (replicated)
            t_in = t_feat if self.aux_alpha != 0. else None
            logits, losses, g_z, _, _ = ops.head_fwd_bwd(z, t_in, s.predictor.weight.data, s.predictor.bias.data, label,
                                                         temp_t=float(s.temp_t), main_alpha=self.main_alpha, aux_alpha=self.aux_alpha,
                                                         d_wp=gv["predictor.weight"], d_bp=gv["predictor.bias"])

            # ---- backward: the pool backward sees the BAG's softmax statistics and pooled feature
            st.stats.copy_(gstats)
            st.z.copy_(z)
            dHbuf = torch.empty_like(Hbuf)
            pre = "online_encoder.attention.attention."
            pool_g = {"dT1": dHbuf, "d_wa": gv[pre + "0.weight"], "d_wc": gv[pre + "2.weight"]}
            ops.abmil_pool_bwd(sc, st, g_z, prep_s["wa_t"], grads=pool_g, defer=defer, wa_t_frag=prep_s.get("wa_t_frag"))
            dT2 = dHbuf[n:]                                        # zero on the ranks whose tokens were excluded
            if cm.world > 1:
                yield lambda: cm.all_reduce_sum(dT2)               # = broadcast from rank 0
            mgr = {"d_ln_w": gv["merge.norm.weight"], "d_ln_b": gv["merge.norm.bias"], "d_wkv": gv["merge.attn.to_kv.weight"],
                   "d_wq": gv["merge.attn.to_q.weight"], "d_wo": gv["merge.attn.to_out.0.weight"], "d_bo": gv["merge.attn.to_out.0.bias"]}
            if shard_merge:
                # dX for the own rows lands in dHbuf directly (those rows were excluded from the pool: their slots hold zeros); the
                # parameter gradients are this rank's PARTIAL sums - the flat-gradient all-reduce below adds them (terms every rank
                # computes alike are weighted by rep: counted once)
                mgr["dX"] = dHbuf
                ops.merge_bwd(s._merge_w(plan, need_t=True, q=q_old, tr=prep_s.get("merge_t"), x_rows=rows[:R], own=own, rep=rep), Hbuf, dT2, mws,
                              grads=mgr, defer=defer)
            else:
                mg = ops.merge_bwd(s._merge_w(plan, need_t=True, q=q_old, tr=prep_s.get("merge_t")), Hm, dT2, mws, grads=mgr, defer=defer)
                ops.shard_scatter(mg["dX"], rows[:R], lo, n, dHbuf)    # (those rows were excluded from the pool: their slots hold zeros)
            if ops.bag_wgrad_ok(x, E, n):
                ops.bag_wgrad(dHbuf, DACT, x, None, n, out_w=gv["feature.0.weight"], out_b=gv["feature.0.bias"], defer=defer)
            else:
                dpre, _ = ops.rows_dpre(dHbuf, DACT, None, n, colsum_out=gv["feature.0.bias"], defer=defer)
                ops.gemm_tn(dpre, x, out=gv["feature.0.weight"], splits=8 if n >= 2048 else 1, prec="bf16x3", M=n, defer=defer)
            ops.reduce_flush(defer)
            if cm.world > 1:
                if cm.rank != 0:                                   # replicated terms: counted once in the SUM
                    for name in fl.train_names:
                        if name.startswith("predictor.") or (name.startswith("merge.") and not shard_merge):
                            gv[name].zero_()
                yield lambda: cm.all_reduce_sum(fl.grad[:fl.n_train])

            self.step_count += 1
            ops.adam_ema(fl.student, fl.grad, fl.m, fl.v, None if fl.same_teacher else fl.teacher, fl.n_train, self.step_count, lr=self.lr, beta1=self.betas[0],
                         beta2=self.betas[1], eps=self.eps, weight_decay=self.wd, grad_scale=1.0, ema_mm=self.mm, zero_grad=True,
                         step_dev=self.opt_step)
        finally:
            s._tick = t._tick = None
        self.last = {"logits": logits, "losses": losses, "patch_num": N, "keep_num": Lk + k, "rows": rows,
                     "len_keep": len_keep, "score": score, "teacher_feat": t_feat}

    # -------------------------------------------------------------------------------------------------
    def capture(self, x_local, label, warmup=2, i=None):
        """Capture the fixed-shape step as hipGraph segments with the exchanges between them (graph | collective | graph ...: a
        collective inside a captured graph depends on the RCCL build, and ranks must not mix replayed and eager collectives).  Returns
        a callable; every call replays one step on the SAME (x_local, label) buffers (copy the next shard into them)."""
        if self.s.baseline != "attn":
            raise L.MhimxError("capture(): the sharded TransMIL step (sharded_transmil.py) plans its exchanges on the host every step: run it eagerly")
        if not (self.fixed_shape_ok(x_local) and self.s.v2_counts(self._bag_rows(x_local), i) is not None):
            raise L.MhimxError("capture(): this model takes the generic sharded step (one host read-back per step): run it eagerly")
        if self.s.mrh_sche is not None:
            raise L.MhimxError("capture(): the HAM-ratio schedule changes the launch shapes per iteration")
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            for _ in range(warmup):
                self.train_step(x_local, label, i=i)
        torch.cuda.current_stream().wait_stream(cs)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        plan = []

        def new_graph():
            g = torch.cuda.CUDAGraph()
            g.register_generator_state(self.gen)
            g.capture_begin(pool=pool)
            return g

        with torch.cuda.stream(cs):
            gen = self._fixed_gen(x_local, label, None, None, i)
            g = new_graph()
            while True:
                try:
                    exchange = next(gen)
                except StopIteration:
                    g.capture_end()
                    plan.append(g.replay)
                    break
                g.capture_end()
                plan.append(g.replay)
                exchange()                                         # (runs now too: the capture pass is a real step)
                plan.append(exchange)
                g = new_graph()
        torch.cuda.current_stream().wait_stream(cs)
        self._captured = plan                                      # keeps graphs and thunks (and, through them, the static buffers) alive

        def replay():
            for f in plan:
                f()
            return self.last["logits"], self.last["losses"]

        return replay

    def _step_generic(self, x_local, label, perm=None, ids_shuffle=None, i=None):
        """Any scorer / feature shape: compacted local row lists (ONE host read-back of the data-dependent local counts per step)."""
        s, t, fl, cm = self.s, self.t, self.flat, self.comm
        gv = fl.grad_views
        x = s._check_x(x_local)
        n, dev, E = x.shape[0], x.device, s.mlp_dim
        counts = self.counts if self.counts is not None else [n] * cm.world
        N, lo = sum(counts), sum(counts[:cm.rank])
        assert counts[cm.rank] == n, "counts[rank] must equal the local row count"
        shared_seed, local_seed = self._seeds()

        # ---- teacher: local rows -> partial pool -> global (stats, z); scores need the global denominators
        with torch.no_grad():
            p = t.dropout_p if t.training else 0.0
            Ht = t._feature(x, None, p, local_seed ^ 0x5bd1e995)
            wp = t.predictor.weight.data if t.attn2score else None
            st_t = ops.abmil_pool_fwd(t._scorer(), Ht, None, wp=wp, no_backward=True)
            gstats, t_feat = self._pool_merge(st_t, E, dev)
            if t.attn2score:
                sc_loc = ops.pseudo_score(st_t.s, gstats, st_t.cproj, t.predictor.bias.data)
            else:
                sc_loc = ops.softmax_from_stats(st_t.s, gstats)
            score = cm.all_gather_rows(sc_loc, counts)
            del Ht, st_t

        # ---- select: replicated, identical on every rank (shared generator / injected draws)
        rows, len_keep, Lk, R = s.student_rows(N, i, score.view(1, -1), perm=perm, ids_shuffle=ids_shuffle, generator=None,
                                               seed=shared_seed)
        rows_local, n_stay, merge_pos = partition_rows(rows, Lk, lo, n)
        n_loc = rows_local.numel()
        n_merge = n_loc - n_stay
        if n_stay == 0:
            raise L.MhimxError("a shard without kept rows is not supported (bag too small for this many ranks)")
        plan = BagPlan(rows=rows_local, L=n_loc, Lk=n_stay, R=R, drop_seed=local_seed, mca_seed=shared_seed, training=True)

        # ---- student forward
        need_pre = L.act_code(s.act, _FEATURE_ACTS) == L.ACT["gelu"]
        H = torch.empty((n_loc, E), device=dev)
        PRE = torch.empty_like(H) if need_pre else None
        s._feature(x, rows_local, s.dropout_p, local_seed, None, out=H, pre_out=PRE, M=n_loc)
        Hm = torch.zeros((R, E), device=dev)
        if n_merge:
            Hm.index_copy_(0, merge_pos, H[n_stay:])
        cm.all_reduce_sum(Hm)                                      # every row has exactly one non-zero contributor: exact
        z_tok, q_new, mws = ops.merge_fwd(s._merge_w(plan), Hm, update_q=True)
        q_old = s.merge.global_q_mm.data.clone()
        s.merge.global_q_mm.data.copy_(q_new.view_as(s.merge.global_q_mm))
        sc = s._scorer()
        st = ops.abmil_pool_fwd(sc, H[:n_stay], z_tok if cm.rank == 0 else None)    # merged tokens counted once
        gstats, z = self._pool_merge(st, E, dev)

        # ---- head (replicated): predictor + CE + distillation and their gradients
        t_in = t_feat if self.aux_alpha != 0. else None
        logits, losses, g_z, _, _ = ops.head_fwd_bwd(z, t_in, s.predictor.weight.data, s.predictor.bias.data, label,
                                                     temp_t=float(s.temp_t), main_alpha=self.main_alpha, aux_alpha=self.aux_alpha,
                                                     d_wp=gv["predictor.weight"], d_bp=gv["predictor.bias"])

        # ---- backward: the pool backward sees the BAG's softmax statistics and pooled feature
        st.stats.copy_(gstats)
        st.z.copy_(z)
        dH = torch.empty_like(H)
        att = s.online_encoder.attention
        pre = "online_encoder.attention."
        pool_g = {"dT1": dH[:n_stay]}
        if s.online_encoder.gated:
            pool_g.update(d_wa=gv[pre + "attention_a.0.weight"], d_wb=gv[pre + "attention_b.0.weight"],
                          d_wc=gv[pre + "attention_c.weight"])
            g = ops.abmil_pool_bwd(sc, st, g_z, ops.transpose(att.attention_a[0].weight.data),
                                   ops.transpose(att.attention_b[0].weight.data), grads=pool_g)
        else:
            pool_g.update(d_wa=gv[pre + "attention.0.weight"], d_wc=gv[pre + "attention.2.weight"])
            g = ops.abmil_pool_bwd(sc, st, g_z, ops.transpose(att.attention[0].weight.data), grads=pool_g)
        dT2 = g["dT2"] if cm.rank == 0 else torch.zeros_like(z_tok)
        cm.all_reduce_sum(dT2)                                     # = broadcast from rank 0
        mgr = {"d_ln_w": gv["merge.norm.weight"], "d_ln_b": gv["merge.norm.bias"], "d_wkv": gv["merge.attn.to_kv.weight"],
               "d_wq": gv["merge.attn.to_q.weight"], "d_wo": gv["merge.attn.to_out.0.weight"], "d_bo": gv["merge.attn.to_out.0.bias"]}
        mg = ops.merge_bwd(s._merge_w(plan, need_t=True, q=q_old), Hm, dT2, mws, grads=mgr)
        if n_merge:
            dH[n_stay:] = mg["dX"].index_select(0, merge_pos)
        ops.act_bwd(dH, H, PRE, L.act_code(s.act, _FEATURE_ACTS), s.dropout_p, local_seed, None, rows_local,
                    colsum_out=gv["feature.0.bias"], want_colsum=True)
        ops.gemm_tn(dH, x, out=gv["feature.0.weight"], rows=rows_local, splits=8 if n_loc >= 2048 else 1,
                    prec="f32" if s.prec == "f32" else "bf16x3", M=n_loc)
        if cm.rank != 0:                                           # replicated terms: counted once in the SUM
            for name in fl.train_names:
                if name.startswith("predictor.") or name.startswith("merge."):
                    gv[name].zero_()
        cm.all_reduce_sum(fl.grad[:fl.n_train])

        # ---- optimiser + EMA teacher (replicated, identical inputs on every rank)
        self.step_count += 1
        ops.tick(self.opt_step)
        ops.adam_ema(fl.student, fl.grad, fl.m, fl.v, None if fl.same_teacher else fl.teacher, fl.n_train, self.step_count, lr=self.lr, beta1=self.betas[0],
                     beta2=self.betas[1], eps=self.eps, weight_decay=self.wd, grad_scale=1.0, ema_mm=self.mm, zero_grad=True,
                     step_dev=self.opt_step)
        self.last = {"logits": logits, "losses": losses, "patch_num": N, "keep_num": Lk + s.merge.k, "rows": rows,
                     "len_keep": len_keep, "score": score, "teacher_feat": t_feat}
        return logits, losses
