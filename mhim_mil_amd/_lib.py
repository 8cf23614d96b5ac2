"""ctypes binding of libmhimx.so (include/mhimx.h).  No fallback: if the HIP library is missing or a
symbol is absent this raises — the product never routes around the native path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, os.environ.get("MHIMX_LIB_NAME", "libmhimx.so"))

c_f32p = C.c_void_p
c_i64p = C.c_void_p
c_u8p = C.c_void_p

ABI_VERSION = 610          # == MHIMX_VERSION of the include/mhimx.h this binding was written against

ACT = {None: 0, "none": 0, "identity": 0, "relu": 1, "gelu": 2, "tanh": 3}
PREC = {"f32": 0, "f16s": 1, "bf16x3": 2}


def act_code(name, allowed):
    """Reference semantics: an activation name outside ``allowed`` silently means "no activation"
    (mhim.py:71-74 for the feature; baseline.py:17-22 for the scorer)."""
    n = name.lower() if isinstance(name, str) else None
    return ACT[n] if n in allowed else 0


class GemmNT(C.Structure):
    _fields_ = [("A", c_f32p), ("lda", C.c_int64), ("rows", c_i64p),
                ("B", c_f32p), ("ldb", C.c_int64),
                ("C", c_f32p), ("ldc", C.c_int64),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("bias", c_f32p), ("rowv", c_f32p), ("colv", c_f32p),
                ("pre", c_f32p), ("ldpre", C.c_int64),
                ("act", C.c_int32),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_mask", c_u8p),
                ("accumulate", C.c_int32), ("prec", C.c_int32), ("drop_tick", C.c_void_p),
                ("paired", C.c_int32), ("ws", C.c_void_p), ("ws_floats", C.c_int64),
                ("dact", C.c_void_p), ("lddact", C.c_int64)]


class ProjHead(C.Structure):
    _fields_ = [("wp", c_f32p), ("bias", c_f32p), ("H", c_f32p), ("ldh", C.c_int64), ("dact", C.c_void_p),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_mask", c_u8p), ("resid", c_f32p), ("ldr", C.c_int64)]


class ProjScore(C.Structure):
    """mhimx_proj_score (include/mhimx.h)."""
    _fields_ = [("wa16", c_f32p), ("wc", c_f32p), ("act", C.c_int32), ("C", C.c_int32), ("s", c_f32p), ("cproj", c_f32p),
                ("pm", c_f32p), ("pl", c_f32p), ("pz", c_f32p), ("xch", c_f32p), ("gate", C.c_void_p)]


class BagProject(C.Structure):
    _fields_ = [("X", c_f32p), ("ldx", C.c_int64), ("N", C.c_int64), ("D", C.c_int64), ("E", C.c_int64),
                ("act", C.c_int32), ("n_heads", C.c_int32), ("head", ProjHead * 2), ("drop_tick", C.c_void_p), ("score0", C.c_void_p)]


class BagWgrad(C.Structure):
    _fields_ = [("img", C.c_void_p), ("X", c_f32p), ("ldx", C.c_int64), ("n_bag_rows", C.c_int64), ("rows", C.c_void_p),
                ("L", C.c_int64), ("E", C.c_int64), ("D", C.c_int64), ("C", c_f32p), ("ldc", C.c_int64), ("accumulate", C.c_int32),
                ("ws", c_f32p), ("ws_floats", C.c_int64), ("defer", C.c_void_p), ("ride_tail", C.c_int32), ("ximg", C.c_void_p)]


class PpegBand(C.Structure):
    _fields_ = [("H", C.c_int64), ("cell0", C.c_int64), ("ncell", C.c_int64), ("out0", C.c_int64), ("out1", C.c_int64)]


class PrepJob(C.Structure):
    _fields_ = [("kind", C.c_int32), ("inp", C.c_void_p), ("out", C.c_void_p), ("R", C.c_int64), ("C", C.c_int64)]


class ReduceJob(C.Structure):
    _fields_ = [("kind", C.c_int32), ("accumulate", C.c_int32), ("parts", c_f32p), ("out", c_f32p),
                ("G", C.c_int64), ("W", C.c_int64), ("ld", C.c_int64), ("K1", C.c_int64), ("K2", C.c_int64), ("ldo", C.c_int64)]


REDUCE_MAX = 16


SIDE_BYTES = 512


class SideWork(C.Structure):
    _fields_ = [("pending", C.c_int32), ("reserved", C.c_int32), ("blob", C.c_ubyte * SIDE_BYTES)]


class ParkedGemm(C.Structure):
    _fields_ = [("pending", C.c_int32), ("reserved", C.c_int32), ("blob", C.c_ubyte * 128)]


class ReduceListC(C.Structure):
    _fields_ = [("j", ReduceJob * REDUCE_MAX), ("n", C.c_int32), ("side", SideWork), ("parked", ParkedGemm), ("pre", SideWork)]


class GemmTN(C.Structure):
    _fields_ = [("A", c_f32p), ("lda", C.c_int64),
                ("B", c_f32p), ("ldb", C.c_int64), ("rows", c_i64p),
                ("C", c_f32p), ("ldc", C.c_int64),
                ("M", C.c_int64), ("K1", C.c_int64), ("K2", C.c_int64),
                ("splits", C.c_int32), ("ws", c_f32p),
                ("accumulate", C.c_int32), ("prec", C.c_int32), ("ws_floats", C.c_int64),
                ("defer", C.c_void_p)]


class Scorer(C.Structure):
    _fields_ = [("E", C.c_int64), ("A", C.c_int64), ("act", C.c_int32), ("gated", C.c_int32), ("prec", C.c_int32),
                ("wa", c_f32p), ("ba", c_f32p), ("wb", c_f32p), ("bb", c_f32p), ("wc", c_f32p), ("bc", c_f32p),
                ("wa_frag", c_f32p), ("gate_drop_p", C.c_float), ("gate_drop_seed", C.c_uint64)]


class PoolIO(C.Structure):
    _fields_ = [("T1", c_f32p), ("M1", C.c_int64), ("T2", c_f32p), ("M2", C.c_int64),
                ("s", c_f32p), ("stats", c_f32p), ("z", c_f32p), ("u_pre", c_f32p),
                ("wp", c_f32p), ("C", C.c_int64), ("cproj", c_f32p),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("bp", c_f32p), ("pscore", c_f32p), ("rows1", c_i64p),
                ("excl", C.c_void_p), ("ride_jobs", C.c_void_p), ("n_ride_jobs", C.c_int32),
                ("phase", C.c_int32), ("tail_tokens", C.c_int32), ("ride_merge", C.c_void_p), ("ride_X", c_f32p), ("ride_R", C.c_int64),
                ("ride_ws", C.c_void_p), ("ride_ws_bytes", C.c_int64), ("rode_merge", C.c_int32), ("tail_wa_t", c_f32p), ("tail_row0", C.c_int64),
                ("no_backward", C.c_int32)]


class PoolGrad(C.Structure):
    _fields_ = [("g_z", c_f32p), ("dT1", c_f32p), ("dT2", c_f32p),
                ("d_wa", c_f32p), ("d_ba", c_f32p), ("d_wb", c_f32p), ("d_bb", c_f32p), ("d_wc", c_f32p),
                ("d_bc", c_f32p), ("wa_t", c_f32p), ("wb_t", c_f32p),
                ("accumulate", C.c_int32), ("splits", C.c_int32), ("defer", C.c_void_p), ("wa_t_frag", c_f32p),
                ("img", C.c_void_p), ("img_dact", C.c_void_p), ("img_part", c_f32p), ("img_rows", C.c_int64)]


class Merge(C.Structure):
    _fields_ = [("E", C.c_int64), ("k", C.c_int64), ("heads", C.c_int64), ("dim_head", C.c_int64),
                ("q_param", c_f32p), ("ln_w", c_f32p), ("ln_b", c_f32p),
                ("wkv", c_f32p), ("wq", c_f32p), ("wo", c_f32p), ("bo", c_f32p),
                ("wkv_t", c_f32p), ("wq_t", c_f32p), ("wo_t", c_f32p),
                ("mm", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("prec", C.c_int32), ("drop_tick", C.c_void_p),
                ("wkv_frag", c_f32p), ("x_rows", c_i64p), ("prepared", C.c_int32), ("own_lo", C.c_int64), ("own_n", C.c_int64),
                ("rep", C.c_float), ("rows_done", C.c_int32)]


class MergeGrad(C.Structure):
    _fields_ = [("d_ln_w", c_f32p), ("d_ln_b", c_f32p), ("d_wkv", c_f32p), ("d_wq", c_f32p), ("d_wo", c_f32p),
                ("d_bo", c_f32p), ("accumulate", C.c_int32), ("splits", C.c_int32), ("defer", C.c_void_p)]


# symbol -> (restype, argtypes); every symbol include/mhimx.h declares must be listed here
_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float
_U64 = C.c_uint64


class Nys(C.Structure):
    """mhimx_nys (include/mhimx.h): the streamed Nystrom attention's operands."""
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("ld", C.c_int64), ("T", C.c_int64),
                ("ql", C.c_void_p), ("kl", C.c_void_p), ("ldl", C.c_int64), ("scale", C.c_float),
                ("ws", C.c_void_p), ("ws_floats", C.c_int64)]


class BmmStep(C.Structure):
    """mhimx_bmm_step (include/mhimx.h)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("PN", C.c_void_p), ("PT", C.c_void_p),
                ("PN2", C.c_void_p), ("PT2", C.c_void_p), ("D", C.c_void_p), ("D2", C.c_void_p), ("alpha", C.c_float), ("ident", C.c_float),
                ("alpha2", C.c_float), ("ident2", C.c_float), ("dscale", C.c_float), ("d2scale", C.c_float), ("kind", C.c_int32)]


class OptimArgs(C.Structure):
    """mhimx_optim_args (include/mhimx.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("teacher", C.c_void_p),
                ("n_train", C.c_int64), ("n_all", C.c_int64), ("step", C.c_int64), ("step_dev", C.c_void_p),
                ("lr", C.c_float), ("lr_table", C.c_void_p), ("lr_len", C.c_int64),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("grad_scale", C.c_float),
                ("ema_mm", C.c_float), ("mm_table", C.c_void_p), ("mm_len", C.c_int64), ("zero_grad", C.c_int32),
                ("g_extra", C.c_void_p), ("n_extra", C.c_int64), ("extra_pitch", C.c_int64), ("clip_norm", C.c_float),
                ("ws", C.c_void_p), ("ws_floats", C.c_int64), ("fold", C.c_void_p), ("extra_lo", C.c_int64), ("extra_only", C.c_int32)]


class StepParams(C.Structure):
    """mhimx_step_params (include/mhimx.h): one model's parameters as device pointers."""
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "wa", "wc", "wp", "bp", "q", "ln_w", "ln_b", "wkv", "wq", "wo", "bo")]


class StepGrads(C.Structure):
    """mhimx_step_grads."""
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "wa", "wc", "wp", "bp", "ln_w", "ln_b", "wkv", "wq", "wo", "bo")]


class StepCfg(C.Structure):
    """mhimx_step_cfg."""
    _fields_ = [("D", C.c_int64), ("E", C.c_int64), ("A", C.c_int64), ("C", C.c_int64), ("k", C.c_int64),
                ("act", C.c_int32), ("da_act", C.c_int32), ("attn2score", C.c_int32),
                ("drop_p_teacher", C.c_float), ("drop_p_student", C.c_float), ("merge_drop_p", C.c_float), ("merge_mm", C.c_float),
                ("temp_t", C.c_float), ("main_alpha", C.c_float), ("aux_alpha", C.c_float),
                ("student", StepParams), ("teacher", StepParams), ("grad", StepGrads),
                ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("p_teacher", C.c_void_p),
                ("n_train", C.c_int64), ("n_all", C.c_int64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("ema_mm", C.c_float),
                ("mm_table", C.c_void_p), ("mm_len", C.c_int64), ("lr_table", C.c_void_p), ("lr_len", C.c_int64),
                ("tick", C.c_void_p), ("opt_step", C.c_void_p), ("q_out", C.c_void_p), ("time_project", C.c_int32), ("side_stream", C.c_void_p)]


class StepCounts(C.Structure):
    _fields_ = [("k_top", C.c_int64), ("n_sel", C.c_int64), ("len_keep", C.c_int64), ("Lk", C.c_int64), ("R", C.c_int64)]


class StepSeeds(C.Structure):
    _fields_ = [("drop_teacher", C.c_uint64), ("drop_student", C.c_uint64), ("select", C.c_uint64), ("mca", C.c_uint64)]


class StepLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("total", "logits", "losses", "score", "rows_all", "H_teacher", "H_student", "dact", "z_teacher",
                                        "z_student", "g_z", "dH")]


class WindowLayout(C.Structure):
    """mhimx_window_layout: bag b's copy of a per-bag buffer lies at bag.<field> + b * bag_stride."""
    _fields_ = [("total", C.c_int64), ("bag0", C.c_int64), ("bag_stride", C.c_int64), ("grad_slab", C.c_int64), ("bag", StepLayout)]


WINDOW_MAX = 8               # MHIMX_WINDOW_MAX

SYMBOLS = {
    "mhimx_last_error": (C.c_char_p, []),
    "mhimx_version": (C.c_int, []),
    "mhimx_gemm_nt": (C.c_int, [_P, C.POINTER(GemmNT)]),
    "mhimx_pool_finalize": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mhimx_proj_score_parts": (_I64, [_I64]),
    "mhimx_proj_score_xch_floats": (_I64, [_I64]),
    "mhimx_bag_project": (C.c_int, [_P, C.POINTER(BagProject)]),
    "mhimx_bag_project_multi": (C.c_int, [_P, _P, _I32]),
    "mhimx_gemm_nn": (C.c_int, [_P, C.POINTER(GemmNT), _F, _I32, _P]),
    "mhimx_prep_batch": (C.c_int, [_P, C.POINTER(PrepJob), _I32]),
    "mhimx_pair_planes": (C.c_int, [_P, _P, _I64, _I64, _I64, _P]),
    "mhimx_lse_merge": (C.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "mhimx_gemm_batched": (C.c_int, [_P, _I32, C.POINTER(GemmNT), _I32, _I64, _I64, _I64, _F, _I32, _P]),
    "mhimx_gemm_tn": (C.c_int, [_P, C.POINTER(GemmTN)]),
    "mhimx_transpose": (C.c_int, [_P, _P, _P, _I64, _I64]),
    "mhimx_abmil_pool_ws_bytes": (_I64, [_I64, _I64, _I64, _I32]),
    "mhimx_abmil_pool_fwd": (C.c_int, [_P, C.POINTER(Scorer), C.POINTER(PoolIO)]),
    "mhimx_abmil_pool_bwd": (C.c_int, [_P, C.POINTER(Scorer), C.POINTER(PoolIO), C.POINTER(PoolGrad)]),
    "mhimx_softmax_from_stats": (C.c_int, [_P, _P, _P, _P, _I64]),
    "mhimx_pseudo_score": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64]),
    "mhimx_select_ws_bytes": (_I64, [_I64]),
    "mhimx_select_mask": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _P, _P, _I64, _P, _P, _P, _P, _I64]),
    "mhimx_select_rows": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _U64, _P, _I64, _P, _P, _P, _I64, _I32]),
    "mhimx_select_rows_img": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _U64, _P, _I64, _P, _P, _I64, _P, _I64, _I32]),
    "mhimx_vote_scores": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _P, _P, _I64]),
    "mhimx_compose_ids": (C.c_int, [_P, _P, _P, _P, _I64]),
    "mhimx_random_perm": (C.c_int, [_P, _I64, _U64, _P, _P, _P]),
    "mhimx_merge_ws_bytes": (_I64, [_I64, _I64, _I64, _I64, _I64]),
    "mhimx_merge_fwd": (C.c_int, [_P, C.POINTER(Merge), _P, _I64, _P, _P, _I32, _P, _I64]),
    "mhimx_merge_part_floats": (_I64, []),
    "mhimx_merge_fwd_part": (C.c_int, [_P, C.POINTER(Merge), _P, _I64, _P, _P, _I64]),
    "mhimx_merge_fwd_finish": (C.c_int, [_P, C.POINTER(Merge), _P, _I32, _I64, _P, _P, _I32, _P, _I64]),
    "mhimx_merge_bwd": (C.c_int, [_P, C.POINTER(Merge), _P, _I64, _P, _P, C.POINTER(MergeGrad), _P, _I64]),
    "mhimx_merge_bwd_park": (C.c_int, [C.POINTER(Merge), _P, _I64, _P, _P, C.POINTER(MergeGrad), _P, _I64]),
    "mhimx_act_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I32, _F, _U64, _P, _P, _P, _I32, _P, _I64, _P]),
    "mhimx_softmax_cols": (C.c_int, [_P, _P, _P, _I64, _I64, _F]),
    "mhimx_softmax_cols_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _F]),
    "mhimx_colmax": (C.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "mhimx_rowmax": (C.c_int, [_P, _P, _I64, _I64, _P]),
    "mhimx_dsmil_head": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "mhimx_mul_colsum": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _P, _I64, _P]),
    "mhimx_rows_dpre": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _I32, _P, _I64, _P]),
    "mhimx_bmm_affine": (C.c_int, [_P, _I32, _P, _I32, _I64, _I64, _I64, C.c_float, C.c_float]),
    "mhimx_bmm_affine2": (C.c_int, [_P, _I32, _P, _I32, _I64, _I64, _I64, C.c_float, C.c_float, _P, C.c_float, C.c_float]),
    "mhimx_bmm_affine_pair": (C.c_int, [_P, _I32, C.POINTER(GemmNT), _F, _F, _I32, C.POINTER(GemmNT), _F, _F, _I32, _I64]),
    "mhimx_shard_flags": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _I64, _I32, _P]),
    "mhimx_shard_gather": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I64, _P]),
    "mhimx_shard_scatter": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I64, _P]),
    "mhimx_rows_dpre_image": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _I32, _P, _I64, _P]),
    "mhimx_rows_dpre_image_c": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _I32, _P, _I64, _P]),
    "mhimx_rows_dpre_image_k": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _I32, _P, _I64, _P]),
    "mhimx_wgrad_image_bytes": (_I64, [_I64, _I64]),
    "mhimx_wgrad_ws_floats": (_I64, [_I64, _I64, _I64]),
    "mhimx_bag_wgrad": (C.c_int, [_P, _P]),
    "mhimx_bag_wgrad_multi": (C.c_int, [_P, _P, _I32]),
    "mhimx_wgrad_multi_ws_floats": (C.c_int64, [_I64, _I64, _I64, _I32]),
    "mhimx_reduce_flush": (C.c_int, [_P, _P]),
    "mhimx_cls_metrics_ws_bytes": (C.c_int64, [_I64, _I64, _I64]),
    "mhimx_cls_metrics": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I32, _P, _I64, _P, _P, _I64]),
    "mhimx_colsum": (C.c_int, [_P, _P, _I64, _I64, _P, _I32, _P, _I64]),
    "mhimx_head_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _P, _P, _P, _P, _P, _I32, _P, _P]),
    "mhimx_adam_ema": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _F, _F, _F, _F, _F, _F, _I32, _P, _P, _I64]),
    "mhimx_bmm_chain": (C.c_int, [_P, C.POINTER(BmmStep), _I32, _I32, _P]),
    "mhimx_optim_step": (C.c_int, [_P, C.POINTER(OptimArgs)]),
    "mhimx_stream_copy": (C.c_int, [_P, _P, _P, _I64]),
    "mhimx_tick": (C.c_int, [_P, _P]),
    "mhimx_layernorm_fwd": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "mhimx_layernorm_bwd": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "mhimx_layernorm_bwd_res": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "mhimx_dropout_apply": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _U64, _P]),
    "mhimx_dropout_apply_proj": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _U64, _P]),
    "mhimx_softmax_rows": (C.c_int, [_P, _P, _P, _I64, _I64, _F]),
    "mhimx_softmax_rows_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _F]),
    "mhimx_landmark_mean": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _P]),
    "mhimx_landmark_mean_bwd": (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _I64, _I32]),
    "mhimx_affine_ident": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _F]),
    "mhimx_axpby": (C.c_int, [_P, _P, _P, _I64, _F, _F]),
    "mhimx_pinv_init": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P]),
    "mhimx_pinv_init_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P]),
    "mhimx_comm_unique_id": (C.c_int, [_P]),
    "mhimx_comm_init": (C.c_int, [C.POINTER(C.c_void_p), _P, _I32, _I32]),
    "mhimx_comm_allreduce": (C.c_int, [_P, _P, _P, _I64, _I32]),
    "mhimx_comm_destroy": (C.c_int, [_P]),
    "mhimx_bn_ws_floats": (_I64, [_I64, _I64]),
    "mhimx_bn_fwd": (C.c_int, [_P, _P, _I64, _I64, _P, _P, C.c_float, _I32, _P, _P, _P, _P, _P]),
    "mhimx_bn_bwd": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _I32, _P, _P, _P, _P]),
    "mhimx_sincos_add": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "mhimx_nys_ws_floats": (_I64, [_I64]),
    "mhimx_nys_a3v_fwd": (C.c_int, [_P, C.POINTER(Nys), _P, _P]),
    "mhimx_nys_out_fwd": (C.c_int, [_P, C.POINTER(Nys), _P, _P, _I64, _P, C.c_int32]),
    "mhimx_nys_out_bwd": (C.c_int, [_P, C.POINTER(Nys), _P, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P]),
    "mhimx_nys_a3v_bwd": (C.c_int, [_P, C.POINTER(Nys), _P, _P, _P, _P, _P, _I64, _I32, _P, _I64]),
    "mhimx_nys_cls_attn": (C.c_int, [_P, C.POINTER(Nys), _P, _P, _P]),
    "mhimx_resconv": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _I32, _I32]),
    "mhimx_resconv_dw_ws_floats": (_I64, [_I64, _I64, _I64, _I64]),
    "mhimx_resconv_dw": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I64, _I64, _I64, _P, _P]),
    "mhimx_ppeg_combine": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "mhimx_ppeg_fwd": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _I64]),
    "mhimx_ppeg_bwd_ws_floats": (_I64, [_I64, _I64]),
    "mhimx_ppeg_bwd": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _I64]),
    "mhimx_ppeg_side": (_I64, [_I64, _P]),
    "mhimx_ppeg_band_fwd": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "mhimx_ppeg_band_bwd_ws_floats": (_I64, [_I64, _I64]),
    "mhimx_ppeg_band_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "mhimx_scale_heads": (C.c_int, [_P, _P, _I64, _P, _I64, _I64, _I64, _I64, _P]),
    "mhimx_step_counts_of": (C.c_int, [_I64, C.c_double, C.c_double, C.c_double, C.POINTER(StepCounts)]),
    "mhimx_step_layout_of": (C.c_int, [C.POINTER(StepCfg), _I64, C.POINTER(StepCounts), C.POINTER(StepLayout)]),
    "mhimx_step_run": (C.c_int, [_P, C.POINTER(StepCfg), _P, _I64, _I64, _P, C.POINTER(StepCounts), C.POINTER(StepSeeds), _I64, _P, _I64, _I32]),
    "mhimx_step_project_ms": (C.c_int, [_P, _P, _I32]),
    "mhimx_window_layout_of": (C.c_int, [C.POINTER(StepCfg), _I32, _I64, C.POINTER(StepCounts), C.POINTER(WindowLayout)]),
    "mhimx_window_run": (C.c_int, [_P, C.POINTER(StepCfg), _I32, _P, _I64, _I64, _P, C.POINTER(StepCounts), C.POINTER(StepSeeds), _I64, _P, _I64, _I32]),   # (labels: void*[n])
    "mhimx_step_run_many": (C.c_int, [_P, C.POINTER(StepCfg), _I32, _P, _P, _P, _P, C.POINTER(StepCounts), C.POINTER(StepSeeds), _I64, _P, _I64]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X HIP library has not been built "
            f"(run `python -m mhim_mil_amd.build` or __graft_entry__.build()); there is no fallback path")
    # torch first: its wheel bundles its own HIP runtime, and libmhimx.so must bind to THAT copy (the one that owns the device
    # context and the pointers we are handed).  Loaded the other way round, the system runtime gets pulled in first and every
    # launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.mhimx_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} is ABI version {L.mhimx_version()}, this binding expects {ABI_VERSION}: a stale build "
                           f"(rebuild with `python -m mhim_mil_amd.build --force`)")
    _lib = L
    return L


class MhimxError(RuntimeError):
    pass


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib().mhimx_last_error()
        raise MhimxError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
