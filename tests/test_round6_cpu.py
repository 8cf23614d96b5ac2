"""Round-6 host-logic tests that need no GPU: the accumulation-window executor's layout and argument checks (csrc/step.hip: check_window,
layout - host code), the gradient-slab description of mhimx_optim_args."""
import ctypes as C

import pytest

from mhim_mil_amd import _lib as L


def _cfg(D=1024, C_=2, k=5, g=1 << 20, n_all=None):
    """A mhimx_step_cfg over a fake flat gradient buffer at address ``g`` (nothing is dereferenced on the host): the reference-ordered views
    feature.0.weight | feature.0.bias | attention | predictor | merge.*, every tensor padded to 4 floats."""
    E, A, I = 512, 128, 512
    sizes = [("w1", E * D), ("b1", E), ("wa", A * E), ("wc", A), ("wp", C_ * E), ("bp", C_), ("ln_w", E), ("ln_b", E), ("wkv", 2 * I * E),
             ("wq", I * E), ("wo", E * I), ("bo", E)]
    off, views = 0, {}
    for name, n in sizes:
        views[name] = g + 4 * off
        off += (n + 3) // 4 * 4
    par = L.StepParams(**{f: 256 for f, _ in L.StepParams._fields_})
    cfg = L.StepCfg(D=D, E=E, A=A, C=C_, k=k, act=2, da_act=1, attn2score=1, student=par, teacher=par, grad=L.StepGrads(**views), tick=256,
                    p=256, g=g, m=256, v=256, n_train=off, n_all=n_all if n_all is not None else off + 4 * 512)
    return cfg, views, off


def _layout(cfg, n_bags, N):
    lib = L.lib()
    cnt, lay = L.StepCounts(), L.WindowLayout()
    assert lib.mhimx_step_counts_of(N, 0.03, 0.5, 0.9, C.byref(cnt)) == 0
    rc = lib.mhimx_window_layout_of(C.byref(cfg), n_bags, N, C.byref(cnt), C.byref(lay))
    return rc, cnt, lay


def test_window_layout_is_n_copies_of_the_per_bag_part_at_one_stride():
    cfg, _, n_train = _cfg()
    lib = L.lib()
    for n_bags, N in ((2, 64), (8, 10000), (8, 16384), (3, 1500)):
        rc, cnt, lay = _layout(cfg, n_bags, N)
        assert rc == 0, lib.mhimx_last_error()
        assert lay.bag0 % 256 == 0 and lay.bag_stride % 256 == 0 and lay.total == lay.bag0 + n_bags * lay.bag_stride
        bag = lay.bag
        for f in ("logits", "losses", "score", "rows_all", "H_teacher", "H_student", "dact", "z_teacher", "z_student", "g_z", "dH"):
            o = getattr(bag, f)
            assert lay.bag0 <= o < lay.bag0 + lay.bag_stride and o % 256 == 0, (f, o)
        assert lay.bag0 <= lay.grad_slab and lay.grad_slab + 4 * cfg.n_all <= lay.bag0 + lay.bag_stride
        # one bag's share is what a single step takes, less the weight-gradient workspace the window shares, plus its slab and query scratch
        one = L.StepLayout()
        assert lib.mhimx_step_layout_of(C.byref(cfg), N, C.byref(cnt), C.byref(one)) == 0
        assert lay.bag_stride < one.total + 4 * cfg.n_all + (1 << 16)
    r8 = _layout(cfg, 8, 10000)[2]
    r4 = _layout(cfg, 4, 10000)[2]
    assert r8.bag_stride == r4.bag_stride and r8.total > r4.total


@pytest.mark.parametrize("what", ["one bag", "nine bags", "too many rows", "merge_k 7", "q_out", "side_stream", "w1 not first", "view outside g",
                                  "n_all % 4"])
def test_window_refuses_what_it_cannot_batch(what):
    lib = L.lib()
    cfg, views, n_train = _cfg()
    n_bags, N = 8, 10000
    if what == "one bag":
        n_bags = 1
    elif what == "nine bags":
        n_bags = 9
    elif what == "too many rows":
        N = 16385
    elif what == "merge_k 7":
        cfg.k = 7
    elif what == "q_out":
        cfg.q_out = 256
    elif what == "side_stream":
        cfg.side_stream = 256
    elif what == "w1 not first":
        g = cfg.grad
        g.w1, g.b1 = views["b1"], views["w1"]
        cfg.grad = g
    elif what == "view outside g":
        g = cfg.grad
        g.bo = views["w1"] + 4 * (n_train + 64)
        cfg.grad = g
    elif what == "n_all % 4":
        cfg.n_all = n_train + 2
    rc, _, _ = _layout(cfg, n_bags, N)
    assert rc != 0 and lib.mhimx_last_error()


def test_window_run_checks_its_arguments_before_it_touches_the_device():
    """Null bag / label tables and a short workspace are refused on the host (no launch is attempted: this runs without a GPU)."""
    lib = L.lib()
    cfg, _, _ = _cfg()
    rc, cnt, lay = _layout(cfg, 2, 512)
    assert rc == 0
    seeds = (L.StepSeeds * 2)()
    X = (C.c_void_p * 2)(4096, 8192)
    lab = (C.c_void_p * 2)(4096, 4104)
    assert lib.mhimx_window_run(None, C.byref(cfg), 2, None, 1024, 512, lab, C.byref(cnt), seeds, 1, 4096, lay.total, 1) != 0
    assert lib.mhimx_window_run(None, C.byref(cfg), 2, X, 1024, 512, None, C.byref(cnt), seeds, 1, 4096, lay.total, 1) != 0
    assert lib.mhimx_window_run(None, C.byref(cfg), 2, X, 1024, 512, lab, C.byref(cnt), seeds, 1, 4096, lay.total - 1, 1) != 0
    assert b"workspace" in lib.mhimx_last_error()
    Xbad = (C.c_void_p * 2)(4096, 8200)                       # bag 1 not 16-byte aligned
    assert lib.mhimx_window_run(None, C.byref(cfg), 2, Xbad, 1024, 512, lab, C.byref(cnt), seeds, 1, 4096, lay.total, 1) != 0


def test_optim_args_describe_the_slabs():
    """mhimx_optim_args.extra_lo / extra_only only make sense with gradient slabs, and extra_lo is a multiple of 4 (the update kernel's
    vector path keeps four elements on one side of it): refused on the host otherwise."""
    lib = L.lib()
    base = dict(p=4096, g=8192, m=12288, v=16384, n_train=1024, n_all=1024, step=1, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8)
    for bad in (dict(extra_lo=4), dict(extra_only=1), dict(g_extra=20480, n_extra=2, extra_pitch=1024, extra_lo=6)):
        a = L.OptimArgs(**base, **bad)
        assert lib.mhimx_optim_step(None, C.byref(a)) != 0, bad
        assert b"extra_lo" in lib.mhimx_last_error() or b"slab" in lib.mhimx_last_error()
