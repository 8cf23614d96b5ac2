"""mhimx_bmm_chain (include/mhimx.h): dependent batched 256^3 products in one persistent launch, against torch fp64 matmul of the
reference lines they replace (nystrom_attention.py:12-27) and against the launch-per-product path."""
import ctypes as C

import pytest
import torch

from mhim_mil_amd import _lib as L
from mhim_mil_amd import nystrom as NY

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _image(t):
    """What a split image holds, decoded back to fp32: hi + lo planes per head."""
    pl = t.view(torch.bfloat16).reshape(8, 2, 256, 256).float()
    return pl[:, 0] + pl[:, 1]


def _mats(n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return [torch.randn(8, 256, 256, device=DEV, generator=g) * 0.1 for _ in range(n)]


def test_split_images_and_products_match_torch():
    """kind 1 (images of alpha A + D), kind 0 with both outputs, all four images, identity terms."""
    a, b, d = _mats(3, 0)
    aN, bT, sN, sT, c1, pN, pT, pN2, pT2 = (torch.empty_like(a) for _ in range(9))
    steps = [NY._step(1, A=a, PN=aN), NY._step(1, A=b, PT=bT), NY._step(1, A=a, D=d, PN=sN, PT=sT, alpha=0.5),
             NY._step(0, A=aN, B=bT, C=c1, PN=pN, PT=pT, PN2=pN2, PT2=pT2, alpha=-1.0, ident=15.0, alpha2=0.25, ident2=7.0)]
    NY._run_chain(steps, 1, torch.device(DEV))
    eye = torch.eye(256, device=DEV, dtype=torch.float64)
    assert _rel(_image(aN), a) < 2e-5 and _rel(_image(bT), b.transpose(1, 2)) < 2e-5                       # hi + lo = 16 mantissa bits
    s = 0.5 * a.double() + d.double()
    assert _rel(_image(sN), s) < 2e-5 and _rel(_image(sT), s.transpose(1, 2)) < 2e-5
    ref = a.double() @ b.double()
    x1, x2 = 15 * eye - ref, 7 * eye + 0.25 * ref
    assert _rel(c1, x1) < 2e-5
    assert _rel(_image(pN), x1) < 3e-5 and _rel(_image(pT), x1.transpose(1, 2)) < 3e-5
    assert _rel(_image(pN2), x2) < 3e-5 and _rel(_image(pT2), x2.transpose(1, 2)) < 3e-5
    assert not NY.chain_gave_up(torch.device(DEV))
    ctr = NY._chain_counters(torch.device(DEV))
    assert int(ctr[:513].abs().sum()) == 0                                                                          # re-usable as they are


def test_two_groups_addend_and_idle_slots():
    """groups = 2: two independent products per stage (256 workgroups), an idle slot, the addend D (also in place, C = D), and a stage
    reading both products of the stage before it."""
    a, b, c, d = _mats(4, 1)
    aN, bT, cN, dT, p, q, qN, pT, r = (torch.empty_like(a) for _ in range(9))
    acc = d.clone()
    steps = [NY._step(1, A=a, PN=aN), NY._step(1, A=b, PT=bT),
             NY._step(1, A=c, PN=cN), NY._step(1, A=d, PT=dT),
             NY._step(0, A=aN, B=bT, C=p, PT=pT, alpha=0.25), NY._step(0, A=cN, B=dT, C=q, PN=qN, alpha=-1.0),      # p = a b / 4 | q = -c d
             NY._step(0, A=qN, B=pT, C=r, D=a, alpha=1.0), NY._IDLE(),                                              # r = a + q p
             NY._step(0, A=aN, B=dT, C=acc, D=acc, alpha=-1.0), NY._IDLE()]                                         # acc = d - a d  (in place)
    NY._run_chain(steps, 2, torch.device(DEV))
    A, B, Cc, D = (t.double() for t in (a, b, c, d))
    P, Q = 0.25 * A @ B, -(Cc @ D)
    assert _rel(p, P) < 2e-5 and _rel(q, Q) < 2e-5
    assert _rel(r, A + Q @ P) < 5e-5
    assert _rel(acc, D - A @ D) < 2e-5
    assert not NY.chain_gave_up(torch.device(DEV))


@pytest.mark.parametrize("levels3", [False, True])
def test_pinv_chain_equals_launch_path_and_is_reproducible(levels3):
    """The forward (one launch) and backward (two launches) of the pseudo-inverse against the launch-per-product path (the same 3-term
    products, another summation order: fp32 rounding, amplified by the iteration), and bit-reproducible under other load.  levels3: the
    expanded three-level form of the iteration (MHIMX_PINV_LEVELS=3; round 5) - the same polynomial, the same bounds."""
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    lm = torch.randn(256, 2 * 512, device=DEV, generator=g) * 0.5
    dz = torch.randn(8, 256, 256, device=DEV, generator=g) * 0.1

    def run():
        a2, z, z0, stats, chain = NY._landmark_pinv_forward(lm, 0.125)
        dlm = torch.empty_like(lm)
        NY._landmark_pinv_backward(lm, 0.125, a2, z0, stats, chain, dz.clone(), dlm, accumulate=False)
        return z, dlm

    old, old3 = NY._CHAIN, NY._PINV3
    try:
        NY._CHAIN = False
        z0_, d0 = run()
        NY._CHAIN, NY._PINV3 = True, levels3
        z1, d1 = run()
        assert _rel(z1, z0_) < 1e-4 and _rel(d1, d0) < 1e-4
        side, big = torch.cuda.Stream(), torch.randn(4096, 4096, device=DEV)
        for rep in range(20):
            if rep % 2:
                with torch.cuda.stream(side):
                    big @ big
            z2, d2 = run()
            assert torch.equal(z2, z1) and torch.equal(d2, d1)
        torch.cuda.synchronize()
        assert not NY.chain_gave_up(dev)
    finally:
        NY._CHAIN, NY._PINV3 = old, old3


def test_scaled_addends_of_a_product():
    """Round 5: X = ident I + alpha A B + dscale D + d2scale D2 (the expanded pseudo-inverse polynomial's levels and its backward)."""
    a, b, d, e = _mats(4, 7)
    aN, bT, x, y, yN = (torch.empty_like(a) for _ in range(5))
    steps = [NY._step(1, A=a, PN=aN), NY._step(1, A=b, PT=bT),
             NY._step(0, A=aN, B=bT, C=x, alpha=-1.0, ident=-15.0, D=d, dscale=7.0), NY._step(0, A=aN, B=bT, C=y, PN=yN, alpha=0.25, D=d, D2=e, dscale=3.25)]
    NY._run_chain(steps, 2, torch.device(DEV))
    A, B, D, E = (t.double() for t in (a, b, d, e))
    eye = torch.eye(256, device=DEV, dtype=torch.float64)
    assert _rel(x, -15 * eye - A @ B + 7 * D) < 2e-5
    assert _rel(y, 0.25 * A @ B + 3.25 * D + E) < 2e-5 and _rel(_image(yN), 0.25 * A @ B + 3.25 * D + E) < 3e-5
    assert not NY.chain_gave_up(torch.device(DEV))


def test_chain_rejects_bad_tables():
    a, = _mats(1, 3)
    ctr = NY._chain_counters(torch.device(DEV))
    lib = L.lib()
    one = (L.BmmStep * 1)(NY._step(0, A=a, B=a, C=a))                                 # overwrites its own operand
    assert lib.mhimx_bmm_chain(NY._st(), one, 1, 1, NY._ptr(ctr)) != 0
    many = (L.BmmStep * 64)(*[NY._step(1, A=a, PN=torch.empty_like(a)) for _ in range(64)])
    assert lib.mhimx_bmm_chain(NY._st(), many, 64, 1, NY._ptr(ctr)) != 0              # more than 38 steps
    two = (L.BmmStep * 1)(NY._step(0, A=a, B=a, C=torch.empty_like(a), PN2=torch.empty_like(a), D=a))
    assert lib.mhimx_bmm_chain(NY._st(), two, 1, 1, NY._ptr(ctr)) != 0                # an addend with two outputs
    assert lib.mhimx_bmm_chain(NY._st(), one, 1, 3, NY._ptr(ctr)) != 0                # groups


def test_chains_competing_for_the_cus_complete_and_agree():
    """Three two-group chains (256 workgroups each, one per CU) launched together on three streams cannot all be resident: the ticket
    order lets each complete with whatever part of its grid runs (the first form of the kernel - tile = block index - hung until its
    spin bound and returned garbage here).  Every chain must reproduce the result of running alone, bit for bit."""
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV).manual_seed(4)
    lm = torch.randn(256, 2 * 512, device=DEV, generator=g) * 0.5
    dz = torch.randn(8, 256, 256, device=DEV, generator=g) * 0.1

    def run():
        a2, z, z0, stats, chain = NY._landmark_pinv_forward(lm, 0.125)
        dlm = torch.empty_like(lm)
        NY._landmark_pinv_backward(lm, 0.125, a2, z0, stats, chain, dz.clone(), dlm, accumulate=False)
        return z, dlm

    old = NY._CHAIN
    try:
        NY._CHAIN = True
        z_ref, d_ref = run()
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(3)]
        for rep in range(5):
            outs = []
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(run())
            torch.cuda.synchronize()
            for z, d in outs:
                assert torch.equal(z, z_ref) and torch.equal(d, d_ref)
        assert not NY.chain_gave_up(dev)
        for c in NY._CTRS.values():
            assert int(c[:513].abs().sum()) == 0
    finally:
        NY._CHAIN = old


def test_one_stage_chain_and_empty_permutation():
    """Edge cases: a chain of ONE stage (nothing to wait for), and mhimx_random_perm of zero elements."""
    from mhim_mil_amd import ops
    a, = _mats(1, 5)
    aN, aT = torch.empty_like(a), torch.empty_like(a)
    NY._run_chain([NY._step(1, A=a, PN=aN, PT=aT, alpha=2.0)], 1, torch.device(DEV))
    assert _rel(_image(aN), 2 * a) < 2e-5 and _rel(_image(aT), 2 * a.transpose(1, 2)) < 2e-5
    assert int(NY._chain_counters(torch.device(DEV))[:513].abs().sum()) == 0
    assert ops.random_perm(0, 1, device=torch.device(DEV)).numel() == 0
