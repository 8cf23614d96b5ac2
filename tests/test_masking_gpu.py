"""mhim_mil_amd/masking.py (the reference's free functions select_mask_fn / mask_fn, masking.py:9-110) against the reference's own
outputs (tests/golden g5_*) and the oracle: both multi-head fusions, select_inv, the union with an earlier mask."""
import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def test_mean_fusion_matches_the_reference():
    from mhim_mil_amd.masking import select_mask_fn
    meta, a = G.load("g5_select_mean_n600")
    lk, ids = select_mask_fn(meta["n"], _t(a["attn"])[None], True, meta["mask_ratio_h"], len_keep_other=meta["n"],
                             random_ratio=meta["mask_ratio_hr"], msa_fusion="mean", perm=a["perm"])
    assert lk == int(a["len_keep"]) and ids.shape == (1, meta["n"])
    ids = ids[0].cpu().numpy()
    assert np.array_equal(ids[:lk], a["kept"]) and np.array_equal(ids[lk:], a["masked"])


def test_select_inv_matches_the_reference():
    from mhim_mil_amd.masking import select_mask_fn
    meta, a = G.load("g5_select_inv_n512")
    lk, ids = select_mask_fn(meta["n"], _t(a["score"])[None], True, meta["mask_ratio_h"], select_inv=True)
    ids = ids[0].cpu().numpy()
    assert lk == int(a["len_keep"])
    assert np.array_equal(np.sort(ids[:lk]), np.sort(a["first"])) and np.array_equal(ids[lk:], a["rest"])
    assert np.array_equal(ids[:lk], a["first"])              # tie-free scores: torch.topk's order is value-descending, as ours


def test_two_d_and_vote_match_the_reference_fixtures():
    from mhim_mil_amd.masking import select_mask_fn
    meta, a = G.load("g5_select_low_n512")
    lk, ids = select_mask_fn(meta["n"], _t(a["score"])[None], False, meta["mask_ratio_l"])
    ids = ids[0].cpu().numpy()
    assert lk == int(a["len_keep"]) and np.array_equal(ids[:lk], a["kept"]) and np.array_equal(np.sort(ids[lk:]), np.sort(a["masked"]))
    meta, a = G.load("g5_select_vote_n600")
    lk, ids = select_mask_fn(meta["n"], _t(a["attn"])[None], True, meta["mask_ratio_h"], len_keep_other=meta["n"],
                             random_ratio=meta["mask_ratio_hr"], msa_fusion="vote", perm=a["perm"])
    ids = ids[0].cpu().numpy()
    assert lk == int(a["len_keep"]) and len(set(ids.tolist())) == meta["n"] and np.all(np.diff(ids[:lk]) > 0)


@pytest.mark.parametrize("fusion", ["mean", "vote"])
@pytest.mark.parametrize("inv", [False, True])
def test_options_against_the_oracle_with_an_earlier_mask(fusion, inv):
    """The union with an earlier mask (masking.py:36-39,74-75) under both fusions and select_inv: tie-free per-head scores."""
    from mhim_mil_amd.masking import select_mask_fn
    n, h = 900, 8
    a = ((np.stack([synth.permutation(300 + i, n) for i in range(h)]) + 0.25 * synth.uniform(91, (h, n))) / n).astype(np.float32)
    other = np.sort(synth.permutation(17, n)[:60]).astype(np.int64)
    ratio, rr = 0.12, 0.5
    k = int(np.ceil(n * ratio / rr))
    if fusion == "mean":
        kk = k // h
        kc = len(np.unique(np.concatenate([O.topk_indices(a[i], kk, True) for i in range(h)])))
    else:
        kc = k
    perm = synth.permutation(23, kc)
    lk_o, ids_o, _ = O.select_mask(n, a, True, ratio, other_masked=other, random_ratio=rr, perm=perm, select_inv=inv, msa_fusion=fusion)
    rest = np.setdiff1d(np.arange(n), other)
    prev = _t(np.concatenate([rest, other]))[None]            # the earlier call's mask_ids: kept ++ masked (mhim.py:136-139 passes both)
    lk, ids = select_mask_fn(n, _t(a)[None], True, ratio, mask_ids_other=prev, len_keep_other=len(rest), cls_attn_topk_idx_other=_t(other),
                             random_ratio=rr, select_inv=inv, msa_fusion=fusion, perm=perm)
    # (the earlier mask derived from mask_ids_other alone shrinks ps for the ratio, masking.py:37-40: the oracle's shrink_ps)
    lk2_o, ids2_o, _ = O.select_mask(n, a[0], True, ratio, other_masked=other, shrink_ps=True)
    lk2, ids2 = select_mask_fn(n, _t(a[0])[None], True, ratio, mask_ids_other=prev, len_keep_other=len(rest))
    assert lk2 == lk2_o and np.array_equal(ids2[0].cpu().numpy()[:lk2], ids2_o[:lk2_o])
    ids = ids[0].cpu().numpy()
    assert lk == lk_o
    if fusion == "mean":                                      # candidate order is fixed (ascending union): exact
        assert np.array_equal(ids, ids_o)
    else:                                                     # votes tie structurally: the sets obey the contract, not one order
        assert len(set(ids.tolist())) == n and set(other.tolist()) <= set((ids[:lk] if inv else ids[lk:]).tolist())


def test_mask_fn_gathers_the_kept_rows():
    from mhim_mil_amd.masking import mask_fn
    x = torch.randn(1, 700, 96, device=DEV)
    ids = _t(synth.permutation(5, 700).astype(np.int64))[None]
    out = mask_fn(x, ids, 333)
    assert out.shape == (1, 333, 96) and torch.equal(out[0], x[0][ids[0, :333]])


# ---------------------------------------------------------------------------------------------------- scoring.py's free functions
def test_get_pseudo_score_matches_the_oracle():
    from mhim_mil_amd.scoring import get_pseudo_score
    n, E, Cc = 1500, 512, 2
    h = torch.from_numpy(synth.normal(61, (n, E), std=1.0).astype(np.float32)).abs()
    s = torch.from_numpy(synth.normal(62, (n,), std=2.0).astype(np.float32))
    attn = torch.softmax(s, 0)
    clf = torch.nn.Linear(E, Cc)
    with torch.no_grad():
        clf.weight.copy_(torch.from_numpy(synth.normal(63, (Cc, E), std=0.5).astype(np.float32)))
        clf.bias.copy_(torch.tensor([0.3, -0.2]))
    ref = O.pseudo_score(h.double(), attn.double(), clf.weight.detach().double(), clf.bias.detach().double())
    got = get_pseudo_score(clf.to(DEV), h.to(DEV)[None], attn.to(DEV)[None])
    assert got.shape == (1, n)
    assert float((got[0].cpu().double() - ref).abs().max()) <= 2e-5


def test_get_pseudo_score_trans_matches_the_oracle():
    from mhim_mil_amd.scoring import get_pseudo_score_trans
    n, hh, d, E, Cc = 900, 8, 64, 512, 2
    v = torch.from_numpy(synth.normal(71, (hh, n, d), std=1.0).astype(np.float32))
    attn = torch.softmax(torch.from_numpy(synth.normal(72, (hh, n), std=2.0).astype(np.float32)), 1)
    to_out = torch.nn.Sequential(torch.nn.Linear(hh * d, E), torch.nn.Dropout(0.1)).eval()
    clf = torch.nn.Linear(E, Cc)
    with torch.no_grad():
        to_out[0].weight.copy_(torch.from_numpy(synth.normal(73, (E, hh * d), std=0.05).astype(np.float32)))
        to_out[0].bias.copy_(torch.from_numpy(synth.normal(74, (E,), std=0.1).astype(np.float32)))
        clf.weight.copy_(torch.from_numpy(synth.normal(75, (Cc, E), std=0.5).astype(np.float32)))
        clf.bias.copy_(torch.tensor([0.1, 0.4]))
    params = {"p.to_out.0.weight": to_out[0].weight.detach().double(), "p.to_out.0.bias": to_out[0].bias.detach().double(),
              "predictor.weight": clf.weight.detach().double(), "predictor.bias": clf.bias.detach().double()}
    ref = O.pseudo_score_trans(v.double(), attn.double(), params, to_out_prefix="p.")
    got = get_pseudo_score_trans(clf.to(DEV), v.to(DEV)[None], attn.to(DEV)[None], to_out.to(DEV))
    assert got.shape == (1, n)
    assert float((got[0].cpu().double() - ref).abs().max()) <= 5e-5
